// ilupp_amd/csrc/ichol.hip -- incomplete Cholesky for gfx950: IChol(0) (first part) and ICholT (second part).
//
// Replaces IChol0 / compute_ichol0 / sparse_dot_product (reference IChol.hpp:16-73) and
// natural_triangular_part (sparse_implementation.h:2568-2592).
//
// Layout: L is CSR (for either input orientation), lower triangle of the input's major slices with the
// diagonal LAST in each row (IChol.hpp:63-73).  Arithmetic follows the reference entry by entry: for (i,j) in
// row order, dp = sum over matching columns of row i and row j (both without their last entry), accumulated
// in ascending column order with separate multiply and add; (A_ij - dp)/L_jj below the diagonal,
// sqrt(A_ii - dp) on it.
//
// Execution: first-generation persistent dataflow kernel (one launch, thread-per-row-block, rows of a lane in
// order, finished rows published write-through + drained + flagged, consumers poll the flag; see ilu0.hip for
// the protocol).  The working row lives in LDS.  This is the correctness-first kernel of this path; it does not
// yet have the loader/consumer structure of the ILU(0) kernels.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ilupp {

// ---------------------------------------------------------------------------------------------
// triangular part (keep idx <= major), sparse_implementation.h:2568-2592 with increasing = true
// ---------------------------------------------------------------------------------------------
__global__ void k_tri_count(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int lower,
                            int32_t *__restrict__ cnt, int32_t *__restrict__ bad_diag)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int c = 0, has = 0;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int j = idx[q];
        c += lower ? (j <= r) : (j >= r);
        has |= (j == r);
    }
    cnt[r] = c;
    if (!has) atomicMin(bad_diag, r);
}

__global__ void k_tri_fill(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                           const double *__restrict__ val, int lower, const int32_t *__restrict__ tptr,
                           int32_t *__restrict__ tidx, double *__restrict__ tval)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int o = tptr[r];
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int j = idx[q];
        if (lower ? (j <= r) : (j >= r)) { tidx[o] = j; tval[o] = val[q]; ++o; }
    }
}

int triangular_part(hipStream_t st, const DevMat &A, bool lower, DevMat *T, int32_t *first_missing_diag)
{
    const int32_t n = A.n;
    int32_t *cnt = nullptr, *bad = nullptr;
    ILUPP_HIP(pool_malloc(&cnt, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&bad, 16));
    const int32_t big = 0x7fffffff;
    ILUPP_HIP(hipMemcpyAsync(bad, &big, sizeof(int32_t), hipMemcpyHostToDevice, st));
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_tri_count, dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, lower ? 1 : 0, cnt, bad);
    T->n = n; T->is_csr = true; T->owns = true;
    ILUPP_HIP(pool_malloc(&T->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(hipMemsetAsync(T->ptr, 0, sizeof(int32_t), st));
    size_t tmp_bytes = 0;
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, cnt, T->ptr + 1, n, st));
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, cnt, T->ptr + 1, n, st));
    int32_t tot = 0, miss = 0;
    ILUPP_HIP(hipMemcpyAsync(&tot, T->ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipMemcpyAsync(&miss, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    ILUPP_HIP(pool_free(tmp)); ILUPP_HIP(pool_free(cnt)); ILUPP_HIP(pool_free(bad));
    T->nnz = tot;
    ILUPP_HIP(pool_malloc(&T->idx, sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1)));
    ILUPP_HIP(pool_malloc(&T->val, sizeof(double) * (size_t)(tot > 0 ? tot : 1)));
    hipLaunchKernelGGL(k_tri_fill, dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, A.val, lower ? 1 : 0, T->ptr, T->idx, T->val);
    if (first_missing_diag) *first_missing_diag = (miss == big) ? -1 : miss;
    return (miss == big) ? ILUPP_OK : ILUPP_ERR_NO_DIAGONAL;
}

// ---------------------------------------------------------------------------------------------
// numeric IChol(0): in place on L.val (on entry the lower triangle of A, on exit the factor)
// ---------------------------------------------------------------------------------------------
static constexpr unsigned kCholSpinLimit = 1u << 22;

template <int MAXLEN, bool GLOBAL_W>
__global__ void __launch_bounds__(kThreads)
k_ichol0_numeric(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, double *val,
                 int32_t nb, const int32_t *__restrict__ bstart, int32_t *done, int32_t *ctrl,
                 double *wscratch, int32_t wstride)
{
    extern __shared__ __attribute__((aligned(16))) double wlds[];
    __shared__ unsigned wg_ticket;
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(&ctrl[0], 1);
    __syncthreads();
    const int tid = threadIdx.x;
    const int64_t slot = (int64_t)wg_ticket * kThreads + tid;
#define W(q) (GLOBAL_W ? wscratch[(size_t)slot * wstride + (q)] : wlds[(q) * kThreads + tid])

    int r = 0, rend = 0;
    if (slot < nb) { r = bstart[slot]; rend = bstart[slot + 1]; }
    bool active = r < rend;
    bool need_init = true;
    int lo = 0, len = 0, p = 0;
    unsigned spins = 0;

    for (;;) {
        if (!__any(active)) break;
        bool progressed = false;
        if (active) {
            if (need_init) {
                lo = ptr[r];
                len = ptr[r + 1] - lo;
                for (int q = 0; q < len; ++q) W(q) = val[lo + q];      // A's lower row (IChol.hpp:47)
                p = 0;
                need_init = false;
                progressed = true;
            }
            while (p < len) {
                const int j = idx[lo + p];
                double dp = 0.0;                                        // sparse_dot_product, IChol.hpp:16-28
                if (j < r) {
                    if (ld_agent_i32(&done[j]) == 0) break;             // row j not finished: retry next round
                    order_after_poll();
                    const int j0 = ptr[j], j1 = ptr[j + 1] - 1;         // row j minus its diagonal
                    int a = 0, b = j0;
                    while (a < len - 1 && b < j1) {                     // row i minus its diagonal
                        const int ca = idx[lo + a], cb = idx[b];
                        if (ca == cb) { const double pr = W(a) * ld_agent_f64(&val[b]); dp = dp + pr; ++a; ++b; }
                        else if (ca < cb) ++a;
                        else ++b;
                    }
                    const double L_jj = ld_agent_f64(&val[j1]);         // diagonal is the last entry of row j
                    W(p) = (W(p) - dp) / L_jj;                          // IChol.hpp:49-51
                } else {
                    // diagonal: dot product of the row with itself (without the diagonal)
                    for (int a = 0; a < len - 1; ++a) { const double pr = W(a) * W(a); dp = dp + pr; }
                    W(p) = sqrt(W(p) - dp);                             // IChol.hpp:52-53
                }
                ++p;
                progressed = true;
            }
            if (p == len) {
                for (int q = 0; q < len; ++q) st_agent_f64(&val[lo + q], W(q));
                drain_stores();
                st_agent_i32(&done[r], 1);
                ++r;
                need_init = true;
                active = r < rend;
                progressed = true;
            }
        }
        if (__any(progressed)) {
            spins = 0;
        } else {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kCholSpinLimit) {
                if ((tid & 63) == 0) atomicExch(&ctrl[1], 1);
                break;
            }
        }
    }
#undef W
}

int ichol0_numeric(hipStream_t st, DevMat *L, const Schedule &fwd, int32_t max_row_len, int32_t *d_done, int32_t *d_ctrl,
                   float *kernel_ms)
{
    const int32_t n = L->n;
    ILUPP_HIP(hipMemsetAsync(d_done, 0, sizeof(int32_t) * (size_t)n, st));
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    const unsigned grid = (unsigned)((fwd.nb + kThreads - 1) / kThreads);
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    double *wscratch = nullptr;
    ILUPP_HIP(hipEventRecord(e0, st));
#define LAUNCH(ML, GW, LDSB)                                                                               \
    hipLaunchKernelGGL((k_ichol0_numeric<ML, GW>), dim3(grid), dim3(kThreads), (LDSB), st, L->ptr, L->idx, L->val, \
                       fwd.nb, fwd.start, d_done, d_ctrl, wscratch, max_row_len)
    if (max_row_len <= 8) LAUNCH(8, false, 8 * kThreads * sizeof(double));
    else if (max_row_len <= 16) LAUNCH(16, false, 16 * kThreads * sizeof(double));
    else if (max_row_len <= 32) LAUNCH(32, false, 32 * kThreads * sizeof(double));
    else if (max_row_len <= 64) {
        ILUPP_HIP(hipFuncSetAttribute((const void *)k_ichol0_numeric<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * kThreads * (int)sizeof(double)));
        LAUNCH(64, false, 64 * kThreads * sizeof(double));
    } else {
        ILUPP_HIP(pool_malloc(&wscratch, sizeof(double) * (size_t)grid * kThreads * (size_t)max_row_len));
        LAUNCH(1, true, 0);
    }
#undef LAUNCH
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4];
    ILUPP_HIP(hipMemcpyAsync(ctrl, d_ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    ILUPP_HIP(hipEventDestroy(e0));
    ILUPP_HIP(hipEventDestroy(e1));
    if (wscratch) ILUPP_HIP(pool_free(wscratch));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

}  // namespace ilupp

// =============================================================================================
// ICholT: incomplete Cholesky with threshold and additional fill (reference IChol.hpp:78-164)
// =============================================================================================
// Left-looking by columns.  Two things in the reference make the column order matter for the BITS of the
// result, not just for correctness: (1) column j subtracts the contributing columns k in the order of a linked
// list that is re-threaded after every column (ILUC.hpp:37-63), (2) every column j updates the running diagonal
// D[i] of ALL rows i > j of its PRE-drop working column (IChol.hpp:135-141).  The dataflow kernel in
// icholt_df.hip reproduces both in parallel (its largest capacity class gives one wave the whole LDS of a CU); this
// file only holds the entry point.
namespace ilupp {


__global__ void k_fill_i32(int32_t *p, long count, int32_t v)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < count; i += stride) p[i] = v;
}

// A_tri: lower triangle by columns (CSC of A's lower part, diagonal first).  L comes back CSC-labelled.
int icholt_factor(hipStream_t st, const DevMat &Atri, int32_t add_fill_in, double threshold, DevMat *L, float *kernel_ms)
{
    const int rc = icholt_factor_df(st, Atri, add_fill_in, threshold, L, kernel_ms);
    return rc == 1 ? ILUPP_ERR_MEMORY : rc;      // (1 = a working column that does not fit the largest capacity class)
}

}  // namespace ilupp
