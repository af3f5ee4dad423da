// ilupp_amd/csrc/ilu0.hip -- ILU(0) for gfx950: symbolic split + persistent dataflow numeric kernel.
//
// Replaces ILU0.hpp:69-106 of the reference (compute_ilu0 :26-66, sparse_vec_update :8-23).
// Arithmetic follows the reference operation for operation (row-wise IKJ, ascending k, merge update
// `u_ij -= l_ik * u_kj` as separate multiply and subtract: the library is built with
// -ffp-contract=off), so factor VALUES are bit-identical to the CPU, not merely within 1e-12.
//
// Layout in HBM (all int32 / fp64):
//   L  CSR, strictly-lower entries of the row in ascending column order, then the unit diagonal LAST
//   U  CSR, the diagonal FIRST, then the strictly-upper entries ascending
// The symbolic pass writes both patterns (and L's unit diagonal) once; the numeric pass streams A's
// values in, keeps the working row in LDS, and writes L/U values once.  Algorithmic traffic per row
// of the 7-point problem: 88 B in, 104 B out.
//
// Numeric kernel = one launch, no per-level launches and no grid barrier:
//   * a lane owns a contiguous block of rows (symbolic.hip) and walks them in order;
//   * row r may use row k<r only after done[k] is set: the U row of k is stored write-through (sc1),
//     the storing wave drains its stores (s_waitcnt vmcnt(0)), then sets done[k] (sc1); consumers
//     poll done[k] with sc1 loads and read the U row with sc1 loads (MI355X has 8 XCDs with private,
//     mutually non-coherent L2s: cdna_hip_programming.md guideline 16, form R1);
//   * lanes never block inside divergent code: every lane retries its pending dependency once per
//     wave iteration, so a lane may wait on another lane of its own wave;
//   * workgroup ids come from an atomic ticket, so every block a lane can wait on belongs to a
//     workgroup that has already started (forward progress without assuming dispatch order).
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ilupp {

// ---------------------------------------------------------------------------------------------
// symbolic
// ---------------------------------------------------------------------------------------------
__global__ void k_ilu0_count(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                             int32_t *__restrict__ lrow, int32_t *__restrict__ urow, int32_t *__restrict__ missing)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int lo = ptr[r], hi = ptr[r + 1];
    int cl = 0, ceq = 0;
    for (int q = lo; q < hi; ++q) {
        const int c = idx[q];
        cl += (c < r);
        ceq += (c == r);
    }
    lrow[r] = cl + 1;                 // + unit diagonal (ILU0.hpp:93)
    urow[r] = (hi - lo) - cl;         // entries with column >= r (ILU0.hpp:43)
    if (ceq == 0) atomicMin(missing, r);
}

__global__ void k_ilu0_pattern(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                               const int32_t *__restrict__ Lptr, const int32_t *__restrict__ Uptr,
                               int32_t *__restrict__ Lidx, double *__restrict__ Lval, int32_t *__restrict__ Uidx)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int lo = ptr[r], hi = ptr[r + 1];
    int l = Lptr[r], u = Uptr[r];
    for (int q = lo; q < hi; ++q) {
        const int c = idx[q];
        if (c < r) Lidx[l++] = c; else Uidx[u++] = c;
    }
    Lidx[l] = r;
    Lval[l] = 1.0;
}

int ilu0_symbolic(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, int32_t *first_missing_diag)
{
    const int32_t n = A.n;
    int32_t *lrow, *urow, *missing;
    ILUPP_HIP(hipMalloc(&lrow, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(hipMalloc(&urow, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(hipMalloc(&missing, sizeof(int32_t)));
    const int32_t big = 0x7fffffff;
    ILUPP_HIP(hipMemcpyAsync(missing, &big, sizeof(int32_t), hipMemcpyHostToDevice, st));
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_ilu0_count, dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, lrow, urow, missing);

    L->n = U->n = n; L->is_csr = U->is_csr = true; L->owns = U->owns = true;
    ILUPP_HIP(hipMalloc(&L->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(hipMalloc(&U->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(hipMemsetAsync(L->ptr, 0, sizeof(int32_t), st));
    ILUPP_HIP(hipMemsetAsync(U->ptr, 0, sizeof(int32_t), st));
    size_t tmp_bytes = 0;
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, lrow, L->ptr + 1, n, st));
    void *tmp = nullptr;
    ILUPP_HIP(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, lrow, L->ptr + 1, n, st));
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, urow, U->ptr + 1, n, st));
    int32_t tot[2], miss;
    ILUPP_HIP(hipMemcpyAsync(&tot[0], L->ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipMemcpyAsync(&tot[1], U->ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipMemcpyAsync(&miss, missing, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    ILUPP_HIP(hipFree(tmp)); ILUPP_HIP(hipFree(lrow)); ILUPP_HIP(hipFree(urow)); ILUPP_HIP(hipFree(missing));
    if (first_missing_diag) *first_missing_diag = (miss == big) ? -1 : miss;
    L->nnz = tot[0]; U->nnz = tot[1];
    ILUPP_HIP(hipMalloc(&L->idx, sizeof(int32_t) * (size_t)(L->nnz > 0 ? L->nnz : 1)));
    ILUPP_HIP(hipMalloc(&L->val, sizeof(double) * (size_t)(L->nnz > 0 ? L->nnz : 1)));
    ILUPP_HIP(hipMalloc(&U->idx, sizeof(int32_t) * (size_t)(U->nnz > 0 ? U->nnz : 1)));
    ILUPP_HIP(hipMalloc(&U->val, sizeof(double) * (size_t)(U->nnz > 0 ? U->nnz : 1)));
    if (miss != big) return ILUPP_ERR_NO_DIAGONAL;
    hipLaunchKernelGGL(k_ilu0_pattern, dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, L->ptr, U->ptr, L->idx, L->val, U->idx);
    return ILUPP_OK;
}

// ---------------------------------------------------------------------------------------------
// numeric: persistent dataflow kernel
// ---------------------------------------------------------------------------------------------
static constexpr unsigned kSpinLimit = 1u << 22;

// ctrl[0] = workgroup ticket, ctrl[1] = error word
template <int MAXLEN, bool GLOBAL_W>
__global__ void __launch_bounds__(kThreads)
k_ilu0_numeric(const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval,
               const int32_t *__restrict__ Lptr, double *__restrict__ Lval,
               const int32_t *__restrict__ Uptr, const int32_t *__restrict__ Uidx, double *Uval,
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
    int a0 = 0, len = 0, cl = 0, p = 0;
    unsigned spins = 0;

    for (;;) {
        if (!__any(active)) break;
        bool progressed = false;
        if (active) {
            if (need_init) {
                a0 = Aptr[r];
                len = Aptr[r + 1] - a0;
                cl = Lptr[r + 1] - Lptr[r] - 1;
                for (int q = 0; q < len; ++q) W(q) = Aval[a0 + q];     // U[i,:] = A[i,:]   (ILU0.hpp:36-37)
                p = 0;
                need_init = false;
                progressed = true;
            }
            while (p < cl) {                                           // for k < i in row  (ILU0.hpp:47-62)
                const int k = Aidx[a0 + p];
                if (ld_agent_i32(&done[k]) == 0) break;                // row k not finished: retry next round
                order_after_poll();
                const int u0 = Uptr[k], u1 = Uptr[k + 1];
                const double piv = ld_agent_f64(&Uval[u0]);            // diag_U[k]: first entry of U row k
                const double l_ik = W(p) / piv;                        // ILU0.hpp:52
                int pp = p + 1;
                for (int j = u0 + 1; j < u1; ++j) {                    // sparse_vec_update (ILU0.hpp:8-23)
                    const int m = Uidx[j];
                    while (pp < len && Aidx[a0 + pp] < m) ++pp;
                    if (pp >= len) break;
                    if (Aidx[a0 + pp] == m) {
                        const double u_kj = ld_agent_f64(&Uval[j]);
                        const double prod = l_ik * u_kj;
                        W(pp) = W(pp) - prod;
                        ++pp;
                    }
                }
                W(p) = l_ik;                                           // ILU0.hpp:61
                ++p;
                progressed = true;
            }
            if (p == cl) {
                // split (ILU0.hpp:85-98): L gets the multipliers (unit diagonal written by the symbolic
                // pass), U the rest, diagonal first.  U is published write-through for other CUs.
                const int l0 = Lptr[r];
                for (int q = 0; q < cl; ++q) Lval[l0 + q] = W(q);
                const int ub = Uptr[r];
                for (int q = cl; q < len; ++q) st_agent_f64(&Uval[ub + q - cl], W(q));
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
            if (++spins > kSpinLimit) {
                if ((tid & 63) == 0) atomicExch(&ctrl[1], 1);
                break;
            }
        }
    }
#undef W
}

int ilu0_numeric(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const Schedule &fwd,
                 int32_t max_row_len, int32_t *d_done, int32_t *d_ctrl, float *kernel_ms)
{
    const int32_t n = A.n;
    ILUPP_HIP(hipMemsetAsync(d_done, 0, sizeof(int32_t) * (size_t)n, st));
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    const unsigned grid = (unsigned)((fwd.nb + kThreads - 1) / kThreads);
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    double *wscratch = nullptr;
    ILUPP_HIP(hipEventRecord(e0, st));
#define LAUNCH(ML, GW, LDSB)                                                                              \
    hipLaunchKernelGGL((k_ilu0_numeric<ML, GW>), dim3(grid), dim3(kThreads), (LDSB), st, A.ptr, A.idx, A.val, \
                       L->ptr, L->val, U->ptr, U->idx, U->val, fwd.nb, fwd.start, d_done, d_ctrl, wscratch, max_row_len)
    if (max_row_len <= 8) LAUNCH(8, false, 8 * kThreads * sizeof(double));
    else if (max_row_len <= 16) LAUNCH(16, false, 16 * kThreads * sizeof(double));
    else if (max_row_len <= 32) LAUNCH(32, false, 32 * kThreads * sizeof(double));
    else if (max_row_len <= 64) LAUNCH(64, false, 64 * kThreads * sizeof(double));
    else {
        ILUPP_HIP(hipMalloc(&wscratch, sizeof(double) * (size_t)grid * kThreads * (size_t)max_row_len));
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
    if (wscratch) ILUPP_HIP(hipFree(wscratch));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

}  // namespace ilupp
