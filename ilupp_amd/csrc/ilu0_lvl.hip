// ilupp_amd/csrc/ilu0_lvl.hip -- ILU(0) numeric phase with one WAVE per row, rows in level order (gfx950 only).
//
// For matrices whose rows are too long for the level-major forms (9-point, 27-point stencils, general matrices with up to 64
// entries per row).  The CSR-streaming kernels give every lane a block of consecutive rows and walk a row's eliminations one
// after the other, each with its own chain of dependent loads (pointer -> indices -> values): 20 us per row of a mesh line,
// 122 ms for a 2048 x 2048 9-point matrix.  Here
//   * the rows are taken in the order of their dependency level (sptrsv_lvl.hip: level(i) = 1 + max level(k), k < i in row i;
//     the same order later serves the forward sweep), so that a row's pivot rows belong to earlier tickets and have usually
//     finished when the row starts;
//   * a wave owns one row: lane q holds the q-th stored entry (column, working value) in registers;
//   * the U rows of up to 16 pivots are fetched TOGETHER, lane e taking entry e of each; U's value array starts as all-sentinel
//     and is written with write-through stores, so this fetch is also the wait (data-is-flag: no flag array, no second trip);
//   * where entry e of pivot row k lands in row i is a binary search over the lanes' columns with wave shuffles; it only needs the
//     patterns and runs before the values have arrived;
//   * the eliminations themselves then cost two broadcasts, a division, a multiplication and a subtraction each, in ascending
//     k as in the reference (ILU0.hpp:47-62: l_ik = w_k / u_kk; w_j -= l_ik u_kj for the j of row i's pattern), without fused
//     multiply-add: same bits.
#include "common.h"

namespace ilupp {

static constexpr int kFB = 16;                       // pivots whose U rows are fetched together
static constexpr unsigned kFlSpinLimit = 1u << 22;

__device__ __forceinline__ double bcast_f64(double x, int lane)       // lane: wave-uniform
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}

__global__ void __launch_bounds__(kThreads)
k_ilu0_lvl(int32_t n, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval,
           const int32_t *__restrict__ Lptr, double *__restrict__ Lval, const int32_t *__restrict__ Uptr,
           const int32_t *__restrict__ Uidx, double *Uval, const int32_t *__restrict__ perm, int32_t *ctrl)
{
    __shared__ unsigned wg_ticket;
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(&ctrl[0], 1);
    __syncthreads();
    const int ln = threadIdx.x & 63;
    const int64_t t = (int64_t)wg_ticket * (kThreads / 64) + (threadIdx.x >> 6);
    if (t >= n) return;
    const int i = perm[t];
    const int a0 = Aptr[i], len = Aptr[i + 1] - a0;
    const bool have = ln < len;
    const int col = have ? Aidx[a0 + ln] : 0x7fffffff;
    double w = have ? Aval[a0 + ln] : 0.0;                              // U[i,:] = A[i,:]   (ILU0.hpp:36-37)
    const int cl = __popcll(__ballot(have && col < i));                 // entries left of the diagonal = pivots, ascending
    int u0 = 0, ul = 0;                                                 // lane p < cl: extent of the U row of pivot p
    if (ln < cl) { u0 = Uptr[col]; ul = Uptr[col + 1] - u0; }
    const unsigned long long *Uvb = reinterpret_cast<const unsigned long long *>(Uval);

    for (int pb = 0; pb < cl; pb += kFB) {
        int um[kFB], epos[kFB], s0[kFB], sl[kFB];
        unsigned long long ub[kFB];
#pragma unroll
        for (int u = 0; u < kFB; ++u) {
            const int p = pb + u;
            s0[u] = p < cl ? __builtin_amdgcn_readlane(u0, p < 64 ? p : 0) : 0;
            sl[u] = p < cl ? __builtin_amdgcn_readlane(ul, p < 64 ? p : 0) : 0;
            um[u] = ln < sl[u] ? Uidx[s0[u] + ln] : -1;                 // entry 0 is the pivot itself
            ub[u] = kSentinel;
        }
        // where the entries of each pivot row meet this row: lane q looks for its column among the pivot row's columns
#pragma unroll
        for (int u = 0; u < kFB; ++u) {
            epos[u] = -1;
            if (pb + u < cl) {
                int lo = 1, hi = sl[u];
                while (__ballot(lo < hi) != 0ull) {
                    const int mid = (lo + hi) >> 1;
                    const int mv = __shfl(um[u], mid & 63);
                    if (lo < hi) { if (mv < col) lo = mid + 1; else hi = mid; }
                }
                const int fv = __shfl(um[u], lo & 63);
                epos[u] = (have && ln > pb + u && lo < sl[u] && fv == col) ? lo : -1;
            }
        }
        // the values: all of them are there = the pivot rows are finished
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < kFB; ++u) {
                if (ln < sl[u] && ub[u] == kSentinel) ub[u] = ld_agent_u64(Uvb + s0[u] + ln);
                ok = ok && !(ln < sl[u] && ub[u] == kSentinel);
            }
            if (__ballot(!ok) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kFlSpinLimit) {
                if (ln == 0) atomicExch(&ctrl[1], 1);
                return;
            }
        }
#pragma unroll
        for (int u = 0; u < kFB; ++u) {
            const int p = pb + u;
            if (p < cl) {
                const double uv = __longlong_as_double((long long)ub[u]);
                const double piv = bcast_f64(uv, 0);
                const double wp = bcast_f64(w, p < 64 ? p : 0);
                const double l_ik = wp / piv;                           // ILU0.hpp:52
                const double u_kj = __shfl(uv, epos[u] & 63);
                const double prod = l_ik * u_kj;                        // sparse_vec_update (ILU0.hpp:8-23)
                const double nw = w - prod;
                w = epos[u] >= 0 ? nw : w;
                if (ln == p) w = l_ik;                                  // ILU0.hpp:61
            }
        }
    }
    // split (ILU0.hpp:85-98): the multipliers to L (its unit diagonal is already there), the rest to U, diagonal first,
    // write-through: every value is its own flag
    if (ln < cl) {
        Lval[Lptr[i] + ln] = w;
    } else if (have) {
        unsigned long long b = (unsigned long long)__double_as_longlong(w);
        if (b == kSentinel) b = kCanonNaN;
        st_agent_u64(reinterpret_cast<unsigned long long *>(Uval) + Uptr[i] + (ln - cl), b);
    }
}

int ilu0_numeric_lvl(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const int32_t *perm, int32_t *d_ctrl, float *kernel_ms)
{
    const int32_t n = A.n;
    fill_u64(st, reinterpret_cast<unsigned long long *>(U->val), U->nnz, kSentinel);
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    ILUPP_HIP(hipEventRecord(e0, st));
    const int rows_per_wg = kThreads / 64;
    hipLaunchKernelGGL(k_ilu0_lvl, dim3((unsigned)((n + rows_per_wg - 1) / rows_per_wg)), dim3(kThreads), 0, st, n, A.ptr, A.idx, A.val,
                       L->ptr, L->val, U->ptr, U->idx, U->val, perm, d_ctrl);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4] = {0, 0, 0, 0};
    ILUPP_HIP(hipMemcpyAsync(ctrl, d_ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    ILUPP_HIP(hipEventDestroy(e0));
    ILUPP_HIP(hipEventDestroy(e1));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

}  // namespace ilupp
