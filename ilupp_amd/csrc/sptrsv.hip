// ilupp_amd/csrc/sptrsv.hip -- sparse triangular solves of apply() for gfx950.
//
// Replaces matrix_sparse::triangular_solve (reference sparse_implementation.h:4040-4087).  The four
// loops of the reference reduce to three gather sweeps over row-major storage (the two scatter loops
// T2/T4 perform, for every unknown, the same subtractions in the same order as a gather over the
// transposed storage; see DESIGN.md "Solve variants"):
//
//   SWEEP_FWD_LAST_ASC    rows ascending,  x_r = (x_r - sum_{j<last} v_j x[c_j]) / v_last     (T1, T2')
//   SWEEP_BWD_FIRST_ASC   rows descending, x_r = (x_r - sum_{j>first} v_j x[c_j]) / v_first   (T3)
//   SWEEP_BWD_FIRST_DESC  the same with the off-diagonal entries taken last-to-first          (T4')
//
// The accumulation is sequential in stored order with separate multiply and subtract, and the
// diagonal is found by POSITION and always divided by (unit diagonals too), exactly as the
// reference does, so solve vectors are bit-identical to the CPU.
//
// One persistent launch per sweep; hand-off between lanes is "the data is the flag": the output
// vector starts as all-sentinel (a NaN payload no arithmetic produces), a finished x_r is stored
// with one 8-byte write-through (sc1) store, consumers poll x[c] with sc1 loads until it is not the
// sentinel.  No flag array, no fences, no grid barrier.  The right-hand side is read once per row
// and immediately overwritten with the sentinel, which makes that buffer the ready-made output of
// the next sweep: L-solve  x -> y,  U-solve  y -> x  leaves the result in place in x and y reset.
#include "common.h"

namespace ilupp {

static constexpr unsigned kSolveSpinLimit = 1u << 22;

template <int KIND>
__global__ void __launch_bounds__(kThreads)
k_sptrsv(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
         double *rhs, double *out, int32_t nb, const int32_t *__restrict__ bstart, int32_t *ticket, int32_t *err)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr bool DESC = (KIND == SWEEP_BWD_FIRST_DESC);
    __shared__ unsigned wg_ticket;
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(ticket, 1);
    __syncthreads();
    const int tid = threadIdx.x;
    const int64_t slot = (int64_t)wg_ticket * kThreads + tid;

    // rows of this lane in sweep order: r, r+dir, ..., until r == rstop
    int r = 0, rstop = 0;
    if (slot < nb) {
        const int b = FWD ? (int)slot : (int)(nb - 1 - slot);
        const int lo = bstart[b], hi = bstart[b + 1];
        if (FWD) { r = lo; rstop = hi; } else { r = hi - 1; rstop = lo - 1; }
    }
    constexpr int dir = FWD ? 1 : -1;
    bool active = (r != rstop);
    bool need_init = true;
    int j = 0, jend = 0, dpos = 0;
    double acc = 0.0, prev_val = 0.0;
    int prev_row = -1;
    unsigned spins = 0;
    const unsigned long long *outb = reinterpret_cast<const unsigned long long *>(out);

    for (;;) {
        if (!__any(active)) break;
        bool progressed = false;
        if (active) {
            if (need_init) {
                const int lo = ptr[r], hi = ptr[r + 1];
                acc = rhs[r];
                reinterpret_cast<unsigned long long *>(rhs)[r] = kSentinel;   // this buffer is the next sweep's output
                if (FWD)       { j = lo;     jend = hi - 1; dpos = hi - 1; }
                else if (!DESC){ j = lo + 1; jend = hi;     dpos = lo; }
                else           { j = hi - 1; jend = lo;     dpos = lo; }
                need_init = false;
                progressed = true;
            }
            while (j != jend) {
                const int c = idx[j];
                double xc;
                if (c == prev_row) {
                    xc = prev_val;                      // own previous row: never leaves the lane
                } else {
                    const unsigned long long bits = ld_agent_u64(outb + c);
                    if (bits == kSentinel) break;       // x[c] not there yet: retry next round
                    xc = __longlong_as_double((long long)bits);
                }
                const double prod = val[j] * xc;
                acc = acc - prod;                       // x[k] -= data[j]*x[indices[j]]  (:4049, :4070)
                j += DESC ? -1 : 1;
                progressed = true;
            }
            if (j == jend) {
                acc = acc / val[dpos];                  // x[k] /= diagonal (by position)  (:4051, :4072)
                if (acc != acc) acc = __longlong_as_double((long long)kCanonNaN);   // never store the sentinel
                st_agent_f64(out + r, acc);
                prev_row = r;
                prev_val = acc;
                r += dir;
                need_init = true;
                active = (r != rstop);
                progressed = true;
            }
        }
        if (__any(progressed)) {
            spins = 0;
        } else {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSolveSpinLimit) {
                if ((tid & 63) == 0) atomicExch(err, 1);
                break;
            }
        }
    }
}

int sptrsv(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, double *rhs_and_reset,
           double *out, int32_t *d_ticket, int32_t *d_err)
{
    // *d_ticket must be zero on entry (the caller zeroes the whole control block once per apply)
    const unsigned grid = (unsigned)((sch.nb + kThreads - 1) / kThreads);
    switch (kind) {
    case SWEEP_FWD_LAST_ASC:
        hipLaunchKernelGGL((k_sptrsv<SWEEP_FWD_LAST_ASC>), dim3(grid), dim3(kThreads), 0, st,
                           M.ptr, M.idx, M.val, rhs_and_reset, out, sch.nb, sch.start, d_ticket, d_err);
        break;
    case SWEEP_BWD_FIRST_ASC:
        hipLaunchKernelGGL((k_sptrsv<SWEEP_BWD_FIRST_ASC>), dim3(grid), dim3(kThreads), 0, st,
                           M.ptr, M.idx, M.val, rhs_and_reset, out, sch.nb, sch.start, d_ticket, d_err);
        break;
    default:
        hipLaunchKernelGGL((k_sptrsv<SWEEP_BWD_FIRST_DESC>), dim3(grid), dim3(kThreads), 0, st,
                           M.ptr, M.idx, M.val, rhs_and_reset, out, sch.nb, sch.start, d_ticket, d_err);
        break;
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

}  // namespace ilupp
