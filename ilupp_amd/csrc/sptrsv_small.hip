// ilupp_amd/csrc/sptrsv_small.hip -- triangular sweeps of SMALL factors (n <= kSmallSweepMax) by ONE workgroup with the unknowns in LDS.
//
// The levels of a multilevel preconditioner get smaller and denser from level to level (Schur complements fill); the sweeps of a dense
// triangular factor are one dependency chain of n rows, and the general kernels pay a trip through memory per link of that chain (2-4 us:
// write-through store, cache-bypassing poll) -- 8 ms per sweep at n = 2 000.  Here the chain runs through LDS: one lane per row (rows t,
// t + 256, ... per lane), the unknowns in an LDS array that starts all-sentinel (the data is the flag, as everywhere in this library), a lane
// consumes its row's entries strictly in stored order as the unknowns they need appear -- the reference's arithmetic (triangular_solve,
// sparse_implementation.h:4040-4087: x_k -= d_j x[idx_j] one entry after the other, then the division by the diagonal), hence its bits.
// No lane ever blocks: every trip of the loop each lane either consumes up to eight entries (those of its register group whose unknowns
// are there, in order), finishes a row, or does nothing, so lanes of one wave can wait for each other.  Entries are fetched eight at a
// time into registers, the next eight while the current eight are consumed.
#include "common.h"

namespace ilupp {

static constexpr int kSmallThreads = 256;      // one wave per SIMD: the chain of a dense factor advances one row per two trips of the loop, and a trip is as long as the waves that share a SIMD make it

// kind SWEEP_FWD_LAST_ASC: lower CSR, diagonal LAST in the row (T1);  SWEEP_BWD_FIRST_ASC: upper CSR, diagonal FIRST (T3)
static constexpr int kG = 8;                    // entries of a row held in one register group

template <bool FWD>
__global__ void __launch_bounds__(kSmallThreads)
k_sptrsv_small(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
               const double *__restrict__ rhs, double *__restrict__ out, int32_t *__restrict__ err)
{
    extern __shared__ unsigned long long xs[];      // the unknowns, sentinel = not yet
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += kSmallThreads) xs[i] = kSentinel;
    __syncthreads();
    int r = FWD ? tid : n - 1 - tid;                // this lane's current row
    bool alive = FWD ? r < n : r >= 0;
    int j = 0, jend = 0, jf = 0;                    // the off-diagonal entries of the row still to consume: [j, jend); fetched up to jf
    double acc = 0.0, diag = 1.0;
    // two groups of four entries in registers: one is consumed while the other one's loads are in flight
    int c0[kG] = {0}, c1[kG] = {0};
    double v0[kG] = {0.0}, v1[kG] = {0.0};
    int have0 = 0, have1 = 0, at = 0;
    auto fetch1 = [&]() {                           // the next group into c1 / v1
        have1 = jend - jf < kG ? jend - jf : kG;
#pragma unroll
        for (int q = 0; q < kG; ++q) if (q < have1) { c1[q] = idx[jf + q]; v1[q] = val[jf + q]; }
        jf += have1;
    };
    // Every trip of the loop below some lane of the WORKGROUP consumes an entry or finishes a row, unless an unknown never leaves the
    // sentinel (an index outside the triangle, a right-hand side that carries the sentinel's bits).  A wave gives up -- ILUPP_ERR_TIMEOUT,
    // as everywhere else in this library -- only when NO wave of the workgroup has made progress for kIdleLimit of its own trips: the
    // trips a wave spends waiting for rows of other waves do not count against it (a chain-like factor makes a wave wait for a long time),
    // and whichever wave gives up reports it.
    constexpr unsigned kIdleLimit = 1u << 20;
    __shared__ unsigned s_progress;
    if (tid == 0) s_progress = 0;
    __syncthreads();
    unsigned seen = 0, idle = 0;
    bool broken = false;
    auto open_row = [&]() {
        const int b = ptr[r], e = ptr[r + 1];
        if (e <= b) { broken = true; return; }      // (a row without its diagonal: nothing to divide by)
        if (FWD) { j = b; jend = e - 1; diag = val[e - 1]; }
        else { j = b + 1; jend = e; diag = val[b]; }
        acc = rhs[r];
        jf = j;
        have0 = jend - jf < kG ? jend - jf : kG;
#pragma unroll
        for (int q = 0; q < kG; ++q) if (q < have0) { c0[q] = idx[jf + q]; v0[q] = val[jf + q]; }
        jf += have0;
        at = 0;
        fetch1();
    };
    if (alive) open_row();
    while (__ballot(alive) != 0ull) {
        if (__ballot(broken) != 0ull || idle > kIdleLimit) { if ((tid & 63) == 0 && err) atomicExch(err, 1); break; }
        bool did = false;
        if (alive) {
            if (j < jend) {
                if (at == have0) {                  // the group is used up: the other one takes its place, the one after it is asked for
#pragma unroll
                    for (int q = 0; q < kG; ++q) { c0[q] = c1[q]; v0[q] = v1[q]; }
                    have0 = have1; at = 0;
                    fetch1();
                }
                // as many entries of the group as have their unknown, in stored order (a backward sweep's row waits for its FIRST entry --
                // the unknown next to the diagonal, the last one to appear -- and then finds all the others there: one trip per entry made the
                // backward sweep of a dense factor 70 times slower than the forward one)
                unsigned long long xb[kG];
#pragma unroll
                for (int q = 0; q < kG; ++q)
                    xb[q] = (q >= at && q < have0) ? __hip_atomic_load(&xs[c0[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : kSentinel;
#pragma unroll
                for (int q = 0; q < kG; ++q)
                    if (q == at && q < have0 && xb[q] != kSentinel) {
                        const double p = v0[q] * __longlong_as_double((long long)xb[q]);
                        acc = acc - p;
                        ++at; ++j;
                        did = true;
                    }
            } else {
                const double x = acc / diag;
                __hip_atomic_store(&xs[r], (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                out[r] = x;
                r += FWD ? kSmallThreads : -kSmallThreads;
                alive = FWD ? r < n : r >= 0;
                if (alive) open_row();
                did = true;
            }
        }
        if (__ballot(did) != 0ull) {
            if ((tid & 63) == 0) atomicAdd(&s_progress, 1u);
            idle = 0;
        } else {
            const unsigned now = __hip_atomic_load(&s_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (now != seen) { seen = now; idle = 0; } else ++idle;
        }
    }
}

// out = T^-1 rhs for a small triangular factor in the storage a gather sweep wants; rhs is left as it is
int sptrsv_small(hipStream_t st, SweepKind kind, const DevMat &M, const double *rhs, double *out, int32_t *err)
{
    const size_t lds = sizeof(unsigned long long) * (size_t)M.n;
    if (kind == SWEEP_FWD_LAST_ASC) {
        static thread_local int attr_dev = -1;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (attr_dev != dev) {
            ILUPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sptrsv_small<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSweepMax * 8));
            ILUPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sptrsv_small<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSweepMax * 8));
            attr_dev = dev;
        }
        hipLaunchKernelGGL(k_sptrsv_small<true>, dim3(1), dim3(kSmallThreads), lds, st, M.n, M.ptr, M.idx, M.val, rhs, out, err);
    } else if (kind == SWEEP_BWD_FIRST_ASC) {
        static thread_local int attr_dev2 = -1;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (attr_dev2 != dev) {
            ILUPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sptrsv_small<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSweepMax * 8));
            ILUPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sptrsv_small<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSweepMax * 8));
            attr_dev2 = dev;
        }
        hipLaunchKernelGGL(k_sptrsv_small<false>, dim3(1), dim3(kSmallThreads), lds, st, M.n, M.ptr, M.idx, M.val, rhs, out, err);
    } else {
        return ILUPP_ERR_UNSUPPORTED;
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

}  // namespace ilupp
