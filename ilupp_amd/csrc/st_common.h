// ilupp_amd/csrc/st_common.h -- what the static level-major kernels (st.hip, st_direct.hip) share: value markers, the barrier of a
// step, the descriptor of a value that comes from an earlier workgroup.
#pragma once

#include "common.h"

namespace ilupp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define ST_STREAM_STORE(v, p) __builtin_nontemporal_store(v, p)
static constexpr int kStH = 8;                 // steps of hand-off history kept in LDS = steps the streams are read ahead
static constexpr int kStPF = 8, kStPS = 2;   // steps ahead the courier polls the values of earlier workgroups: factor kernel, sweeps (measured: +-4 %)
static constexpr int kStMaxSkew = 30000;
static constexpr unsigned kStSpinLimit = 1u << 21;
static constexpr int64_t kStMaxChunks = 1 << 21;      // record offsets are 32-bit byte offsets

struct __attribute__((aligned(8))) D2s { double v[2]; };

#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long st_bits(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double st_dbl(unsigned long long b) { return __longlong_as_double((long long)b); }
// a value that enters the records must not look like one of the two markers
__device__ __forceinline__ double st_clean(double x)
{
    const unsigned long long b = st_bits(x);
    return (b == kSentinel || b == kAbsent) ? st_dbl(kCanonNaN) : x;
}

// the barrier of a step: this wave's LDS writes of the previous step have landed, then everybody's have
// (NOT __syncthreads(): that would also drain the global loads in flight, i.e. the read-ahead)
#define ST_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ double st_lds(const unsigned char *base, unsigned off) { return *reinterpret_cast<const double *>(base + off); }
__device__ __forceinline__ int st_med3(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
#endif

// forward-lane fields of the lane table that only the direct-feed factor kernel (st_direct.hip) reads
enum { ST_P0 = 26, ST_DFL = 27, ST_Q = 28 };
// ST_DFL: entries right of the diagonal | own-chain entry left << 2 | own-chain entry right << 3 | entries per full row << 4
// ST_Q + j: which of the producer row's entries right of its diagonal is the transposed entry of dependency j (-1: none)

static constexpr int kSdHist = 4;     // hand-off slots of the direct-feed kernel: an in-workgroup dependency lies at most kSdHist-1 steps back

#if defined(__HIPCC__)
// the direct-feed kernel's lane fields, and its premise at lane level (dflags |= 1: a chain without its backward lane, |= 2: a
// lane whose entries are not produced where its template says for ALL of its rows, or whose own-chain entries are not r-1 / r+1)
__device__ __forceinline__ void sd_tab_lane(const int f, int32_t *__restrict__ ltabF, const int32_t *__restrict__ ltabB, const int32_t *__restrict__ uslot,
                                            const int32_t *__restrict__ Aptr, int32_t *__restrict__ dflags)
{
    int32_t *T = ltabF + (size_t)f * kStTab;
    const int cnt = T[ST_CNT], nd = T[ST_ND];
    T[ST_P0] = 0; T[ST_DFL] = 0; T[ST_Q] = -1; T[ST_Q + 1] = -1; T[ST_Q + 2] = -1;
    if (cnt <= 0) return;
    int bad = 0;
    const int su = uslot[f];
    if (su < 0) { atomicOr(dflags, 1); return; }
    const int32_t *TB = ltabB + (size_t)su * kStTab;
    const int ndU = TB[ST_ND];
    if (TB[ST_CNT] != cnt) bad = 1;
    int ownL = 0, ownU = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nd) {
            const int ty = T[ST_SRC + j] & 3;
            if (ty == ST_OWN) {
                // the own-chain entry: column r - 1, the last one left of the diagonal
                if (j != nd - 1 || T[ST_OFF + j] != -1) bad = 1;
                ownL = 1;
            } else {
                // every row of the lane has the entry, and its producer is where the template says
                if (T[ST_KLO + j] > 0 || T[ST_KHI + j] < cnt) bad = 1;
                if (ty == ST_LOCAL && (T[ST_DT + j] < 1 || T[ST_DT + j] > kSdHist - 1)) bad = 1;
            }
        }
        if (j < ndU) {
            const int ty = TB[ST_SRC + j] & 3;
            if (ty == ST_OWN) {
                if (j != 0 || TB[ST_OFF + j] != 1) bad = 1;
                ownU = 1;
            } else {
                if (TB[ST_KLO + j] > 0 || TB[ST_KHI + j] < cnt) bad = 1;
            }
        }
    }
    // the transposed entry of dependency j: the entry of the pivot row's right side whose offset is the opposite one
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int q = -1;
        if (j < nd) {
            const int os = T[ST_SRC + j] >> 2;
            const int pu = uslot[os];
            if (pu < 0) {
                bad = 1;
            } else {
                const int32_t *TP = ltabB + (size_t)pu * kStTab;
#pragma unroll
                for (int p = 0; p < 3; ++p) if (p < TP[ST_ND] && TP[ST_OFF + p] == -T[ST_OFF + j]) q = p;
            }
        }
        T[ST_Q + j] = q;
    }
    T[ST_P0] = Aptr[T[ST_FIRST]];
    T[ST_DFL] = ndU | (ownL << 2) | (ownU << 3) | ((nd + 1 + ndU) << 4);
    if (bad) atomicOr(dflags, 2);
}

#endif

// st_direct.hip
bool st_direct_prepare(hipStream_t st, const DevMat &A, const Schedule &fwd, int32_t *dflags);
void st_direct_verify(hipStream_t st, const DevMat &A, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu, int32_t *dflags);
int ilu0_numeric_sd(hipStream_t st, const DevMat &A, PackedSweep *pl, PackedSweep *pu, int32_t *d_ctrl, float *kernel_ms,
                    hipEvent_t e0, hipEvent_t e1);

}  // namespace ilupp
