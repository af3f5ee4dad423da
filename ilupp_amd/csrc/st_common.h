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

// st_direct.hip
bool st_direct_prepare(hipStream_t st, const DevMat &A, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu, int32_t *flags_out_dev);
int ilu0_numeric_sd(hipStream_t st, const DevMat &A, PackedSweep *pl, PackedSweep *pu, int32_t *d_ctrl, float *kernel_ms,
                    hipEvent_t e0, hipEvent_t e1);

}  // namespace ilupp
