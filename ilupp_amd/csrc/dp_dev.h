// ilupp_amd/csrc/dp_dev.h -- device helpers of the ONE-WAVE kernels that walk a chain of data-dependent steps (pilucdp.hip: the multilevel
// factorisation with pivoting; ilucp.hip: ILUCP): the working vector with its insertion order, ordered sums, in-register sort, subtraction
// of a stored row with new indices appended in entry order, list nodes read ahead.
#pragma once
#include "common.h"

namespace ilupp {

// an entry of a working vector, by index: its value and its slot in the insertion-ordered list (-1: not in the vector; -2: dead -- a column
// that has been a pivot / a row that has been eliminated: never touched again).  One 16-byte load answers "may I?", "where?" and "how much?".
struct __attribute__((aligned(16))) DpRec { double val; int32_t slot; int32_t pad; };

struct SpVec { DpRec *rec; int32_t *list; };      // by index: value and slot; slot -> index (insertion order)

#define DP_SYNC() do { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); } while (0)

// ---- cross-lane operations of one wave without the LDS crossbar.  __shfl / __shfl_xor compile to ds_bpermute (an LDS-pipe round trip,
// >100 cycles, and these kernels use them in dependent sequences: an ordered sum of 20 terms, a 6-step arg-max, a 21-stage bitonic
// sort were 1-2 us each).  A value of a lane whose number is wave-uniform is a v_readlane (SGPR result, a few cycles); reductions run
// on DPP row shifts / broadcasts.
__device__ __forceinline__ int wv_i32(int x, int i) { return __builtin_amdgcn_readlane(x, i); }
__device__ __forceinline__ unsigned long long wv_u64(unsigned long long x, int i)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, i), hi = (unsigned)__builtin_amdgcn_readlane((int)(x >> 32), i);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double wv_f64(double x, int i) { return __longlong_as_double((long long)wv_u64((unsigned long long)__double_as_longlong(x), i)); }

template <int CTRL, int ROWS> __device__ __forceinline__ unsigned long long wv_dpp_u64(unsigned long long x)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)x, (int)(unsigned)x, CTRL, ROWS, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(x >> 32), (int)(x >> 32), CTRL, ROWS, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}
// the largest of 64 unsigned keys, in every lane (row_shr 1, 2, 4, 8, then row_bcast:15 / :31: lane 63 holds the result)
__device__ __forceinline__ unsigned long long wv_max_u64(unsigned long long x)
{
    unsigned long long y;
    y = wv_dpp_u64<0x111, 0xf>(x); x = y > x ? y : x;
    y = wv_dpp_u64<0x112, 0xf>(x); x = y > x ? y : x;
    y = wv_dpp_u64<0x114, 0xf>(x); x = y > x ? y : x;
    y = wv_dpp_u64<0x118, 0xf>(x); x = y > x ? y : x;
    y = wv_dpp_u64<0x142, 0xa>(x); x = y > x ? y : x;
    y = wv_dpp_u64<0x143, 0xc>(x); x = y > x ? y : x;
    return wv_u64(x, 63);
}
__device__ __forceinline__ int wv_min_i32(int x)
{
    int y;
    y = __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false); x = y < x ? y : x;
    y = __builtin_amdgcn_update_dpp(x, x, 0x112, 0xf, 0xf, false); x = y < x ? y : x;
    y = __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false); x = y < x ? y : x;
    y = __builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xf, false); x = y < x ? y : x;
    y = __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false); x = y < x ? y : x;
    y = __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false); x = y < x ? y : x;
    return wv_i32(x, 63);
}
// the largest of 64 doubles none of which is a NaN (their bit patterns mapped to unsigned integers of the same order)
__device__ __forceinline__ unsigned long long wv_order_bits(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double wv_max_f64(double x)
{
    const unsigned long long m = wv_max_u64(wv_order_bits(x));
    return __longlong_as_double((long long)((m >> 63) ? (m & 0x7fffffffffffffffull) : ~m));
}
// (value, position) pairs, position 0x7fffffff = "this lane has none": the position of the largest value, the smallest position among
// equal values; 0x7fffffff if no lane has a pair.  Every lane gets the winner's value in `mx`.  (A NaN among the values -- which the
// scalar comparisons treat by their own rules -- is left to the shuffle form of the same reduction.)
__device__ __forceinline__ int wv_argmax_first(double &mx, int pos)
{
    if (__ballot(pos != 0x7fffffff && mx != mx)) {
        for (int o = 32; o > 0; o >>= 1) {
            const double om = __shfl_xor(mx, o);
            const int op = __shfl_xor(pos, o);
            if (op != 0x7fffffff && (pos == 0x7fffffff || om > mx || (om == mx && op < pos))) { mx = om; pos = op; }
        }
        return pos;
    }
    const unsigned long long key = pos == 0x7fffffff ? 0ull : wv_order_bits(mx);
    const unsigned long long m = wv_max_u64(key);
    if (m == 0ull) return 0x7fffffff;
    const int p = wv_min_i32(key == m ? pos : 0x7fffffff);
    mx = __longlong_as_double((long long)((m >> 63) ? (m & 0x7fffffffffffffffull) : ~m));
    return p;
}
// n <= 64 DISTINCT keys in the lanes 0 .. n-1: the key of rank `lane` (ascending) for the lanes below n.  Every lane counts the keys
// below its own (n uniform reads), then sends its key to the lane of that rank.
__device__ __forceinline__ unsigned long long dp_sort_n(unsigned long long key, int n, int lane)
{
    int rank = 0;
    for (int i = 0; i < n; ++i) rank += wv_u64(key, i) < key ? 1 : 0;
    if (lane >= n) rank = lane;
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)(unsigned)key), hi = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)(key >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// sum of |x| (mode 0) or x * x (mode 1) over the slots IN ORDER (vector_sparse_dynamic::norm1 / norm2, sparse_implementation.h:1074-1085):
// 64 values per pass, one per lane, added one after the other
__device__ inline double dp_seq_sum(const SpVec &v, int nnz, int mode, int lane)
{
    double acc = 0.0;
    for (int base = 0; base < nnz; base += 64) {
        const int s = base + lane;
        const double x = s < nnz ? v.rec[v.list[s]].val : 0.0;
        const double t = mode == 0 ? fabs(x) : x * x;
        const int cnt = nnz - base < 64 ? nnz - base : 64;
        for (int i = 0; i < cnt; ++i) acc = acc + wv_f64(t, i);
    }
    return acc;
}

// bitonic sort of 64 keys, one per lane, ascending, in registers (rows of a sparse factor rarely keep more than 64 entries: the sort in
// memory below takes 21 passes with a barrier each)
__device__ __forceinline__ unsigned long long dp_sort64(unsigned long long key, int lane)
{
    for (int k2 = 2; k2 <= 64; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned long long other = __shfl_xor(key, j);
            const bool up = (lane & k2) == 0, lower = (lane & j) == 0;
            const bool take_min = lower == up;
            key = take_min ? (key < other ? key : other) : (key > other ? key : other);
        }
    return key;
}

// v[idx] exists afterwards (operator[] inserts a zero, sparse_implementation.h:980-994)
__device__ __forceinline__ void dp_touch(const SpVec &v, int &nnz, int idx, int lane)
{
    if (v.rec[idx].slot < 0) {
        if (lane == 0) { v.list[nnz] = idx; v.rec[idx] = DpRec{0.0, nnz, 0}; }
        ++nnz;
        DP_SYNC();
    }
}

// v -= f * (the entries e0 .. e1 of a stored row / column) where the entry's index is not dead, new indices appended in entry order;
// this lane's entry of the first 64 (index c0, value v0) has been fetched ahead
__device__ __forceinline__ void dp_subtract(const SpVec &v, int &nnz, double f, const int32_t *idx, const double *val, int e0, int e1, int c0, double v0, int lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = e0; base < e1; base += 64) {
        const int e = base + lane;
        const bool act = e < e1;
        const int c = base == e0 ? c0 : (act ? idx[e] : 0);
        const double ev = base == e0 ? v0 : (act ? val[e] : 0.0);
        DpRec r{0.0, -2, 0};
        if (act) r = v.rec[c];
        const bool ok = act && r.slot != -2;
        const bool isnew = ok && r.slot < 0;
        const unsigned long long mask = __ballot(isnew);
        if (ok) {
            const double prod = f * ev;
            if (isnew) { const int s = nnz + __popcll(mask & lt); v.list[s] = c; v.rec[c] = DpRec{0.0 - prod, s, 0}; }
            else v.rec[c].val = r.val - prod;
        }
        nnz += __popcll(mask);
    }
    DP_SYNC();
}

// a node of a row's / column's list (the entry `at` of the other factor's store): which row / column it belongs to, its value, the next node
struct DpNode { int at, who, link; double v; };
struct DpRow { double dinv; int e0, e1; };
struct DpEnt { int c; double v; };
__device__ __forceinline__ DpNode dp_node(const int32_t *who, const double *val, const int32_t *link, int at)
{
    DpNode nd{at, 0, -1, 0.0};
    if (at != -1) { nd.who = who[at]; nd.v = val[at]; nd.link = link[at]; }
    return nd;
}
__device__ __forceinline__ DpRow dp_row(const double *Dinv, const int32_t *ptr, const DpNode &nd)
{
    DpRow r{1.0, 0, 0};
    if (nd.at != -1) { r.dinv = Dinv[nd.who]; r.e0 = ptr[nd.who]; r.e1 = ptr[nd.who + 1]; }
    return r;
}

__device__ __forceinline__ DpEnt dp_ent(const int32_t *idx, const double *val, const DpRow &r, int lane)
{
    DpEnt t{0, 0.0};
    if (r.e0 + lane < r.e1) { t.c = idx[r.e0 + lane]; t.v = val[r.e0 + lane]; }
    return t;
}

}  // namespace ilupp
