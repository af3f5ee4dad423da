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
        for (int i = 0; i < cnt; ++i) { const double ti = __shfl(t, i); acc = acc + ti; }
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
