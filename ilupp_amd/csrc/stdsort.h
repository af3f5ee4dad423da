// ilupp_amd/csrc/stdsort.h -- libstdc++ std::sort (introsort + final insertion sort, bits/stl_algo.h:1855-1957) restated for
// one GPU lane, comparator "larger magnitude first": the kept set of threshold_and_drop under equal magnitudes
// (reference dropping.hpp:25-26) is defined by this algorithm.  Included by ichol.hip and icholt_df.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace ilupp {

struct AbsDescC {
    const double *key;
    __device__ __forceinline__ bool operator()(int a, int b) const { return fabs(key[a]) > fabs(key[b]); }
};
// the same libstdc++ std::sort restatement as ilut.hip (kept local: separate translation units)
static __device__ void c_unguarded_linear_insert(int *last, const AbsDescC &c)
{
    const int v = *last;
    int *next = last - 1;
    while (c(v, *next)) { *last = *next; last = next; --next; }
    *last = v;
}
static __device__ void c_insertion_sort(int *first, int *last, const AbsDescC &c)
{
    if (first == last) return;
    for (int *i = first + 1; i != last; ++i) {
        if (c(*i, *first)) {
            const int v = *i;
            for (int *q = i; q != first; --q) *q = *(q - 1);
            *first = v;
        } else
            c_unguarded_linear_insert(i, c);
    }
}
static __device__ void c_push_heap(int *first, long hole, long top, int v, const AbsDescC &c)
{
    long parent = (hole - 1) / 2;
    while (hole > top && c(first[parent], v)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = v;
}
static __device__ void c_adjust_heap(int *first, long hole, long len, int v, const AbsDescC &c)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (c(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); first[hole] = first[child - 1]; hole = child - 1; }
    c_push_heap(first, hole, top, v, c);
}
static __device__ void c_heapsort(int *first, int *last, const AbsDescC &c)
{
    const long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) { const int v = first[parent]; c_adjust_heap(first, parent, len, v, c); if (parent == 0) break; parent--; }
    }
    while (last - first > 1) { --last; const int v = *last; *last = *first; c_adjust_heap(first, 0, last - first, v, c); }
}
static __device__ void c_sort_slots_by_abs_desc(int *list, int len, const double *key)
{
    AbsDescC c{key};
    if (len <= 0) return;
    int *first = list, *last = list + len;
    long lg = 0, m = len;
    while (m > 1) { m >>= 1; ++lg; }
    struct Frame { int *first, *last; long depth; };
    Frame stack[64];
    int sp = 0;
    stack[sp++] = Frame{first, last, 2 * lg};
    while (sp > 0) {
        Frame f = stack[--sp];
        int *fl = f.last;
        long depth = f.depth;
        while (fl - f.first > 16) {
            if (depth == 0) { c_heapsort(f.first, fl, c); break; }
            --depth;
            int *mid = f.first + (fl - f.first) / 2;
            int *a = f.first + 1, *b = mid, *cc = fl - 1, *res = f.first, *pick;
            if (c(*a, *b)) { if (c(*b, *cc)) pick = b; else if (c(*a, *cc)) pick = cc; else pick = a; }
            else if (c(*a, *cc)) pick = a;
            else if (c(*b, *cc)) pick = cc;
            else pick = b;
            { const int t = *res; *res = *pick; *pick = t; }
            int *lo = f.first + 1, *hi = fl;
            const int *pivot = f.first;
            for (;;) {
                while (c(*lo, *pivot)) ++lo;
                --hi;
                while (c(*pivot, *hi)) --hi;
                if (!(lo < hi)) break;
                const int t = *lo; *lo = *hi; *hi = t;
                ++lo;
            }
            if (sp < 64) stack[sp++] = Frame{lo, fl, depth};
            fl = lo;
        }
    }
    if (last - first > 16) {
        c_insertion_sort(first, first + 16, c);
        for (int *i = first + 16; i != last; ++i) c_unguarded_linear_insert(i, c);
    } else
        c_insertion_sort(first, last, c);
}

}  // namespace ilupp
