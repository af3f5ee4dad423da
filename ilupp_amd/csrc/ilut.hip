// ilupp_amd/csrc/ilut.hip -- Saad's ILUT(p, tau), heap variant, for gfx950.
//
// Replaces ILUT_heap (reference ILUT.hpp:199-278), threshold_and_drop (dropping.hpp:8-34), the working-row
// containers vector_sparse_dynamic / vector_sparse_ordered (sparse.h:169-323, sparse_implementation.h:950-1093),
// append_row_with_prefix/suffix (:3190-3230) and compress (:3696-3722).
//
// The sparsity pattern of L and U depends on the VALUES (thresholds, top-k by magnitude), so nothing can be
// scheduled ahead: this is a row-wise dataflow kernel.  One persistent launch; every wave repeatedly claims the
// next row from an atomic counter (so a row only ever waits on rows claimed earlier: forward progress without
// assuming dispatch order) and computes it exactly like the reference:
//   * working row = sparse accumulator over a dense occupancy array (one n-long int array per resident wave --
//     288 GB of HBM buys what the CPU code does with a single array) + insertion-ordered slot list + binary
//     min-heap of slots keyed by column;
//   * eliminate in ascending column order; row k of U is awaited through done[k] (write-through stored,
//     drained, flagged; consumers poll with sc1 loads);
//   * stage-1 drop `|w_k| < tau*||A[i,<i]||` before the division (strict <), stage-2 threshold_and_drop with
//     the 2-norm accumulated in INSERTION order, strict >, top-(p-1) by magnitude;
//   * the top-k cut under equal magnitudes is defined by libstdc++'s std::sort (introsort, unstable): the same
//     algorithm runs here on the same slot-ordered candidate list (bits/stl_algo.h:1855-1957), so the index
//     arrays stay bit-exact even on structured grids where ties do occur;
//   * rows go to fixed-pitch slabs (p entries per row), a compaction pass drops exact zeros like compress(0.0).
//
// The row algorithm is a chain of dependent pointer operations: the 64 lanes of the wave run it redundantly with
// wave-uniform control flow, and fetch the U rows it eliminates with cooperatively (one memory round trip per row
// instead of one per entry).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ilupp {

struct IlutWork {
    int32_t *occ;       // [workers][n]   index -> slot or -1
    double *sdata;      // [workers][cap] slot -> value
    int32_t *sptr;      // [workers][cap] slot -> index (insertion order)
    int32_t *heap;      // [workers][cap]
    int32_t *listL;     // [workers][cap]
    int32_t *listU;     // [workers][cap]
    int32_t cap;
};

// ---- libstdc++ std::sort on slot ids by DEcreasing |key| (dropping.hpp:25-26), restated ----------------
struct AbsDesc {
    const double *key;
    __device__ __forceinline__ bool operator()(int a, int b) const { return fabs(key[a]) > fabs(key[b]); }
};

__device__ void s_unguarded_linear_insert(int *last, const AbsDesc &c)
{
    const int v = *last;
    int *next = last - 1;
    while (c(v, *next)) { *last = *next; last = next; --next; }
    *last = v;
}

__device__ void s_insertion_sort(int *first, int *last, const AbsDesc &c)
{
    if (first == last) return;
    for (int *i = first + 1; i != last; ++i) {
        if (c(*i, *first)) {
            const int v = *i;
            for (int *q = i; q != first; --q) *q = *(q - 1);
            *first = v;
        } else
            s_unguarded_linear_insert(i, c);
    }
}

__device__ void s_push_heap(int *first, long hole, long top, int v, const AbsDesc &c)
{
    long parent = (hole - 1) / 2;
    while (hole > top && c(first[parent], v)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = v;
}

__device__ void s_adjust_heap(int *first, long hole, long len, int v, const AbsDesc &c)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (c(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    s_push_heap(first, hole, top, v, c);
}

__device__ void s_heapsort(int *first, int *last, const AbsDesc &c)
{
    const long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            const int v = first[parent];
            s_adjust_heap(first, parent, len, v, c);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        const int v = *last;
        *last = *first;
        s_adjust_heap(first, 0, last - first, v, c);
    }
}

__device__ void sort_slots_by_abs_desc(int *list, int len, const double *key)
{
    AbsDesc c{key};
    if (len <= 0) return;
    int *first = list, *last = list + len;
    long lg = 0, m = len;
    while (m > 1) { m >>= 1; ++lg; }
    // __introsort_loop with its recursion (on the right part) turned into an explicit stack
    struct Frame { int *first, *last; long depth; };
    Frame stack[64];
    int sp = 0;
    stack[sp++] = Frame{first, last, 2 * lg};
    while (sp > 0) {
        Frame f = stack[--sp];
        int *fl = f.last;
        long depth = f.depth;
        while (fl - f.first > 16) {
            if (depth == 0) { s_heapsort(f.first, fl, c); break; }
            --depth;
            int *mid = f.first + (fl - f.first) / 2;
            // __move_median_to_first(first, first+1, mid, last-1)
            int *a = f.first + 1, *b = mid, *cc = fl - 1, *res = f.first;
            int *pick;
            if (c(*a, *b)) { if (c(*b, *cc)) pick = b; else if (c(*a, *cc)) pick = cc; else pick = a; }
            else if (c(*a, *cc)) pick = a;
            else if (c(*b, *cc)) pick = cc;
            else pick = b;
            { const int t = *res; *res = *pick; *pick = t; }
            // __unguarded_partition(first+1, last, first)
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
            int *cut = lo;
            // the reference recurses on [cut, last) FIRST and then loops on [first, cut): the two parts are
            // disjoint, so the order of processing does not change the result; push the right part
            if (sp < 64) stack[sp++] = Frame{cut, fl, depth};
            fl = cut;
        }
    }
    if (last - first > 16) {
        s_insertion_sort(first, first + 16, c);
        for (int *i = first + 16; i != last; ++i) s_unguarded_linear_insert(i, c);
    } else
        s_insertion_sort(first, last, c);
}

// kept slots by increasing column index (unique keys: dead slots carry 0 and are never kept)
__device__ void sort_slots_by_index(int *list, int len, const int32_t *sptr)
{
    // binary-insertion-free simple shell/insertion hybrid: lists are at most p-1 long
    for (int gap = len / 2; gap > 0; gap = (gap == 2) ? 1 : (int)(gap / 2.2)) {
        for (int i = gap; i < len; ++i) {
            const int v = list[i];
            int j = i;
            while (j >= gap && sptr[list[j - gap]] > sptr[v]) { list[j] = list[j - gap]; j -= gap; }
            list[j] = v;
        }
    }
}

// threshold_and_drop, dropping.hpp:8-34
__device__ int threshold_and_drop_dev(const double *sdata, const int32_t *sptr, int wnnz, int *list, int nkeep,
                                      double tau, int from, int to)
{
    if (nkeep <= 0) return 0;
    double z = 0.0;
    for (int x = 0; x < wnnz; ++x) {          // norm2(from,to): insertion order (sparse_implementation.h:1087-1093)
        const int i = sptr[x];
        if (from <= i && i < to) { const double sq = sdata[x] * sdata[x]; z = z + sq; }
    }
    const double norm = sqrt(z);
    const double thr = norm * tau;
    int len = 0;
    for (int x = 0; x < wnnz; ++x) {
        const int i = sptr[x];
        if (from <= i && i < to && fabs(sdata[x]) > thr) list[len++] = x;
    }
    if (len > nkeep) {
        sort_slots_by_abs_desc(list, len, sdata);
        len = nkeep;
    }
    sort_slots_by_index(list, len, sptr);
    return len;
}

#ifndef ILUT_SPIN
#define ILUT_SPIN (1u << 24)
#endif
static constexpr unsigned kIlutSpinLimit = ILUT_SPIN;

// ctrl: [0] next row, [1] error/timeout, [2] smallest row with a zero pivot (init INT_MAX)
// The working row (slots, heap, candidate lists) and its occupancy map live in LDS -- a hash table instead of the
// dense array -- as long as the row holds at most kIlutLdsCap entries; a row that outgrows LDS is started over with
// the wave's global-memory working row (dense occupancy array).
static constexpr int kIlutLdsCap = 1024;
static constexpr int kIlutHash = 2048;

__global__ void __launch_bounds__(64)
k_ilut_rows(int32_t n, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval,
            int32_t p, double tau, IlutWork wk,
            int32_t *Lrow_idx, double *Lrow_val, int32_t *Llen,
            int32_t *Urow_idx, double *Urow_val, int32_t *Ulen,
            int32_t *done, int32_t *ctrl)
{
    const int lane = threadIdx.x;
    const size_t w = blockIdx.x;
    __shared__ double l_sdata[kIlutLdsCap];
    __shared__ int32_t l_sptr[kIlutLdsCap], l_heap[kIlutLdsCap], l_listL[kIlutLdsCap], l_listU[kIlutLdsCap];
    __shared__ int32_t h_key[kIlutHash], h_slot[kIlutHash];
    for (int q = lane; q < kIlutHash; q += 64) h_key[q] = -1;
    // row k of U as the wave fetched it (one round trip for the whole row instead of one per entry)
    __shared__ int32_t u_idx[64];
    __shared__ double u_val[64];

    for (;;) {
        int i = 0;
        if (lane == 0) i = atomicAdd(&ctrl[0], 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= n) break;
        // the row is a chain of dependent pointer operations: all 64 lanes execute it redundantly (identical values,
        // wave-uniform control flow -- a lane-0-only region with breaks defeats hipcc's loop structurizer), and share
        // the memory round trips where a whole U row is fetched.
        // the working row starts in LDS and moves to the wave's global-memory arrays if it outgrows them
        bool in_lds = true;
        int32_t *occ = wk.occ + w * (size_t)n;
        double *sdata = l_sdata;
        int32_t *sptr = l_sptr, *heap = l_heap, *listL = l_listL, *listU = l_listU;
        int cap = kIlutLdsCap;
        // occupancy map: column -> slot (or -1).  Open-addressing hash in LDS (erased keys keep their place with
        // slot -1: a column is never touched again after its elimination, ILUT.hpp:234-255), or the dense array
        auto occ_get = [&](int j) -> int {
            if (!in_lds) return occ[j];
            unsigned h = ((unsigned)j * 0x9E3779B1u) >> 21;
            for (;;) {
                const int kk = h_key[h];
                if (kk == j) return h_slot[h];
                if (kk == -1) return -1;
                h = (h + 1) & (kIlutHash - 1);
            }
        };
        auto occ_put = [&](int j, int sl) {
            if (!in_lds) { occ[j] = sl; return; }
            unsigned h = ((unsigned)j * 0x9E3779B1u) >> 21;
            for (;;) {
                const int kk = h_key[h];
                if (kk == j || kk == -1) { h_key[h] = j; h_slot[h] = sl; return; }
                h = (h + 1) & (kIlutHash - 1);
            }
        };
        int wnnz = 0, hlen = 0;
        bool overflow = false;
        // LDS working row full: continue in global memory (all lanes copy; the values are identical in every lane)
        auto migrate = [&]() {
            double *g_sdata = wk.sdata + w * (size_t)wk.cap;
            int32_t *g_sptr = wk.sptr + w * (size_t)wk.cap, *g_heap = wk.heap + w * (size_t)wk.cap;
            for (int q = lane; q < wnnz; q += 64) { g_sdata[q] = l_sdata[q]; g_sptr[q] = l_sptr[q]; }
            for (int q = lane; q < hlen; q += 64) g_heap[q] = l_heap[q];
            for (int q = lane; q < kIlutHash; q += 64) { const int kk = h_key[q]; if (kk != -1) { occ[kk] = h_slot[q]; h_key[q] = -1; } }
            __builtin_amdgcn_s_waitcnt(0);          // the wave reads these arrays right away
            sdata = g_sdata; sptr = g_sptr; heap = g_heap;
            listL = wk.listL + w * (size_t)wk.cap; listU = wk.listU + w * (size_t)wk.cap;
            cap = wk.cap;
            in_lds = false;
        };
        // insert-on-miss accessor (sparse.h:298-311): returns the slot of column j
        auto slot_of = [&](int j) -> int {
            int s = occ_get(j);
            if (s < 0) {
                if (wnnz >= cap) {
                    if (in_lds) migrate(); else { overflow = true; return 0; }
                }
                s = wnnz++;
                occ_put(j, s);
                sptr[s] = j;
                sdata[s] = 0.0;
                // push_heap (min-heap on column index)
                int hole = hlen++;
                while (hole > 0) {
                    const int parent = (hole - 1) / 2;
                    if (sptr[heap[parent]] > j) { heap[hole] = heap[parent]; hole = parent; } else break;
                }
                heap[hole] = s;
            }
            return s;
        };
        double thr1 = 0.0;
        {
            // (2.) scatter the row, norm of the strictly-lower part in CSR order (ILUT.hpp:222-231)
            double norm_wL = 0.0;
            for (int k = Aptr[i]; k < Aptr[i + 1]; ++k) {
                const int c = Aidx[k];
                const double v = Aval[k];
                sdata[slot_of(c)] = v;
                if (c < i) { const double sq = v * v; norm_wL = norm_wL + sq; }
            }
            norm_wL = sqrt(norm_wL);
            thr1 = tau * norm_wL;
        }
        // (3.-9.) eliminate in ascending column order (ILUT.hpp:234-255)
        bool timed_out = false;
        for (;;) {
            int act = 0;            // 0: done with the eliminations, 1: next heap entry, 2: eliminate with row k of U
            int k = 0, x = 0;
            double wkv = 0.0;
            if (hlen > 0) {
                // pop_next_index (sparse.h:313-322)
                x = heap[0];
                const int lastv = heap[--hlen];
                if (hlen > 0) {
                    int hole = 0;
                    for (;;) {
                        int child = 2 * hole + 1;
                        if (child >= hlen) break;
                        if (child + 1 < hlen && sptr[heap[child + 1]] < sptr[heap[child]]) child++;
                        if (sptr[heap[child]] < sptr[lastv]) { heap[hole] = heap[child]; hole = child; } else break;
                    }
                    heap[hole] = lastv;
                }
                k = sptr[x];
                if (k < i) {
                    wkv = sdata[x];
                    if (wkv == 0.0) {
                        act = 1;                                     // stale heap entry (ILUT.hpp:239-240)
                    } else if (fabs(wkv) < thr1) {                   // stage-1 drop, before dividing (ILUT.hpp:244-245)
                        { const int so = occ_get(k); if (so >= 0) { sdata[so] = 0.0; occ_put(k, -1); } }     // zero_set(k)
                        act = 1;
                    } else {
                        act = 2;
                    }
                }
            }
            act = __builtin_amdgcn_readfirstlane(act);       // (already uniform; keeps the branches scalar)
            if (act == 0 || overflow) break;
            if (act == 1) continue;
            k = __builtin_amdgcn_readfirstlane(k);
            // row k of U must be complete
            unsigned spins = 0;
            while (ld_agent_i32(&done[k]) == 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kIlutSpinLimit) { timed_out = true; break; }
            }
            if (timed_out) break;
            order_after_poll();
            const size_t ub = (size_t)k * p;
            const int ul = ld_agent_i32(&Ulen[k]);
            double m = 0.0;
            for (int base = 0; base < ul; base += 64) {
                const int q = base + lane;
                if (q < ul) { u_idx[lane] = ld_agent_i32(&Urow_idx[ub + q]); u_val[lane] = ld_agent_f64(&Urow_val[ub + q]); }
                {
                    const int cntq = ul - base < 64 ? ul - base : 64;
                    int j = 0;
                    if (base == 0) { m = wkv / u_val[0]; sdata[x] = m; j = 1; }      // w[k] /= U[k,k]  (diag first)
                    for (; j < cntq; ++j) {                                            // w -= w[k] * U[k, j>k]
                        const double pr = m * u_val[j];
                        const int s2 = slot_of(u_idx[j]);
                        sdata[s2] = sdata[s2] - pr;
                    }
                }
            }
        }
        {
            // (10.) dropping (ILUT.hpp:259,261)
            const int nL = threshold_and_drop_dev(sdata, sptr, wnnz, listL, p - 1, tau, 0, i);
            const int nU = threshold_and_drop_dev(sdata, sptr, wnnz, listU, p - 1, tau, i + 1, n);
            // (11.) L row = kept entries then (i, 1.0);  (12.) U row = (i, w[i]) then kept entries
            const size_t lb = (size_t)i * p;
            for (int q = 0; q < nL; ++q) { Lrow_idx[lb + q] = sptr[listL[q]]; Lrow_val[lb + q] = sdata[listL[q]]; }
            Lrow_idx[lb + nL] = i; Lrow_val[lb + nL] = 1.0;
            Llen[i] = nL + 1;
            const int sd = slot_of(i);                     // w[i] inserts a zero slot when the row has no diagonal
            const double piv = sdata[sd];
            st_agent_i32(&Urow_idx[lb], i); st_agent_f64(&Urow_val[lb], piv);
            for (int q = 0; q < nU; ++q) { st_agent_i32(&Urow_idx[lb + 1 + q], sptr[listU[q]]); st_agent_f64(&Urow_val[lb + 1 + q], sdata[listU[q]]); }
            st_agent_i32(&Ulen[i], nU + 1);
            if (lane == 0 && piv == 0.0) atomicMin(&ctrl[2], i);        // ILUT.hpp:269-270 (reported after the sweep)
            if (lane == 0 && (timed_out || overflow)) atomicExch(&ctrl[1], overflow ? 2 : 1);
            drain_stores();
            st_agent_i32(&done[i], 1);
            // (13.) zero_reset (sparse_implementation.h:1036-1040)
            if (in_lds) { for (int q = lane; q < kIlutHash; q += 64) h_key[q] = -1; }
            else { for (int q = lane; q < wnnz; q += 64) occ[sptr[q]] = -1; __builtin_amdgcn_s_waitcnt(0); }
        }
    }
}

// ---- compaction of the fixed-pitch slabs into CSR, dropping exact zeros (compress(0.0), :3696-3722) -------
__global__ void k_slab_count(int32_t n, int32_t p, const int32_t *__restrict__ rlen, const double *__restrict__ rval, int32_t *__restrict__ cnt)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int c = 0;
    const size_t b = (size_t)r * p;
    for (int q = 0; q < rlen[r]; ++q) c += (fabs(rval[b + q]) > 0.0) ? 1 : 0;
    cnt[r] = c;
}
__global__ void k_slab_fill(int32_t n, int32_t p, const int32_t *__restrict__ rlen, const int32_t *__restrict__ ridx,
                            const double *__restrict__ rval, const int32_t *__restrict__ ptr,
                            int32_t *__restrict__ idx, double *__restrict__ val)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int o = ptr[r];
    const size_t b = (size_t)r * p;
    for (int q = 0; q < rlen[r]; ++q)
        if (fabs(rval[b + q]) > 0.0) { idx[o] = ridx[b + q]; val[o] = rval[b + q]; ++o; }
}

static void compact_slab(hipStream_t st, int32_t n, int32_t p, const int32_t *rlen, const int32_t *ridx, const double *rval, DevMat *M)
{
    int32_t *cnt = nullptr;
    ILUPP_HIP(pool_malloc(&cnt, sizeof(int32_t) * (size_t)n));
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_slab_count, dim3(gb), dim3(256), 0, st, n, p, rlen, rval, cnt);
    M->n = n; M->is_csr = true; M->owns = true;
    ILUPP_HIP(pool_malloc(&M->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(hipMemsetAsync(M->ptr, 0, sizeof(int32_t), st));
    size_t tmp_bytes = 0;
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, cnt, M->ptr + 1, n, st));
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, cnt, M->ptr + 1, n, st));
    int32_t tot = 0;
    ILUPP_HIP(hipMemcpyAsync(&tot, M->ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    ILUPP_HIP(pool_free(tmp)); ILUPP_HIP(pool_free(cnt));
    M->nnz = tot;
    ILUPP_HIP(pool_malloc(&M->idx, sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1)));
    ILUPP_HIP(pool_malloc(&M->val, sizeof(double) * (size_t)(tot > 0 ? tot : 1)));
    hipLaunchKernelGGL(k_slab_fill, dim3(gb), dim3(256), 0, st, n, p, rlen, ridx, rval, M->ptr, M->idx, M->val);
}

// ILUT of the row-major view held in A; on ILUPP_ERR_ZERO_PIVOT *err_row is the first row with a zero pivot
int ilut_factor(hipStream_t st, const DevMat &A, int32_t max_fill_in, double threshold, DevMat *L, DevMat *U,
                int32_t *err_row, float *kernel_ms)
{
    const int32_t n = A.n;
    int32_t p = max_fill_in;
    if (p < 1) p = 1;
    if (p > n) p = n;                                           // ILUT.hpp:211-212
    int32_t *Lri, *Uri, *Llen, *Ulen, *done, *ctrl;
    double *Lrv, *Urv;
    const size_t slab = (size_t)n * p;
    ILUPP_HIP(pool_malloc(&Lri, sizeof(int32_t) * slab));
    ILUPP_HIP(pool_malloc(&Uri, sizeof(int32_t) * slab));
    ILUPP_HIP(pool_malloc(&Lrv, sizeof(double) * slab));
    ILUPP_HIP(pool_malloc(&Urv, sizeof(double) * slab));
    ILUPP_HIP(pool_malloc(&Llen, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&Ulen, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&done, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&ctrl, 256));
    // the wave-parallel kernel (ilut_wp.hip) first; k_ilut_rows below is the any-capacity fallback and the A/B check
    // (ILUPP_ILUT_SEQUENTIAL=1)
    {
        const char *force_seq = getenv("ILUPP_ILUT_SEQUENTIAL");
        if (!(force_seq && force_seq[0] == '1')) {
            const int rw = ilut_rows_wp(st, A, p, threshold, Lri, Lrv, Llen, Uri, Urv, Ulen, ctrl, kernel_ms);
            if (rw != 1) {
                int32_t hw[4] = {0, 0, 0, 0};
                ILUPP_HIP(hipMemcpyAsync(hw, ctrl, 16, hipMemcpyDeviceToHost, st));
                ILUPP_HIP(hipStreamSynchronize(st));
                int rc = rw;
                if (rc == ILUPP_OK && hw[2] != 0x7fffffff) { rc = ILUPP_ERR_ZERO_PIVOT; if (err_row) *err_row = hw[2]; }
                if (rc == ILUPP_OK) {
                    compact_slab(st, n, p, Llen, Lri, Lrv, L);
                    compact_slab(st, n, p, Ulen, Uri, Urv, U);
                    ILUPP_HIP(hipStreamSynchronize(st));
                }
                for (void *q : {(void *)Lri, (void *)Uri, (void *)Lrv, (void *)Urv, (void *)Llen, (void *)Ulen, (void *)done, (void *)ctrl})
                    ILUPP_HIP(pool_free(q));
                return rc;
            }
        }
    }
    IlutWork wk = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    int32_t h[4] = {0, 0, 0, 0};
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    // resident waves: 4 per CU (LDS working rows), bounded by 48 GB of global working-row storage for the rows that
    // outgrow LDS (one dense occupancy array per wave -- 288 GB of HBM buys what the CPU code does with a single array)
    int workers = device_cu_count() * 4;
    const size_t per_worker = (size_t)n * 4 + (size_t)(n + 16) * (8 + 4 * 4);
    while (workers > 1 && (size_t)workers * per_worker > (48ull << 30)) workers >>= 1;
    if (workers > n) workers = n;
    wk.cap = n + 16;
    ILUPP_HIP(pool_malloc(&wk.occ, sizeof(int32_t) * (size_t)workers * n));
    ILUPP_HIP(pool_malloc(&wk.sdata, sizeof(double) * (size_t)workers * wk.cap));
    ILUPP_HIP(pool_malloc(&wk.sptr, sizeof(int32_t) * (size_t)workers * wk.cap));
    ILUPP_HIP(pool_malloc(&wk.heap, sizeof(int32_t) * (size_t)workers * wk.cap));
    ILUPP_HIP(pool_malloc(&wk.listL, sizeof(int32_t) * (size_t)workers * wk.cap));
    ILUPP_HIP(pool_malloc(&wk.listU, sizeof(int32_t) * (size_t)workers * wk.cap));
    ILUPP_HIP(hipMemsetAsync(wk.occ, 0xff, sizeof(int32_t) * (size_t)workers * n, st));
    ILUPP_HIP(hipMemsetAsync(done, 0, sizeof(int32_t) * (size_t)n, st));
    const int32_t init[4] = {0, 0, 0x7fffffff, 0};
    ILUPP_HIP(hipMemcpyAsync(ctrl, init, 16, hipMemcpyHostToDevice, st));
    ILUPP_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_ilut_rows, dim3((unsigned)workers), dim3(64), 0, st, n, A.ptr, A.idx, A.val, p, threshold, wk,
                       Lri, Lrv, Llen, Uri, Urv, Ulen, done, ctrl);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    ILUPP_HIP(hipMemcpyAsync(h, ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    ILUPP_HIP(hipEventDestroy(e0));
    ILUPP_HIP(hipEventDestroy(e1));
    int rc = ILUPP_OK;
    if (h[1] == 1) rc = ILUPP_ERR_TIMEOUT;
    else if (h[1] == 2) rc = ILUPP_ERR_MEMORY;
    else if (h[2] != 0x7fffffff) { rc = ILUPP_ERR_ZERO_PIVOT; if (err_row) *err_row = h[2]; }
    if (rc == ILUPP_OK) {
        compact_slab(st, n, p, Llen, Lri, Lrv, L);
        compact_slab(st, n, p, Ulen, Uri, Urv, U);
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    for (void *q : {(void *)wk.occ, (void *)wk.sdata, (void *)wk.sptr, (void *)wk.heap, (void *)wk.listL, (void *)wk.listU,
                    (void *)Lri, (void *)Uri, (void *)Lrv, (void *)Urv, (void *)Llen, (void *)Ulen, (void *)done, (void *)ctrl})
        ILUPP_HIP(pool_free(q));
    return rc;
}

}  // namespace ilupp
