// ilupp_amd/csrc/ilucp.hip -- ILUCP: the Crout ILU with column pivoting (gfx950).  SURVEY section 8 (f4).
//
// Reference: ILUCP4, ILUC.hpp:212-370 (ILUCPPreconditioner, preconditioner_implementation.h:1117-1147; binding.cpp:343-356).  Step k computes
// row k of U from row k of the matrix and the rows of U it has multipliers for, takes as pivot the largest entry of that row if it beats
// piv_tol times the entry in column perm[k] (take_largest_elements_by_abs_value_with_threshold_pivot_last, sparse_implementation.h:1151-1216),
// swaps that column to position k, and computes column k of L from the pivot's column.  The pivot of a step decides which entries of all
// later rows are still alive: the steps form a chain, like those of the multilevel factorisation with pivoting (pilucdp.hip), and the
// kernel has the same shape -- ONE WAVE walks the chain, its lanes work inside the step (subtracting stored rows entry per lane with
// new indices appended in entry order, norms in insertion order, candidates collected in insertion order, the reference's own partial
// sort where a bounded fill cuts into them), and one lane keeps the reference's linked lists (which column is next in a row of the
// matrix, which column of L has its next entry in which row: ILUC.hpp:37-101) exactly as the reference threads them, because the ORDER
// in which a row is assembled is the order of its norm's sum, of its candidates and of the stored row.
// The stores are as large as the reference reserves (min(max_fill_in * n, mem_factor * nnz), :229) and running out of them is the
// reference's error ("ILUCP4: Insufficient memory reserved").
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "piluc_dev.h"
#include "dp_dev.h"

namespace ilupp {

struct CpArgs {
    int32_t n;
    const int32_t *Cp, *Ci; const double *Cv;            // the matrix by its major slices ("columns" of the reference's Acol)
    double threshold, piv_tol;
    int32_t rp, max_fill, reserved;
    int32_t *listA, *headA, *firstA, *listL, *firstL, *startU, *linkU, *rowU, *perm, *iperm, *nonpiv;
    int32_t *Uptr, *Uidx; double *Uval;
    int32_t *Lptr, *Lidx; double *Lval;
    DpRec *zrec, *wrec; int32_t *zlist, *wlist;
    double *key; int32_t *cand; unsigned long long *sortk;
    int32_t *ctrl;                                        // [0] status (0 done, 3: memory reserved was insufficient), [1] zero pivots, [5] the step
};

// sum of x * x over the slots whose index is at least lo, in slot order
__device__ double cp_norm2_from(const SpVec &v, int nnz, int lo, int lane)
{
    double acc = 0.0;
    for (int base = 0; base < nnz; base += 64) {
        const int s = base + lane;
        const int c = s < nnz ? v.list[s] : -1;
        const double x = (s < nnz && c >= lo) ? v.rec[c].val : 0.0;
        const bool in = s < nnz && c >= lo;
        const double t = x * x;
        const int cnt = nnz - base < 64 ? nnz - base : 64;
        for (int i = 0; i < cnt; ++i) { const double ti = wv_f64(t, i); const int on = wv_i32((int)in, i); if (on) acc = acc + ti; }
    }
    return acc;
}

// candidates |x| > thr (index >= lo, index != skip) in slot order -> cand / key; returns their number
__device__ int cp_candidates(const CpArgs &A, const SpVec &v, int nnz, double thr, int lo, int skip, int lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    int cnt = 0;
    for (int base = 0; base < nnz; base += 64) {
        const int s = base + lane;
        const bool act = s < nnz;
        const int c = act ? v.list[s] : 0;
        const double a = act ? fabs(v.rec[c].val) : 0.0;
        const bool ok = act && c >= lo && c != skip && a > thr;
        const unsigned long long mask = __ballot(ok);
        if (ok) { const int p = cnt + __popcll(mask & lt); A.cand[p] = c; A.key[p] = a; }
        cnt += __popcll(mask);
    }
    DP_SYNC();
    return cnt;
}

__global__ void __launch_bounds__(64) k_ilucp(CpArgs A)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n = A.n;
    const SpVec z{A.zrec, A.zlist}, w{A.wrec, A.wlist};
    int znnz = 0, wnnz = 0, zero_piv = 0, prev_pivot = -1, pU = 0, pL = 0;
    double piv_tol = A.piv_tol;
#define CP_FAIL(code) do { if (lane == 0) { A.ctrl[0] = (code); A.ctrl[1] = zero_piv; A.ctrl[5] = k; } return; } while (0)

    for (int k = 0; k < n; ++k) {
        if (k == A.rp) piv_tol = 1.0;                                               // :247-248
        for (int s = lane; s < znnz; s += 64) { const int c = z.list[s]; if (c != prev_pivot) z.rec[c].slot = -1; }
        for (int s = lane; s < wnnz; s += 64) w.rec[w.list[s]].slot = -1;
        znnz = wnnz = 0;
        // ---- row k of the matrix: the columns whose next entry lies in row k, in the order of their list (:252-256) ----
        int nrow = 0;
        if (lane == 0) for (int h = A.headA[k]; h != -1; h = A.listA[h]) A.cand[nrow++] = h;
        nrow = __shfl(nrow, 0);
        DP_SYNC();
        for (int base = 0; base < nrow; base += 64) {
            const int i = base + lane;
            const bool act = i < nrow;
            const int h = act ? A.cand[i] : 0;
            const bool ok = act && A.nonpiv[h] != 0;
            const unsigned long long mask = __ballot(ok);
            if (ok) { const int s = znnz + __popcll(mask & lt); z.list[s] = h; z.rec[h] = DpRec{A.Cv[A.firstA[h]], s, 0}; }
            znnz += __popcll(mask);
        }
        DP_SYNC();
        // ---- minus the rows of U this row has multipliers for, in the order of the list of row k of L (:258-267) ----
        for (int h = A.listL[k]; h != -1; ) {
            const double f = A.Lval[A.firstL[h]];
            const int e0 = A.Uptr[h], e1 = A.Uptr[h + 1];
            const int hn = A.listL[h];
            const bool mine = e0 + lane < e1;
            dp_subtract(z, znnz, f, A.Uidx, A.Uval, e0, e1, mine ? A.Uidx[e0 + lane] : 0, mine ? A.Uval[e0 + lane] : 0.0, lane);
            h = hn;
        }
        // ---- the row of U with its pivot last in the list (sparse_implementation.h:1151-1216), :269-286 ----
        const int pk = A.perm[k];
        int nU = 0, off = 0;
        for (int pass = 0; pass < 2 && nU == 0; ++pass) {
            const double tau = pass == 0 ? A.threshold : 0.0;
            if (pass == 1 && !(A.threshold > 0.0)) break;
            double best = 0.0;
            for (int s = lane; s < znnz; s += 64) { const double a = fabs(z.rec[z.list[s]].val); if (a > best) best = a; }
            best = wv_max_f64(best);
            const double norm = sqrt(dp_seq_sum(z, znnz, 1, lane));
            const DpRec rp = z.rec[pk];
            const double at_pivot = rp.slot >= 0 ? rp.val : 0.0;
            const int lim = A.max_fill;
            if (best * piv_tol > fabs(at_pivot)) {                                  // pivoting
                const int cnt = cp_candidates(A, z, znnz, norm * tau, 0, -1, lane);
                if (cnt > lim) {
                    if (lane == 0) select_largest(A.key, A.cand, 0, cnt - 1, lim);
                    DP_SYNC();
                    off = cnt - lim; nU = lim;
                } else {
                    // the first largest candidate goes to the end (strictly greater: the first of equals)
                    double mx = 0.0;
                    int pos = 0x7fffffff;
                    for (int i = lane; i < cnt; i += 64) { const double a = A.key[i]; if (a > mx) { mx = a; pos = i; } }
                    pos = wv_argmax_first(mx, pos);
                    if (pos == 0x7fffffff) pos = 0;
                    if (cnt > 0 && lane == 0) { const int t = A.cand[pos]; A.cand[pos] = A.cand[cnt - 1]; A.cand[cnt - 1] = t; }
                    DP_SYNC();
                    off = 0; nU = cnt;
                }
            } else if (at_pivot != 0) {
                const int cnt = cp_candidates(A, z, znnz, norm * tau, 0, pk, lane);
                if (cnt > lim - 1) {
                    if (lane == 0) { select_largest(A.key, A.cand, 0, cnt - 1, lim); }
                    DP_SYNC();
                    off = cnt - lim + 1;                                            // (n - 1 of them from there, then the pivot: as the reference has it)
                    if (lane == 0) A.cand[off + lim - 1] = pk;
                    DP_SYNC();
                    nU = lim;
                } else {
                    if (lane == 0) A.cand[cnt] = pk;
                    DP_SYNC();
                    off = 0; nU = cnt + 1;
                }
            }
        }
        if (nU == 0) {                                                              // :281-286
            ++zero_piv;
            dp_touch(z, znnz, pk, lane);
            if (lane == 0) { z.rec[pk].val = 1.0; A.cand[0] = pk; }
            DP_SYNC();
            off = 0; nU = 1;
        }
        if ((long)pU + nU > (long)A.reserved) CP_FAIL(3);
        const int c0 = A.cand[off + nU - 1];                                        // the pivot's column
        for (int j = lane; j < nU; j += 64) {
            const int pos = pU + j, c = A.cand[off + nU - 1 - j];
            A.Uval[pos] = z.rec[c].val; A.Uidx[pos] = c;
            if (j > 0) { A.linkU[pos] = A.startU[c]; A.startU[c] = pos; A.rowU[pos] = k; }
        }
        DP_SYNC();
        const double Ukk = A.Uval[pU];
        if (lane == 0) {
            A.Uptr[k + 1] = pU + nU;
            const int p = A.iperm[c0];
            const int t = A.iperm[pk]; A.iperm[pk] = A.iperm[c0]; A.iperm[c0] = t;
            const int u = A.perm[k]; A.perm[k] = A.perm[p]; A.perm[p] = u;
            A.nonpiv[c0] = 0;
            z.rec[c0].slot = -2;                                                    // dead as a column from here on
        }
        prev_pivot = c0;
        pU += nU;
        DP_SYNC();
        // ---- column k of L: the pivot's column below row k, minus the columns of L the pivot's column of U names (:309-327) ----
        {
            const int a0 = A.Cp[c0], a1 = A.Cp[c0 + 1];
            for (int base = a0; base < a1; base += 64) {
                const int e = base + lane;
                const bool act = e < a1;
                const int r = act ? A.Ci[e] : -1;
                const int pr = (act && e > a0) ? A.Ci[e - 1] : -1;
                const bool ok = act && r > k;
                const bool first = ok && r != pr;
                const unsigned long long mask = __ballot(first);
                if (first) { const int s = wnnz + __popcll(mask & lt); w.list[s] = r; w.rec[r] = DpRec{A.Cv[e], s, 0}; }
                wnnz += __popcll(mask);
                unsigned long long dup = __ballot(ok && !first);
                if (dup) {
                    DP_SYNC();
                    if (lane == 0)
                        while (dup) { const int b = __ffsll((long long)dup) - 1; dup &= dup - 1; w.rec[A.Ci[base + b]].val = A.Cv[base + b]; }
                }
            }
            DP_SYNC();
            for (int h = A.startU[c0]; h != -1; ) {
                const int r = A.rowU[h];
                const double uv = A.Uval[h];
                const int hn = A.linkU[h];
                const int e0 = A.Lptr[r], e1 = A.Lptr[r + 1];
                const bool mine = e0 + lane < e1;
                dp_subtract(w, wnnz, uv, A.Lidx, A.Lval, e0, e1, mine ? A.Lidx[e0 + lane] : 0, mine ? A.Lval[e0 + lane] : 0.0, lane);
                h = hn;
            }
        }
        // ---- take_largest_elements_by_abs_value_with_threshold(list_L, max_fill_in - 1, threshold, k + 1, n), :1322-1357; by index ----
        int nL;
        {
            const double norm = sqrt(cp_norm2_from(w, wnnz, k + 1, lane));
            const int cnt = cp_candidates(A, w, wnnz, norm * A.threshold, k + 1, -1, lane);
            const int lim = A.max_fill - 1;
            int o2 = 0;
            if (cnt > lim) {
                if (lane == 0 && lim > 0) select_largest(A.key, A.cand, 0, cnt - 1, lim);
                DP_SYNC();
                o2 = cnt - lim;
            }
            nL = cnt - o2;
            if (nL <= 64) {
                const unsigned long long sorted = dp_sort_n(lane < nL ? (unsigned long long)(unsigned)A.cand[o2 + lane] : ~0ull, nL, lane);
                if (lane < nL) A.cand[lane] = (int)(unsigned)sorted;
                DP_SYNC();
            } else {
                int N = 64;
                while (N < nL) N *= 2;
                for (int i = lane; i < N; i += 64) A.sortk[i] = i < nL ? (unsigned long long)(unsigned)A.cand[o2 + i] : ~0ull;
                DP_SYNC();
                wave_sort_u64<true>(A.sortk, N, lane);
                for (int i = lane; i < nL; i += 64) A.cand[i] = (int)(unsigned)A.sortk[i];
                DP_SYNC();
            }
        }
        if ((long)pL + nL + 1 > (long)A.reserved) CP_FAIL(3);
        for (int j = lane; j < nL; j += 64) { const int r = A.cand[j]; A.Lval[pL + 1 + j] = w.rec[r].val / Ukk; A.Lidx[pL + 1 + j] = r; }
        if (lane == 0) { A.Lval[pL] = 1.0; A.Lidx[pL] = k; A.Lptr[k + 1] = pL + nL + 1; }
        DP_SYNC();
        // ---- the lists move on (ILUC.hpp:86-101 for the matrix, :37-63 for L), as the reference threads them ----
        if (lane == 0) {
            for (int h = A.headA[k]; h != -1; h = A.listA[h]) A.firstA[h] += 1;
            for (int h = A.headA[k]; h != -1; ) {
                const int i = h;
                h = A.listA[i];
                const int f = A.firstA[i];
                if (f < A.Cp[i + 1]) { const int r = A.Ci[f]; A.listA[i] = A.headA[r]; A.headA[r] = i; }
            }
            for (int h = A.listL[k]; h != -1; h = A.listL[h]) A.firstL[h] += 1;
            A.firstL[k] = pL + 1;
            int h = A.listL[k];
            if (nL > 0) { const int j = A.Lidx[pL + 1]; A.listL[k] = A.listL[j]; A.listL[j] = k; }
            while (h != -1) {
                const int i = h;
                h = A.listL[i];
                const int f = A.firstL[i];
                if (f < A.Lptr[i + 1]) { const int j = A.Lidx[f]; A.listL[i] = A.listL[j]; A.listL[j] = i; }
            }
        }
        pL += nL + 1;
        DP_SYNC();
    }
    if (lane == 0) { A.ctrl[0] = 0; A.ctrl[1] = zero_piv; A.ctrl[5] = n; }
#undef CP_FAIL
}

// ---------------------------------------------- set-up and the stores -> matrices ----------------------------------------------
__global__ void k_cp_init(int32_t n, const int32_t *__restrict__ Cp, int32_t *perm, int32_t *iperm, int32_t *nonpiv, int32_t *listA, int32_t *headA,
                          int32_t *firstA, int32_t *listL, int32_t *startU, DpRec *zrec, DpRec *wrec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    perm[i] = iperm[i] = i; nonpiv[i] = 1; listA[i] = -1; headA[i] = -1; firstA[i] = Cp[i]; listL[i] = -1; startU[i] = -1;
    zrec[i] = DpRec{0.0, -1, 0}; wrec[i] = DpRec{0.0, -1, 0};
}
// initialize_sparse_matrix_fields (ILUC.hpp:74-84): sequentially, slice k is pushed in front of the chain of the row of its first entry -- so a
// chain holds its slices by DECREASING k.  From the slices sorted by (first row, k): every slice points at the one before it in its group,
// the head of a row is the last of its group.
__global__ void k_cp_chain_keys(int32_t n, const int32_t *__restrict__ Cp, const int32_t *__restrict__ Ci, unsigned long long *__restrict__ keys)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    keys[k] = Cp[k] < Cp[k + 1] ? (((unsigned long long)(unsigned)Ci[Cp[k]] << 32) | (unsigned)k) : ~0ull;
}
__global__ void k_cp_chains(int32_t n, const unsigned long long *__restrict__ keys, int32_t *__restrict__ listA, int32_t *__restrict__ headA)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long me = keys[i];
    if (me == ~0ull) return;
    const int row = (int)(me >> 32), k = (int)(unsigned)me;
    const unsigned long long before = i > 0 ? keys[i - 1] : ~0ull, after = i + 1 < n ? keys[i + 1] : ~0ull;
    listA[k] = (before != ~0ull && (int)(before >> 32) == row) ? (int)(unsigned)before : -1;
    if (after == ~0ull || (int)(after >> 32) != row) headA[row] = k;
}
__global__ void k_cp_count(int32_t n, const int32_t *__restrict__ ptr, const double *__restrict__ val, int32_t *__restrict__ len)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    int c = 0;
    if (r < n) for (int j = ptr[r]; j < ptr[r + 1]; ++j) c += fabs(val[j]) > 0.0 ? 1 : 0;
    len[r] = c;
}
__global__ void k_cp_compress(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
                              const int32_t *__restrict__ nptr, int32_t *__restrict__ oidx, double *__restrict__ oval)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int q = nptr[r];
    for (int j = ptr[r]; j < ptr[r + 1]; ++j)
        if (fabs(val[j]) > 0.0) { oidx[q] = idx[j]; oval[q] = val[j]; ++q; }
}

// compress() (:365-366): the entries with |x| > 0, in their order
static int cp_compress(hipStream_t st, int32_t n, const int32_t *ptr, const int32_t *idx, const double *val, bool is_csr, DevMat *M)
{
    PoolBlock b_len, b_tmp;
    ILUPP_HIP(b_len.alloc(sizeof(int32_t) * (size_t)(n + 1)));
    M->release();
    M->n = n; M->is_csr = is_csr; M->owns = true;
    ILUPP_HIP(pool_malloc(&M->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    hipLaunchKernelGGL(k_cp_count, dim3((n + 256) / 256), dim3(256), 0, st, n, ptr, val, b_len.as<int32_t>());
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, b_len.as<int32_t>(), M->ptr, n + 1, st));
    ILUPP_HIP(b_tmp.alloc(tb > 0 ? tb : 1));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tb, b_len.as<int32_t>(), M->ptr, n + 1, st));
    int32_t nnz = 0;
    ILUPP_HIP(hipMemcpyAsync(&nnz, M->ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    M->nnz = nnz;
    ILUPP_HIP(pool_malloc(&M->idx, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
    ILUPP_HIP(pool_malloc(&M->val, sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
    hipLaunchKernelGGL(k_cp_compress, dim3((n + 255) / 256), dim3(256), 0, st, n, ptr, idx, val, M->ptr, M->idx, M->val);
    ILUPP_HIP(hipStreamSynchronize(st));
    return ILUPP_OK;
}

// C: the matrix by its major slices; L by columns (the 1 first), U by rows (the pivot first, original column indices), perm (device, n)
int ilucp_factor(hipStream_t st, const DevMat &C, int32_t max_fill_in, double threshold, double piv_tol, int32_t rp, double mem_factor,
                 DevMat *L, DevMat *U, int32_t *perm_out, int32_t *zero_pivots, float *kernel_ms)
{
    const int32_t n = C.n;
    const int64_t nnz = C.nnz;
    if (max_fill_in < 1) max_fill_in = 1;                                           // :226-227
    if (max_fill_in > n) max_fill_in = n;
    int64_t reserved;
    {
        const int64_t a = (int64_t)max_fill_in * (int64_t)n, b = (int64_t)((int32_t)mem_factor) * nnz;      // (Integer) mem_factor * Acol.non_zeroes(), ILUC.hpp:229: the factor is truncated FIRST
        reserved = a < b ? a : b;
        if (reserved < 0) reserved = 0;
        if (reserved > 0x7ffffff0ll) { set_error("ILUCP: the memory to reserve exceeds 2^31 entries"); return ILUPP_ERR_UNSUPPORTED; }
    }
    int sortN = 64;
    while (sortN < n) sortN *= 2;
    const size_t slot = ((size_t)n + 64) & ~(size_t)15;
    PoolBlock b_i, b_d, b_sort, b_ctrl, b_ui, b_ul, b_ur, b_uv, b_li, b_lv, b_k0, b_k1, b_tmp;
    ILUPP_HIP(b_i.alloc(sizeof(int32_t) * slot * 16));
    ILUPP_HIP(b_d.alloc(sizeof(double) * slot * 5));
    ILUPP_HIP(b_sort.alloc(sizeof(unsigned long long) * (size_t)sortN));
    ILUPP_HIP(b_ctrl.alloc(64));
    const size_t cap = (size_t)reserved + 1;
    ILUPP_HIP(b_ui.alloc(sizeof(int32_t) * cap)); ILUPP_HIP(b_ul.alloc(sizeof(int32_t) * cap)); ILUPP_HIP(b_ur.alloc(sizeof(int32_t) * cap));
    ILUPP_HIP(b_uv.alloc(sizeof(double) * cap)); ILUPP_HIP(b_li.alloc(sizeof(int32_t) * cap)); ILUPP_HIP(b_lv.alloc(sizeof(double) * cap));
    int32_t *I = b_i.as<int32_t>();
    auto iarr = [&](int q) { return I + slot * (size_t)q; };
    CpArgs a;
    a.n = n; a.Cp = C.ptr; a.Ci = C.idx; a.Cv = C.val;
    a.threshold = threshold; a.piv_tol = piv_tol; a.rp = rp; a.max_fill = max_fill_in; a.reserved = (int32_t)reserved;
    a.listA = iarr(0); a.headA = iarr(1); a.firstA = iarr(2); a.listL = iarr(3); a.firstL = iarr(4); a.startU = iarr(5);
    a.perm = iarr(6); a.iperm = iarr(7); a.nonpiv = iarr(8); a.zlist = iarr(9); a.wlist = iarr(10); a.cand = iarr(11);
    a.Uptr = iarr(12); a.Lptr = iarr(13);
    a.linkU = b_ul.as<int32_t>(); a.rowU = b_ur.as<int32_t>(); a.Uidx = b_ui.as<int32_t>(); a.Uval = b_uv.as<double>();
    a.Lidx = b_li.as<int32_t>(); a.Lval = b_lv.as<double>();
    a.key = b_d.as<double>();
    a.zrec = reinterpret_cast<DpRec *>(a.key + slot); a.wrec = a.zrec + slot;
    a.sortk = b_sort.as<unsigned long long>();
    a.ctrl = b_ctrl.as<int32_t>();
    ILUPP_HIP(hipMemsetAsync(a.ctrl, 0, 64, st));
    ILUPP_HIP(hipMemsetAsync(a.Uptr, 0, sizeof(int32_t), st));
    ILUPP_HIP(hipMemsetAsync(a.Lptr, 0, sizeof(int32_t), st));
    ILUPP_HIP(hipMemsetAsync(a.firstL, 0, sizeof(int32_t) * (size_t)n, st));
    hipLaunchKernelGGL(k_cp_init, dim3((n + 255) / 256), dim3(256), 0, st, n, C.ptr, a.perm, a.iperm, a.nonpiv, a.listA, a.headA, a.firstA, a.listL, a.startU,
                       a.zrec, a.wrec);
    {
        ILUPP_HIP(b_k0.alloc(sizeof(unsigned long long) * (size_t)n));
        ILUPP_HIP(b_k1.alloc(sizeof(unsigned long long) * (size_t)n));
        hipLaunchKernelGGL(k_cp_chain_keys, dim3((n + 255) / 256), dim3(256), 0, st, n, C.ptr, C.idx, b_k0.as<unsigned long long>());
        size_t tb = 0;
        ILUPP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, b_k0.as<unsigned long long>(), b_k1.as<unsigned long long>(), n, 0, 64, st));
        ILUPP_HIP(b_tmp.alloc(tb > 0 ? tb : 1));
        ILUPP_HIP(hipcub::DeviceRadixSort::SortKeys(b_tmp.p, tb, b_k0.as<unsigned long long>(), b_k1.as<unsigned long long>(), n, 0, 64, st));
        hipLaunchKernelGGL(k_cp_chains, dim3((n + 255) / 256), dim3(256), 0, st, n, b_k1.as<unsigned long long>(), a.listA, a.headA);
    }
    EventPair ev;
    ILUPP_HIP(ev.create());
    ILUPP_HIP(hipEventRecord(ev.a, st));
    hipLaunchKernelGGL(k_ilucp, dim3(1), dim3(64), 0, st, a);
    ILUPP_HIP(hipEventRecord(ev.b, st));
    int32_t ctrl[8] = {0};
    ILUPP_HIP(hipMemcpyAsync(ctrl, a.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    ILUPP_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (kernel_ms) *kernel_ms = ms;
    if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] ilucp: n %d, stores of %lld: status %d at step %d, %.2f ms\n", n, (long long)reserved, ctrl[0], ctrl[5], ms);
    if (ctrl[0] != 0) { set_error("ILUCP4: Insufficient memory reserved. Increase mem_factor"); return ILUPP_ERR_MEMORY; }
    if (zero_pivots) *zero_pivots = ctrl[1];
    { const int rc = cp_compress(st, n, a.Lptr, a.Lidx, a.Lval, false, L); if (rc) return rc; }
    { const int rc = cp_compress(st, n, a.Uptr, a.Uidx, a.Uval, true, U); if (rc) return rc; }
    ILUPP_HIP(hipMemcpyAsync(perm_out, a.perm, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    return ILUPP_OK;
}

}  // namespace ilupp
