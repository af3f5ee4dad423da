// ilupp_amd/csrc/pilucdp.hip -- one level of the multilevel ILU++ preconditioner WITH pivoting (gfx950).
//
// Reference: matrix_sparse::partialILUCDP, ILUCDP.hpp:268-1404 (called from make_preprocessed_multilevelILUCDP,
// preconditioner_implementation.h:1483-1494 / :1614-1625, whenever the parameters ask for row reordering, total pivoting or a
// pivot tolerance -- the reference's default-constructed parameters do).  Crout's form of LDU: step k computes row k of U and
// column k of L; the COLUMN eliminated in step k is the largest entry of the working row (if it beats the diagonal by piv_tol),
// the ROW of step k + 1 is the one with the fewest entries in L so far (FINAL_ROW_CRIT -1..9), and the level ends when a row of
// L has grown past MOVE_LEVEL_FACTOR times the mean row length of A (or at a small pivot).  Every one of these choices depends
// on the values computed in the step before it: the algorithm is a chain of n steps, and there is nothing to run beside it --
// no ready queue, no level schedule (piluc_df.hip has those, for the parameter family that fixes rows and columns beforehand).
//
// What the GPU can do is the work INSIDE a step, and that is how this kernel is laid out: ONE WAVE walks the chain; its 64 lanes
//   * subtract a row of U from the working row z (a column of L from w): one entry per lane, new indices appended in entry order
//     by ballot / prefix count (= the reference's insertion order, which its norms, its pivot search and its selection depend on),
//   * search the pivot (first largest magnitude in insertion order: per-lane first maximum, then a (magnitude, slot) reduction),
//   * scale, collect the candidates of the dropping rule in insertion order, sort the kept ones by index (bitonic, 64-bit keys),
//   * write the row / column and thread it into the column / row lists,
// and only what is order-dependent arithmetic stays sequential: the 1-norm / 2-norm sums (in insertion order, as the reference
// adds them), the selection of the largest entries under a bounded fill (the reference's own partial sort, select_largest), and
// the bucket moves that keep the rows ordered by their number of entries in L.
// State: everything lives in HBM / L2 (dense value arrays by index + slot lists; the two factor stores with their link arrays);
// LDS holds nothing, the kernel is bound by the latency of its dependent loads (a step is a few dozen round trips).
// Afterwards, grid-wide kernels drop the explicit zeros (compress(), :1131-1132), renumber rows and columns by the inverse
// permutations and sort every row / column (permute(), :1150-1151; Anew: :1136-1145) with one radix sort per matrix.
#include <stdlib.h>

#include <chrono>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "piluc_dev.h"
#include "dp_dev.h"

namespace ilupp {

struct DpArgs {
    int32_t n;
    const int32_t *Ap, *Ai; const double *Av;          // the level's matrix by rows
    const int32_t *Cp, *Ci; const double *Cv;          // ... and by columns
    double threshold, shift_schur, min_pivot, min_elim_factor, piv_tol, move_level_factor, row_u_max;
    int32_t small_pivot_terminates, force_finish, begin_total_piv, final_row_crit, bp, bpr, epr, max_fill;
    int32_t rules, combine, scale_invdiag;
    double wgt[5], neutral, min_weight;
    int32_t *perm, *iperm, *prow, *iprow, *numb, *pnum;
    int32_t *nonpiv, *unused;
    double *Dinv;
    int32_t *Uptr, *Uidx, *linkU, *rowU, *startU; double *Uval; int32_t capU;
    int32_t *Lptr, *Lidx, *linkL, *colL, *startL; double *Lval; int32_t capL;
    int32_t *Sptr, *Sidx; double *Sval; int32_t capS;   // the Schur complement's rows as they come (column indices of this level)
    struct DpRec *zrec, *wrec; int32_t *zlist, *wlist;     // the two working vectors (below)
    double *key; int32_t *cand; unsigned long long *sortk;
    int32_t *ctrl;     // [0] status (0 done; 1 / 2 / 3: the store of U / L / the Schur complement has no room for another row: enlarge it and
                       // launch again), [1] last_row_to_eliminate, [2] n_Anew, [3] zero pivots, [4] eliminating (still / to the end),
                       // [5] the step to go on with, [6] / [7] entries of z / w to clear, [8] / [9] entries of U / L so far, [10] of the Schur complement, [11] the pivot column of the step before
    double *dctrl;     // [0] the threshold, [1] the pivot tolerance of the moment
};

// combine() and the weight of a row of U / a column of L, ILUCDP.hpp:717-726 / :905-914 (parameters_implementation.h:526-534)
__device__ double dp_weight(const DpArgs &A, double n2own, double n1other, double dinv)
{
    double w = A.neutral;
    auto comb = [&](double x, double y) {
        switch (A.combine) {
        case 1: return x + y;
        case 2: return x * y;
        case 3: { const double m = x < y ? y : x; return A.min_weight < m ? m : A.min_weight; }
        default: return x < y ? y : x;
        }
    };
    if (A.rules & PILUC_DROP_STANDARD) { const double norm = n2own == 0.0 ? 1e-16 : n2own; w = comb(w, A.wgt[0] / norm); }
    if (A.rules & PILUC_DROP_STANDARD2) w = comb(w, A.wgt[1]);
    if (A.rules & PILUC_DROP_ERR_PROP) w = comb(w, A.wgt[2] * n1other);
    if (A.rules & PILUC_DROP_ERR_PROP2) w = comb(w, A.wgt[3] * n1other / fabs(dinv));
    if (A.rules & PILUC_DROP_PIVOT) w = comb(w, A.wgt[4] * fabs(dinv));
    if (A.scale_invdiag) w = w * fabs(dinv);
    return w;
}

// the entries that pass the dropping rule, in insertion order (take_single_weight_largest_elements_by_abs_value_with_threshold,
// sparse_implementation.h:1360-1415: weight * |x| >= tau; take_largest_elements_by_abs_value_with_threshold, :1322-1357: |x| > norm * tau),
// at most `limit` of them (the largest keys, by the reference's selection), ascending by index in cand[0 .. return)
__device__ int dp_take(const DpArgs &A, const SpVec &v, int nnz, bool single, double weight, double thr, int limit, int lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    int cnt = 0;
    for (int base = 0; base < nnz; base += 64) {
        const int s = base + lane;
        const bool act = s < nnz;
        const int idx = act ? v.list[s] : 0;
        const double x = act ? v.rec[idx].val : 0.0;
        const double kx = single ? weight * fabs(x) : fabs(x);
        const bool ok = act && (single ? kx >= thr : kx > thr);
        const unsigned long long mask = __ballot(ok);
        if (ok) { const int p = cnt + __popcll(mask & lt); A.cand[p] = idx; A.key[p] = kx; }
        cnt += __popcll(mask);
    }
    int off = 0;
    if (cnt > limit) {
        DP_SYNC();
        if (lane == 0 && limit > 0) select_largest(A.key, A.cand, 0, cnt - 1, limit);
        off = cnt - limit;
    }
    DP_SYNC();
    const int nk = cnt - off;
    if (nk <= 64) {
        const unsigned long long sorted = dp_sort64(lane < nk ? (unsigned long long)(unsigned)A.cand[off + lane] : ~0ull, lane);
        if (lane < nk) A.cand[lane] = (int)(unsigned)sorted;
        DP_SYNC();
        return nk;
    }
    int N = 64;
    while (N < nk) N *= 2;
    for (int i = lane; i < N; i += 64) A.sortk[i] = i < nk ? (unsigned long long)(unsigned)A.cand[off + i] : ~0ull;
    DP_SYNC();
    wave_sort_u64<true>(A.sortk, N, lane);
    for (int i = lane; i < nk; i += 64) A.cand[i] = (int)(unsigned)A.sortk[i];
    DP_SYNC();
    return nk;
}

__global__ void __launch_bounds__(64) k_pilucdp(DpArgs A)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n = A.n;
    const SpVec z{A.zrec, A.zlist}, w{A.wrec, A.wlist};
    // the state of the chain between two steps (a launch goes on where the one before it had to stop for a larger store)
    int znnz = A.ctrl[6], wnnz = A.ctrl[7];
    bool eliminate = A.ctrl[4] != 0, end_level_now = false;
    double piv_tol = A.dctrl[1], threshold = A.dctrl[0];
    int last = A.ctrl[1], nA = A.ctrl[2], zero_piv = A.ctrl[3], pos_pivot = -1;
    int pU = A.ctrl[8], pL = A.ctrl[9], pS = A.ctrl[10];
    int prev_pivot = A.ctrl[11];                                                  // the column the step before took as its pivot (-1: none)
    const double nnzA = (double)A.Cp[n];
    const int row_max = (A.max_fill < n ? A.max_fill : n) + 1;                       // what one step can add to a store
#define DP_STOP(code) do { if (lane == 0) { A.ctrl[0] = (code); A.ctrl[1] = last; A.ctrl[2] = nA; A.ctrl[3] = zero_piv; A.ctrl[4] = eliminate ? 1 : 0; \
                                            A.ctrl[5] = k; A.ctrl[6] = znnz; A.ctrl[7] = wnnz; A.ctrl[8] = pU; A.ctrl[9] = pL; A.ctrl[10] = pS; A.ctrl[11] = prev_pivot; \
                                            A.dctrl[0] = threshold; A.dctrl[1] = piv_tol; } return; } while (0)

    for (int k = A.ctrl[5]; k < n; ++k) {
        if ((long)pU + row_max > (long)A.capU) DP_STOP(1);
        if ((long)pL + row_max > (long)A.capL) DP_STOP(2);
        if (!eliminate && (long)pS + row_max > (long)A.capS) DP_STOP(3);
        if (A.begin_total_piv && k == A.bp) piv_tol = 1.0;                          // :448
        const int sel = A.prow[k];                                                  // (2.) :453-466
        // the vectors of the step before are cleared -- but the column that was its pivot and the row of this step are dead from now on
        for (int s = lane; s < znnz; s += 64) { const int c = z.list[s]; if (c != prev_pivot) z.rec[c].slot = -1; }
        for (int s = lane; s < wnnz; s += 64) { const int r = w.list[s]; if (r != sel) w.rec[r].slot = -1; }
        znnz = wnnz = 0;
        prev_pivot = -1;
        if (lane == 0) { A.unused[sel] = 0; w.rec[sel].slot = -2; }
        DP_SYNC();
        {
            const int r0 = A.Ap[sel], r1 = A.Ap[sel + 1];
            for (int base = r0; base < r1; base += 64) {
                const int e = base + lane;
                const bool act = e < r1;
                const int c = act ? A.Ai[e] : -1;
                const int pc = (act && e > r0) ? A.Ai[e - 1] : -1;
                const bool ok = act && A.nonpiv[c] != 0;
                const bool first = ok && c != pc;
                const unsigned long long mask = __ballot(first);
                if (first) { const int s = znnz + __popcll(mask & lt); z.list[s] = c; z.rec[c] = DpRec{A.Av[e], s, 0}; }
                znnz += __popcll(mask);
                unsigned long long dup = __ballot(ok && !first);                     // a column stored twice in the row: the last value stands
                if (dup) {
                    DP_SYNC();
                    if (lane == 0)
                        while (dup) { const int b = __ffsll((long long)dup) - 1; dup &= dup - 1; z.rec[A.Ai[base + b]].val = A.Av[base + b]; }
                }
            }
            DP_SYNC();
        }
        {                                                                           // (3.) :472-487: the rows of U this row has multipliers for
            // the list is walked three nodes ahead: a node's fields, then the pivot and the extent of the row it names, then the first 64 entries
            // of that row are on their way while the rows before it are subtracted (a node costs two dependent round trips -- the records of
            // its columns, the stores -- instead of six)
            DpNode n1 = dp_node(A.colL, A.Lval, A.linkL, A.startL[sel]);
            DpRow a1 = dp_row(A.Dinv, A.Uptr, n1);
            DpEnt t1 = dp_ent(A.Uidx, A.Uval, a1, lane);
            DpNode n2 = dp_node(A.colL, A.Lval, A.linkL, n1.link);
            DpRow a2 = dp_row(A.Dinv, A.Uptr, n2);
            DpNode n3 = dp_node(A.colL, A.Lval, A.linkL, n2.link);
            while (n1.at != -1) {
                const DpEnt t2 = dp_ent(A.Uidx, A.Uval, a2, lane);
                const DpRow a3 = dp_row(A.Dinv, A.Uptr, n3);
                const DpNode n4 = dp_node(A.colL, A.Lval, A.linkL, n3.link);
                const double f = n1.v / a1.dinv;
                dp_subtract(z, znnz, f, A.Uidx, A.Uval, a1.e0, a1.e1, t1.c, t1.v, lane);
                n1 = n2; a1 = a2; t1 = t2; n2 = n3; a2 = a3; n3 = n4;
            }
        }
        double pivot = 0.0;
        if (eliminate) {                                                            // the pivot, :540-558
            double best = 0.0;
            int bslot = 0x7fffffff;
            for (int s = lane; s < znnz; s += 64) { const double v = fabs(z.rec[z.list[s]].val); if (v > best) { best = v; bslot = s; } }
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int os = __shfl_xor(bslot, off);
                if (ob > best || (ob == best && os < bslot)) { best = ob; bslot = os; }
            }
            pos_pivot = bslot == 0x7fffffff ? -1 : z.list[bslot];
            const double val_larg_el = pos_pivot >= 0 ? z.rec[pos_pivot].val : 0.0;
            if (A.nonpiv[sel] != 0) {
                dp_touch(z, znnz, sel, lane);
                const double zs = z.rec[sel].val;
                if (fabs(val_larg_el * piv_tol) > fabs(zs) && pos_pivot >= 0 && A.piv_tol > 0) pivot = val_larg_el;
                else { pos_pivot = sel; pivot = zs; }
            } else {
                if (fabs(val_larg_el) > 0.0 && pos_pivot >= 0) pivot = val_larg_el;
                else { pos_pivot = A.perm[k]; dp_touch(z, znnz, pos_pivot, lane); pivot = z.rec[pos_pivot].val; }
            }
        }
        if (eliminate && !A.force_finish && (double)k > A.min_elim_factor * (double)n && A.small_pivot_terminates && fabs(pivot) < A.min_pivot) {   // :595-612
            eliminate = false;
            end_level_now = true;
            threshold *= A.shift_schur;
            last = k - 1;
            nA = n - k;
        }
        double dinv = 1.0;
        if (eliminate) {                                                            // :613-629
            dinv = 1.0 / pivot;
            for (int s = lane; s < znnz; s += 64) { const int c = z.list[s]; z.rec[c].val = z.rec[c].val * dinv; }
            DP_SYNC();
            if (lane == 0) {
                z.rec[pos_pivot] = DpRec{0.0, -2, 0};                                // (eliminated for the sorting, :619; dead as a column from here on)
                const int pk = A.perm[k], p = A.iperm[pos_pivot];
                const int t = A.iperm[pk]; A.iperm[pk] = A.iperm[pos_pivot]; A.iperm[pos_pivot] = t;
                const int u = A.perm[k]; A.perm[k] = A.perm[p]; A.perm[p] = u;
                A.nonpiv[pos_pivot] = 0;
                A.Dinv[k] = dinv;
            }
            prev_pivot = pos_pivot;
            DP_SYNC();
            {                                                                       // the column of L, :633-651
                const int c = pos_pivot;                                            // = perm[k] now
                const int c0 = A.Cp[c], c1 = A.Cp[c + 1];
                for (int base = c0; base < c1; base += 64) {
                    const int e = base + lane;
                    const bool act = e < c1;
                    const int r = act ? A.Ci[e] : -1;
                    const int pr = (act && e > c0) ? A.Ci[e - 1] : -1;
                    const bool ok = act && A.unused[r] != 0;
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
                DpNode n1 = dp_node(A.rowU, A.Uval, A.linkU, A.startU[c]);
                DpRow a1 = dp_row(A.Dinv, A.Lptr, n1);
                DpEnt t1 = dp_ent(A.Lidx, A.Lval, a1, lane);
                DpNode n2 = dp_node(A.rowU, A.Uval, A.linkU, n1.link);
                DpRow a2 = dp_row(A.Dinv, A.Lptr, n2);
                DpNode n3 = dp_node(A.rowU, A.Uval, A.linkU, n2.link);
                while (n1.at != -1) {
                    const DpEnt t2 = dp_ent(A.Lidx, A.Lval, a2, lane);
                    const DpRow a3 = dp_row(A.Dinv, A.Lptr, n3);
                    const DpNode n4 = dp_node(A.rowU, A.Uval, A.linkU, n3.link);
                    const double f = n1.v / a1.dinv;
                    dp_subtract(w, wnnz, f, A.Lidx, A.Lval, a1.e0, a1.e1, t1.c, t1.v, lane);
                    n1 = n2; a1 = a2; t1 = t2; n2 = n3; a2 = a3; n3 = n4;
                }
            }
            for (int s = lane; s < wnnz; s += 64) { const int r = w.list[s]; w.rec[r].val = w.rec[r].val * dinv; }     // :652
            DP_SYNC();
        }
        // ---- dropping in the row, :714-759 ----
        int nU;
        double n1z = 0.0;
        if (!eliminate) {
            const double norm = sqrt(dp_seq_sum(z, znnz, 1, lane));
            nU = dp_take(A, z, znnz, false, 0.0, norm * threshold, A.max_fill, lane);
        } else {
            const double n2z = (A.rules & PILUC_DROP_STANDARD) ? sqrt(dp_seq_sum(z, znnz, 1, lane)) : 0.0;
            const double n1w = (A.rules & (PILUC_DROP_ERR_PROP | PILUC_DROP_ERR_PROP2)) ? dp_seq_sum(w, wnnz, 0, lane) : 0.0;
            n1z = (A.rules & (PILUC_DROP_ERR_PROP | PILUC_DROP_ERR_PROP2)) ? dp_seq_sum(z, znnz, 0, lane) : 0.0;
            const double weightU = dp_weight(A, n2z, n1w, dinv);
            nU = dp_take(A, z, znnz, true, weightU, threshold, A.max_fill - 1, lane);
        }
        if (eliminate) {                                                            // :761-797: the 1 at the pivot's column, then the list backwards
            const int p0 = pU;
            pU += nU + 1;
            for (int j = lane; j < nU; j += 64) {
                const int pos = p0 + 1 + j, c = A.cand[nU - 1 - j];
                A.Uval[pos] = z.rec[c].val; A.Uidx[pos] = c;
                A.linkU[pos] = A.startU[c]; A.startU[c] = pos; A.rowU[pos] = k;
            }
            if (lane == 0) {
                A.Uval[p0] = 1.0; A.Uidx[p0] = pos_pivot; A.Uptr[k + 1] = p0 + nU + 1;
                if (pivot == 0.0) A.Dinv[k] = 1.0;
            }
            if (pivot == 0.0) { ++zero_piv; dinv = 1.0; }
        } else {                                                                    // :818-847
            const int kA = k - last - 1;
            const int p0 = pU, q0 = pS;
            pU += 1; pS += nU;
            for (int j = lane; j < nU; j += 64) { const int c = A.cand[nU - 1 - j]; A.Sval[q0 + j] = z.rec[c].val; A.Sidx[q0 + j] = c; }
            if (lane == 0) {
                A.Uval[p0] = 1.0; A.Uidx[p0] = A.perm[k]; A.Uptr[k + 1] = p0 + 1; A.Dinv[k] = 1.0;
                A.Sptr[kA + 1] = q0 + nU;
            }
        }
        DP_SYNC();
        // ---- the column of L, :849-1005 ----
        if (eliminate) {
            const double n2w = (A.rules & PILUC_DROP_STANDARD) ? sqrt(dp_seq_sum(w, wnnz, 1, lane)) : 0.0;
            const double weightL = dp_weight(A, n2w, n1z, dinv);
            const int nL = dp_take(A, w, wnnz, true, weightL, threshold, A.max_fill, lane);
            const int p0 = pL;
            pL += nL + 1;
            for (int j = lane; j < nL; j += 64) {
                const int pos = p0 + 1 + j, b = A.cand[j];
                A.Lval[pos] = w.rec[b].val; A.Lidx[pos] = b;
                A.linkL[pos] = A.startL[b]; A.startL[b] = pos; A.colL[pos] = k;
            }
            if (lane == 0) { A.Lval[p0] = 1.0; A.Lidx[p0] = sel; A.Lptr[k + 1] = p0 + nL + 1; }
            DP_SYNC();
            // the rows by their number of entries in L, one move per new entry and in the order of the entries (:964-970)
            if (lane == 0) {
                for (int j = 0; j < nL; ++j) {
                    const int b0 = A.cand[j];
                    if (b0 < A.bpr || b0 > A.epr) continue;
                    const int b = A.iprow[b0];
                    const int cntb = A.numb[b] + 1;
                    const int a = A.pnum[cntb] - 1;
                    A.pnum[cntb] = a;
                    if (a == b) { A.numb[b] = cntb; continue; }
                    const int ra = A.prow[a], rb = A.prow[b];
                    A.iprow[ra] = b; A.iprow[rb] = a;
                    A.prow[a] = rb; A.prow[b] = ra;
                    const int na = A.numb[a];
                    A.numb[a] = cntb; A.numb[b] = na;
                }
            }
            DP_SYNC();
            // a new group of rows with equally many entries begins behind this step: by increasing row index (:980-981; the reference's
            // quicksort_with_inverse leaves the rows -- all different -- in ascending order, and so does any sort)
            const int nk = A.numb[k];
            const int g0 = A.pnum[nk + 1];
            if (g0 == k + 1) {
                const int g1 = A.pnum[nk + 2] - 1;
                const int len = g1 - g0 + 1;
                if (len > 1 && len <= 64) {
                    const unsigned long long sorted = dp_sort64(lane < len ? (unsigned long long)(unsigned)A.prow[g0 + lane] : ~0ull, lane);
                    if (lane < len) { const int r = (int)(unsigned)sorted; A.prow[g0 + lane] = r; A.iprow[r] = g0 + lane; }
                    DP_SYNC();
                } else if (len > 1) {
                    int N = 64;
                    while (N < len) N *= 2;
                    for (int i = lane; i < N; i += 64) A.sortk[i] = i < len ? (unsigned long long)(unsigned)A.prow[g0 + i] : ~0ull;
                    DP_SYNC();
                    wave_sort_u64<true>(A.sortk, N, lane);
                    for (int i = lane; i < len; i += 64) { const int r = (int)(unsigned)A.sortk[i]; A.prow[g0 + i] = r; A.iprow[r] = g0 + i; }
                    DP_SYNC();
                }
            }
            // ---- does the level end here?  :1018-1092 ----
            if (!A.force_finish && (double)k > A.min_elim_factor * (double)n) {
                const double cnt = (double)nk;
                switch (A.final_row_crit) {
                case -1: end_level_now = cnt > (A.move_level_factor * nnzA) / (double)n; break;
                case 0: end_level_now = cnt > (0.5 * nnzA) / (double)n; break;
                case 1: end_level_now = cnt > nnzA / (double)n; break;
                case 2: end_level_now = cnt > (2.0 * nnzA) / (double)n; break;
                case 3: end_level_now = cnt > (4.0 * nnzA) / (double)n; break;
                case 4: end_level_now = cnt > (6.0 * nnzA) / (double)n; break;
                case 5: end_level_now = nk > 10; break;
                case 6: end_level_now = cnt > (1.5 * nnzA) / (double)n; break;
                case 7: end_level_now = sqrt(dp_seq_sum(z, znnz, 1, lane)) > A.row_u_max; break;
                case 8: end_level_now = cnt > (3.0 * nnzA) / (double)n; break;
                case 9: end_level_now = cnt > (1.2 * nnzA) / (double)n; break;
                default: break;
                }
                if (end_level_now) {
                    eliminate = false;
                    threshold *= A.shift_schur;
                    last = k;
                    nA = n - k - 1;
                }
            }
        } else {
            const int p0 = pL;
            pL += 1;
            if (lane == 0) { A.Lval[p0] = 1.0; A.Lidx[p0] = sel; A.Lptr[k + 1] = p0 + 1; }
            DP_SYNC();
        }
    }
    if (lane == 0) { A.ctrl[0] = 0; A.ctrl[1] = last; A.ctrl[2] = nA; A.ctrl[3] = zero_piv; A.ctrl[4] = eliminate ? 1 : 0; A.ctrl[5] = n; }
#undef DP_STOP
}

// ---------------------------------------------- the stores -> matrices ----------------------------------------------
__global__ void k_dp_init(int32_t n, int32_t epr, int32_t *perm, int32_t *iperm, int32_t *prow, int32_t *iprow, int32_t *numb, int32_t *pnum,
                          int32_t *nonpiv, int32_t *unused, int32_t *startU, int32_t *startL, DpRec *zrec, DpRec *wrec, double *Dinv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        perm[i] = iperm[i] = prow[i] = iprow[i] = i;                               // :408-413
        numb[i] = 0; nonpiv[i] = 1; unused[i] = 1; startU[i] = -1; startL[i] = -1; zrec[i] = DpRec{0.0, -1, 0}; wrec[i] = DpRec{0.0, -1, 0}; Dinv[i] = 1.0;
    }
    if (i < n + 2) pnum[i] = i == 0 ? 0 : epr + 1;                                  // :417, :437
}

// entries of a segment that compress() keeps (|x| > 0: sparse_implementation.h:3704-3726)
__global__ void k_dp_count(int32_t nseg, const int32_t *__restrict__ ptr, const double *__restrict__ val, int32_t *__restrict__ len)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nseg) return;
    int c = 0;
    if (r < nseg) for (int j = ptr[r]; j < ptr[r + 1]; ++j) c += fabs(val[j]) > 0.0 ? 1 : 0;
    len[r] = c;
}
__global__ void k_dp_keys(int32_t nseg, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
                          const int32_t *__restrict__ nptr, const int32_t *__restrict__ map, int32_t shift, unsigned long long *__restrict__ keys,
                          double *__restrict__ vals)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nseg) return;
    int q = nptr[r];
    for (int j = ptr[r]; j < ptr[r + 1]; ++j) {
        if (!(fabs(val[j]) > 0.0)) continue;
        keys[q] = ((unsigned long long)(unsigned)r << 32) | (unsigned)(map[idx[j]] - shift);
        vals[q] = val[j];
        ++q;
    }
}
__global__ void k_dp_low(int64_t nnz, const unsigned long long *__restrict__ keys, int32_t *__restrict__ idx)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nnz) idx[j] = (int32_t)(unsigned)keys[j];
}

// compress(), then every index through `map` (minus shift), then normal_order(): one segmented sort
int seg_compress_sort(hipStream_t st, int32_t nseg, const int32_t *ptr, const int32_t *idx, const double *val, const int32_t *map, int32_t shift,
                      bool is_csr, DevMat *M)
{
    PoolBlock b_len, b_k0, b_k1, b_v0, b_tmp;
    ILUPP_HIP(b_len.alloc(sizeof(int32_t) * (size_t)(nseg + 1)));
    M->release();
    M->n = nseg; M->is_csr = is_csr; M->owns = true;
    ILUPP_HIP(pool_malloc(&M->ptr, sizeof(int32_t) * (size_t)(nseg + 1)));
    hipLaunchKernelGGL(k_dp_count, dim3((nseg + 256) / 256), dim3(256), 0, st, nseg, ptr, val, b_len.as<int32_t>());
    {
        size_t tb = 0;
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, b_len.as<int32_t>(), M->ptr, nseg + 1, st));
        ILUPP_HIP(b_tmp.alloc(tb > 0 ? tb : 1));
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tb, b_len.as<int32_t>(), M->ptr, nseg + 1, st));
    }
    int32_t nnz = 0;
    ILUPP_HIP(hipMemcpyAsync(&nnz, M->ptr + nseg, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    M->nnz = nnz;
    ILUPP_HIP(pool_malloc(&M->idx, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
    ILUPP_HIP(pool_malloc(&M->val, sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
    if (nnz > 0) {
        ILUPP_HIP(b_k0.alloc(sizeof(unsigned long long) * (size_t)nnz));
        ILUPP_HIP(b_k1.alloc(sizeof(unsigned long long) * (size_t)nnz));
        ILUPP_HIP(b_v0.alloc(sizeof(double) * (size_t)nnz));
        hipLaunchKernelGGL(k_dp_keys, dim3((nseg + 255) / 256), dim3(256), 0, st, nseg, ptr, idx, val, M->ptr, map, shift, b_k0.as<unsigned long long>(),
                           b_v0.as<double>());
        size_t tb = 0;
        ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, b_k0.as<unsigned long long>(), b_k1.as<unsigned long long>(), b_v0.as<double>(), M->val,
                                                     (int)nnz, 0, 64, st));
        PoolBlock b_t2;
        ILUPP_HIP(b_t2.alloc(tb > 0 ? tb : 1));
        ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(b_t2.p, tb, b_k0.as<unsigned long long>(), b_k1.as<unsigned long long>(), b_v0.as<double>(), M->val,
                                                     (int)nnz, 0, 64, st));
        hipLaunchKernelGGL(k_dp_low, dim3((unsigned)(((int64_t)nnz + 255) / 256)), dim3(256), 0, st, (int64_t)nnz, b_k1.as<unsigned long long>(), M->idx);
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    return ILUPP_OK;
}

int pilucdp_level(hipStream_t st, const DevMat &Arow, const PilucParams &P, bool force_finish, double tau, int32_t bp, int32_t bpr, int32_t epr,
                  DevMat *L, DevMat *U, double **Dinv_out, DevMat *Anew, int32_t *pc2, int32_t *pr2, float *kernel_ms)
{
    const int32_t n = Arow.n;
    const int64_t nnz = Arow.nnz;
    const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    if (epr < 0) epr = 0;                                                           // :372-375
    if (epr >= n) epr = n - 1;
    if (bpr < 0) bpr = 0;
    if (bpr >= n) bpr = n - 1;
    int32_t max_fill = P.max_fill_in > 0 ? P.max_fill_in : n;                        // :352-355
    if (max_fill < 1) max_fill = 1;
    if (max_fill > n) max_fill = n;
    struct MatGuard { DevMat m; ~MatGuard() { m.release(); } } gc;
    transpose_storage(st, Arow, &gc.m);                                             // Akcol = Akrow.change_orientation(), :1460-1461
    const DevMat &Acol = gc.m;

    int sortN = 64;
    while (sortN < n) sortN *= 2;
    PoolBlock b_i, b_d, b_sort, b_ctrl;
    // int arrays of n (+2) entries: perm iperm prow iprow numb pnum nonpiv unused startU startL - zlist - wlist cand Uptr Lptr Sptr
    const size_t slot = ((size_t)n + 64) & ~(size_t)15;
    ILUPP_HIP(b_i.alloc(sizeof(int32_t) * slot * 18));
    ILUPP_HIP(b_d.alloc(sizeof(double) * slot * 5));                                // key, the records of z and of w (16 bytes each)
    ILUPP_HIP(b_sort.alloc(sizeof(unsigned long long) * (size_t)sortN));
    ILUPP_HIP(b_ctrl.alloc(64));
    int32_t *I = b_i.as<int32_t>();
    auto iarr = [&](int q) { return I + slot * (size_t)q; };
    double *Dinv = nullptr;
    ILUPP_HIP(pool_malloc(&Dinv, sizeof(double) * (size_t)n));
    struct DinvGuard { double **p; bool keep = false; ~DinvGuard() { if (!keep && *p) { (void)pool_free(*p); *p = nullptr; } } } gd{&Dinv};

    // the stores: what the factors of this level may grow to is not known beforehand.  The kernel stops BETWEEN two steps when a store
    // has no room for another row; the store is doubled (contents copied) and the kernel goes on with that step (the reference's
    // enlarge_fields_keep_data, :763-769)
    struct Store {
        PoolBlock idx, link, who, val;
        int64_t cap = 0;
        bool lists;
        int grow(hipStream_t st, int64_t ncap, int64_t used) {
            PoolBlock ni, nl, nw, nv;
            ILUPP_HIP(ni.alloc(sizeof(int32_t) * (size_t)ncap));
            ILUPP_HIP(nv.alloc(sizeof(double) * (size_t)ncap));
            if (lists) { ILUPP_HIP(nl.alloc(sizeof(int32_t) * (size_t)ncap)); ILUPP_HIP(nw.alloc(sizeof(int32_t) * (size_t)ncap)); }
            if (used > 0) {
                ILUPP_HIP(hipMemcpyAsync(ni.p, idx.p, sizeof(int32_t) * (size_t)used, hipMemcpyDeviceToDevice, st));
                ILUPP_HIP(hipMemcpyAsync(nv.p, val.p, sizeof(double) * (size_t)used, hipMemcpyDeviceToDevice, st));
                if (lists) {
                    ILUPP_HIP(hipMemcpyAsync(nl.p, link.p, sizeof(int32_t) * (size_t)used, hipMemcpyDeviceToDevice, st));
                    ILUPP_HIP(hipMemcpyAsync(nw.p, who.p, sizeof(int32_t) * (size_t)used, hipMemcpyDeviceToDevice, st));
                }
                ILUPP_HIP(hipStreamSynchronize(st));
            }
            idx.swap(ni); val.swap(nv); link.swap(nl); who.swap(nw);
            cap = ncap;
            return ILUPP_OK;
        }
    } SU, SL, SS;
    SU.lists = SL.lists = true; SS.lists = false;
    int64_t cap0 = 2 * nnz + 8 * (int64_t)n + 1024, capS0 = nnz + 2 * (int64_t)n + 1024;
    if (const char *e = getenv("ILUPP_DP_STORE")) { cap0 = capS0 = (int64_t)n + 2 + atol(e); }        // (tests: stores that fill up after a few steps)
    { int rc = SU.grow(st, cap0, 0); if (rc) return rc; rc = SL.grow(st, cap0, 0); if (rc) return rc; rc = SS.grow(st, capS0, 0); if (rc) return rc; }
    PoolBlock b_dctrl;
    ILUPP_HIP(b_dctrl.alloc(64));
    DpArgs a;
    a.n = n;
    a.Ap = Arow.ptr; a.Ai = Arow.idx; a.Av = Arow.val;
    a.Cp = Acol.ptr; a.Ci = Acol.idx; a.Cv = Acol.val;
    a.threshold = tau; a.shift_schur = P.threshold_shift_schur; a.min_pivot = P.min_pivot; a.min_elim_factor = P.min_elim_factor;
    a.piv_tol = P.piv_tol; a.move_level_factor = P.move_level_factor; a.row_u_max = P.row_u_max;
    a.small_pivot_terminates = P.small_pivot_terminates ? 1 : 0; a.force_finish = force_finish ? 1 : 0; a.begin_total_piv = P.begin_total_piv ? 1 : 0;
    a.final_row_crit = P.final_row_crit; a.bp = bp; a.bpr = bpr; a.epr = epr; a.max_fill = max_fill;
    a.rules = P.rules; a.combine = P.combine; a.scale_invdiag = P.scale_invdiag ? 1 : 0;
    for (int q = 0; q < 5; ++q) a.wgt[q] = P.wgt[q];
    a.neutral = P.neutral; a.min_weight = P.min_weight;
    a.perm = iarr(0); a.iperm = iarr(1); a.prow = iarr(2); a.iprow = iarr(3); a.numb = iarr(4); a.pnum = iarr(5);
    a.nonpiv = iarr(6); a.unused = iarr(7); a.startU = iarr(8); a.startL = iarr(9);
    a.zlist = iarr(11); a.wlist = iarr(13); a.cand = iarr(14);
    a.Uptr = iarr(15); a.Lptr = iarr(16); a.Sptr = iarr(17);
    a.Dinv = Dinv;
    a.key = b_d.as<double>();
    a.zrec = reinterpret_cast<DpRec *>(a.key + slot); a.wrec = a.zrec + slot;
    a.sortk = b_sort.as<unsigned long long>();
    a.ctrl = b_ctrl.as<int32_t>();
    a.dctrl = b_dctrl.as<double>();
    {
        int32_t c0[16] = {0};
        c0[1] = n - 1; c0[4] = 1; c0[11] = -1;                                      // last_row_to_eliminate, eliminating, no pivot yet
        const double d0[2] = {tau, P.piv_tol};
        ILUPP_HIP(hipMemcpyAsync(a.ctrl, c0, sizeof(c0), hipMemcpyHostToDevice, st));
        ILUPP_HIP(hipMemcpyAsync(a.dctrl, d0, sizeof(d0), hipMemcpyHostToDevice, st));
        ILUPP_HIP(hipMemsetAsync(a.Uptr, 0, sizeof(int32_t), st));
        ILUPP_HIP(hipMemsetAsync(a.Lptr, 0, sizeof(int32_t), st));
        ILUPP_HIP(hipMemsetAsync(a.Sptr, 0, sizeof(int32_t), st));
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(k_dp_init, dim3((n + 2 + 255) / 256), dim3(256), 0, st, n, epr, a.perm, a.iperm, a.prow, a.iprow, a.numb, a.pnum, a.nonpiv, a.unused,
                       a.startU, a.startL, a.zrec, a.wrec, Dinv);
    int32_t ctrl[16] = {0};
    for (int launch = 0;; ++launch) {
        a.Uidx = SU.idx.as<int32_t>(); a.linkU = SU.link.as<int32_t>(); a.rowU = SU.who.as<int32_t>(); a.Uval = SU.val.as<double>(); a.capU = (int32_t)SU.cap;
        a.Lidx = SL.idx.as<int32_t>(); a.linkL = SL.link.as<int32_t>(); a.colL = SL.who.as<int32_t>(); a.Lval = SL.val.as<double>(); a.capL = (int32_t)SL.cap;
        a.Sidx = SS.idx.as<int32_t>(); a.Sval = SS.val.as<double>(); a.capS = (int32_t)SS.cap;
        hipEvent_t e0, e1;
        ILUPP_HIP(hipEventCreate(&e0)); ILUPP_HIP(hipEventCreate(&e1));
        ILUPP_HIP(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_pilucdp, dim3(1), dim3(64), 0, st, a);
        ILUPP_HIP(hipEventRecord(e1, st));
        ILUPP_HIP(hipMemcpyAsync(ctrl, a.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        float ms = 0.f;
        ILUPP_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (kernel_ms) *kernel_ms += ms;
        if (dbg) fprintf(stderr, "[ilupp] pilucdp: n %d, launch %d (stores of %lld / %lld / %lld): status %d at step %d, %.2f ms\n", n, launch, (long long)SU.cap,
                         (long long)SL.cap, (long long)SS.cap, ctrl[0], ctrl[5], ms);
        if (ctrl[0] == 0) break;
        Store &S = ctrl[0] == 1 ? SU : ctrl[0] == 2 ? SL : SS;
        const int64_t used = ctrl[0] == 1 ? ctrl[8] : ctrl[0] == 2 ? ctrl[9] : ctrl[10];
        if (S.cap >= 0x7ffffff0ll || (launch > 40 && !getenv("ILUPP_DP_STORE"))) { set_error("ILU++ with pivoting: the factors of a level outgrow 2^31 entries"); return ILUPP_ERR_UNSUPPORTED; }
        int64_t ncap = getenv("ILUPP_DP_STORE") ? S.cap + (int64_t)n + 2 + atol(getenv("ILUPP_DP_STORE")) : 2 * S.cap + (int64_t)n + 1024;
        if (ncap > 0x7ffffff0ll) ncap = 0x7ffffff0ll;
        { const int rc = S.grow(st, ncap, used); if (rc) return rc; }
    }
    {
        const int32_t last = ctrl[1], nA = ctrl[2];
        const bool to_the_end = ctrl[4] != 0;
        // compress(), permute(permrows, ROW) / U.permute(perm, COLUMN), :1131-1151
        { const int rc = seg_compress_sort(st, n, a.Lptr, a.Lidx, a.Lval, a.iprow, 0, false, L); if (rc) return rc; }
        { const int rc = seg_compress_sort(st, n, a.Uptr, a.Uidx, a.Uval, a.iperm, 0, true, U); if (rc) return rc; }
        if (to_the_end) {                                                           // :1133
            Anew->release();
            Anew->n = 0; Anew->nnz = 0; Anew->is_csr = true; Anew->owns = true;
            ILUPP_HIP(pool_malloc(&Anew->ptr, sizeof(int32_t)));
            ILUPP_HIP(hipMemsetAsync(Anew->ptr, 0, sizeof(int32_t), st));
            ILUPP_HIP(pool_malloc(&Anew->idx, sizeof(int32_t)));
            ILUPP_HIP(pool_malloc(&Anew->val, sizeof(double)));
        } else {
            const int rc = seg_compress_sort(st, nA, a.Sptr, a.Sidx, a.Sval, a.iperm, last + 1, true, Anew);      // :1136-1145
            if (rc) return rc;
        }
        ILUPP_HIP(hipMemcpyAsync(pc2, a.perm, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
        ILUPP_HIP(hipMemcpyAsync(pr2, a.prow, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    gd.keep = true;
    *Dinv_out = Dinv;
    return ILUPP_OK;
}

}  // namespace ilupp
