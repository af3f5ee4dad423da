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
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

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
    double wgt[7], neutral, min_weight;
    double *inv;       // inverse-based dropping: xU yU vxU vyU xL yL vxL vyL, n each (null without that rule)
    double *wts;       // weighted dropping: weightsU, weightsL, n each (null without those rules)
    int32_t *perm, *iperm, *prow, *iprow, *numb, *pnum;
    int32_t *nonpiv, *unused;
    double *Dinv;
    int32_t *Uptr, *Uidx, *linkU, *rowU, *startU; double *Uval; int32_t capU;
    int32_t *Lptr, *Lidx, *linkL, *colL, *startL; double *Lval; int32_t capL;
    int32_t *Sptr, *Sidx; double *Sval; int32_t capS;   // the Schur complement's rows as they come (column indices of this level)
    struct DpRec *zrec, *wrec; int32_t *zlist, *wlist;     // the two working vectors (below)
    double *key; int32_t *cand; unsigned long long *sortk;
    char *lvmem;       // k_piluc_chain_mem: the working vectors, their tables and the node list (LvMem below), in global memory
#ifdef DP_PROF
    long long *prof;   // shader-clock ticks per phase of a step, summed over the steps (diagnostic build only)
#endif
    int32_t *ctrl;     // [0] status (0 done; 1 / 2 / 3: the store of U / L / the Schur complement has no room for another row: enlarge it and
                       // launch again), [1] last_row_to_eliminate, [2] n_Anew, [3] zero pivots, [4] eliminating (still / to the end),
                       // [5] the step to go on with, [6] / [7] entries of z / w to clear, [8] / [9] entries of U / L so far, [10] of the Schur complement, [11] the pivot column of the step before
    double *dctrl;     // [0] the threshold, [1] the pivot tolerance of the moment
};

// combine() and the weight of a row of U / a column of L, ILUCDP.hpp:717-726 / :905-914 (parameters_implementation.h:526-534)
__device__ double dp_weight(const DpArgs &A, double n2own, double n1other, double dinv, double inv, double accumulated)
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
    if (A.rules & PILUC_DROP_INVERSE) w = comb(w, A.wgt[5] * inv);
    if (A.rules & PILUC_DROP_WEIGHTED) w = comb(w, A.wgt[6] * accumulated);
    if (A.rules & PILUC_DROP_ERR_PROP) w = comb(w, A.wgt[2] * n1other);
    if (A.rules & PILUC_DROP_ERR_PROP2) w = comb(w, A.wgt[3] * n1other / fabs(dinv));
    if (A.rules & PILUC_DROP_PIVOT) w = comb(w, A.wgt[4] * fabs(dinv));
    if (A.scale_invdiag) w = w * fabs(dinv);
    return w;
}

// Inverse-based dropping (ILUCDP.hpp:680-713 for the row of U, :882-916 for the column of L): two estimates x, y of the growth of the
// inverse factor at index pk, built from running products VX, VY over the steps in their order.  v: the scaled working vector of the
// step (its slots through acc.idx(s) / acc.val(s), in insertion order -- the two 1-norms below are summed in that order, as the
// reference does; the counts are order-free).  Returns max(|x[pk]|, |y[pk]|), the factor of the dropping weight.
template <class Acc>
__device__ double dp_inverse_update(const Acc &acc, int nnz, int k, int pk, double *X, double *Y, double *VX, double *VY, int lane)
{
    auto smax = [](double a, double b) { return a < b ? b : a; };                // std::max
    if (k == 0) {
        for (int s = lane; s < nnz; s += 64) { const int c = acc.idx(s); const double x = acc.val(s); VX[c] = x; VY[c] = x; }
        if (lane == 0) { X[pk] = 1.0; Y[pk] = 1.0; }
        return 1.0;
    }
    const double vxp = VX[pk], vyp = VY[pk];
    const double xplus = 1.0 - vxp, xminus = -1.0 - vxp, yplus = 1.0 - vyp, yminus = -1.0 - vyp;
    double nuplus = 0.0, numinus = 0.0;
    int nplus = 0, nminus = 0;
    for (int base = 0; base < nnz; base += 64) {
        const int s = base + lane;
        const bool act = s < nnz;
        const int c = act ? acc.idx(s) : 0;
        const double zv = act ? acc.val(s) : 0.0;
        const double vx = act ? VX[c] : 0.0, vi = act ? VY[c] : 0.0;
        const double tp = act ? fabs(vx + zv * xplus) : 0.0, tm = act ? fabs(vx + zv * xminus) : 0.0;
        const int cnt = nnz - base < 64 ? nnz - base : 64;
        for (int i = 0; i < cnt; ++i) { nuplus = nuplus + wv_f64(tp, i); numinus = numinus + wv_f64(tm, i); }
        const double yp = fabs(vi + zv * yplus), ym = fabs(vi + zv * yminus), lim = smax(2.0 * fabs(vi), 0.5);
        nplus += __popcll(__ballot(act && yp > lim)) - __popcll(__ballot(act && smax(2.0 * yp, 0.5) < fabs(vi)));
        nminus += __popcll(__ballot(act && ym > lim)) - __popcll(__ballot(act && smax(2.0 * ym, 0.5) < fabs(vi)));
    }
    const double xk = nuplus > numinus ? xplus : xminus, yk = nplus > nminus ? yplus : yminus;
    for (int s = lane; s < nnz; s += 64) {
        const int c = acc.idx(s);
        const double zv = acc.val(s);
        VX[c] = VX[c] + zv * xk;
        VY[c] = VY[c] + zv * yk;
    }
    const double xe = smax(fabs(xplus), fabs(xminus)), ye = smax(fabs(yplus), fabs(yminus));
    if (lane == 0) { X[pk] = xe; Y[pk] = ye; }
    return smax(fabs(xe), fabs(ye));
}
// weighted dropping (ILUCDP.hpp:629-631, :670-674): every index of the step's vector collects the magnitude of its entry (one add per index
// and step, the steps in their order).  Returns what the accumulated weight of `pk` is after it (pk_in: pk is an index of the vector and
// gets + pk_abs, the magnitude of its entry -- the zeroed pivot of a row of U: +|0|).
template <class Acc>
__device__ double dp_accumulate_weights(const Acc &acc, int nnz, double *W, int pk, bool pk_in, int lane, double pk_abs = 0.0)
{
    const double old = W[pk];
    for (int s = lane; s < nnz; s += 64) { const int c = acc.idx(s); W[c] = W[c] + fabs(acc.val(s)); }
    return pk_in ? old + pk_abs : old;
}
struct SpAcc { SpVec v; __device__ int idx(int s) const { return v.list[s]; } __device__ double val(int s) const { return v.rec[v.list[s]].val; } };

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
        const unsigned long long sorted = dp_sort_n(lane < nk ? (unsigned long long)(unsigned)A.cand[off + lane] : ~0ull, nk, lane);
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

#ifdef DP_PROF
    long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = (long long)__builtin_amdgcn_s_memtime();
#define DP_T(i) do { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pt[i] += now_ - t_last; t_last = now_; } while (0)
#else
#define DP_T(i) do {} while (0)
#endif
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
        DP_T(0);
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
        DP_T(1);
        double pivot = 0.0;
        if (eliminate) {                                                            // the pivot, :540-558
            double best = 0.0;
            int bslot = 0x7fffffff;
            for (int s = lane; s < znnz; s += 64) { const double v = fabs(z.rec[z.list[s]].val); if (v > best) { best = v; bslot = s; } }
            bslot = wv_argmax_first(best, bslot);
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
        DP_T(2);
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
            DP_T(3);
        }
        double wtdU = 0.0, wtdL = 0.0;
        if (A.wts) { wtdU = dp_accumulate_weights(SpAcc{z}, znnz, A.wts, eliminate ? pos_pivot : 0, eliminate, lane); DP_SYNC(); }      // :629-631
        if (eliminate) {
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
                DP_T(4);
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
            DP_T(5);
            for (int s = lane; s < wnnz; s += 64) { const int r = w.list[s]; w.rec[r].val = w.rec[r].val * dinv; }     // :652
            DP_SYNC();
            DP_T(6);
            if (A.wts) { wtdL = dp_accumulate_weights(SpAcc{w}, wnnz, A.wts + n, sel, false, lane); DP_SYNC(); }               // :670-674
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
            double invU = 0.0;
            if (A.rules & PILUC_DROP_INVERSE) {
                invU = dp_inverse_update(SpAcc{z}, znnz, k, pos_pivot, A.inv, A.inv + n, A.inv + 2 * (size_t)n, A.inv + 3 * (size_t)n, lane);
                DP_SYNC();
            }
            const double weightU = dp_weight(A, n2z, n1w, dinv, invU, wtdU);
            nU = dp_take(A, z, znnz, true, weightU, threshold, A.max_fill - 1, lane);
        }
        DP_T(7);
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
        DP_T(8);
        // ---- the column of L, :849-1005 ----
        if (eliminate) {
            const double n2w = (A.rules & PILUC_DROP_STANDARD) ? sqrt(dp_seq_sum(w, wnnz, 1, lane)) : 0.0;
            double invL = 0.0;
            if (A.rules & PILUC_DROP_INVERSE) {
                invL = dp_inverse_update(SpAcc{w}, wnnz, k, sel, A.inv + 4 * (size_t)n, A.inv + 5 * (size_t)n, A.inv + 6 * (size_t)n, A.inv + 7 * (size_t)n, lane);
                DP_SYNC();
            }
            const double weightL = dp_weight(A, n2w, n1z, dinv, invL, wtdL);
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
            DP_T(9);
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
            DP_T(10);
            // a new group of rows with equally many entries begins behind this step: by increasing row index (:980-981; the reference's
            // quicksort_with_inverse leaves the rows -- all different -- in ascending order, and so does any sort)
            const int nk = A.numb[k];
            const int g0 = A.pnum[nk + 1];
            if (g0 == k + 1) {
                const int g1 = A.pnum[nk + 2] - 1;
                const int len = g1 - g0 + 1;
                if (len > 1 && len <= 64) {
                    const unsigned long long sorted = dp_sort_n(lane < len ? (unsigned long long)(unsigned)A.prow[g0 + lane] : ~0ull, len, lane);
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
            DP_T(11);
        } else {
            const int p0 = pL;
            pL += 1;
            if (lane == 0) { A.Lval[p0] = 1.0; A.Lidx[p0] = sel; A.Lptr[k + 1] = p0 + 1; }
            DP_SYNC();
        }
    }
#ifdef DP_PROF
    if (lane == 0) for (int i = 0; i < 12; ++i) A.prof[i] += pt[i];
#endif
    if (lane == 0) { A.ctrl[0] = 0; A.ctrl[1] = last; A.ctrl[2] = nA; A.ctrl[3] = zero_piv; A.ctrl[4] = eliminate ? 1 : 0; A.ctrl[5] = n; }
#undef DP_STOP
}

// =====================================================================================================================================
// The same chain with the two working vectors in LDS (k_pilucdp_lds).  The kernel above keeps z and w in HBM / L2: every phase of a step
// (clear, subtract, pivot search, scale, norms, dropping, sort) is a handful of dependent round trips, ~100 per step.  Here a working
// vector is a slot-ordered pair of LDS arrays (index, value: the reference's insertion order is the slot order) with an LDS hash table
// index -> slot; all of those phases run at LDS latency, and what is left of the trips to memory is what the step really depends on:
// the row / column of A, the walk along the list of multipliers (one trip per node, the fields of the next nodes, their rows' extents,
// entries and liveness flags read ahead), the list heads for the new row and column, and the bucket moves of the rows whose count
// grew -- prefetched for all new entries at once and applied in parallel when no two of them touch the same position (the usual case;
// otherwise one lane applies them in order).  Global side effects of a step happen only after both vectors are complete: a step whose
// vector outgrows its LDS capacity is abandoned untouched (status 4) and the chain goes on in the kernel above.
constexpr int kLvCap = 2048;                    // entries a working vector may hold
constexpr int kLvHash = 4096;                   // slots of its index -> slot table (linear probing, at most half full)
constexpr int kPnL = 1024;                      // bucket boundaries (pnum) cached in LDS: counts below this

// where a chain's working vectors live.  LvLds: in LDS (the kernels below).  LvMem: the same code on global memory, for the steps of partialILUC
// whose vectors outgrow LDS (k_piluc_chain_mem) -- the arrays are private to the wave (one CU, one L1: plain accesses are coherent once the stores
// are acknowledged); only the table's keys are entered with an atomic (which works at L2) and therefore read past L1.
struct LvLds {
    static constexpr int cap = kLvCap, hash = kLvHash, hbits = 12, nodes = 4096;
    static constexpr bool mem = false;
    static __device__ __forceinline__ void sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
};
struct LvMem {
    static constexpr int cap = 32768, hash = 65536, hbits = 16, nodes = 65536;
    static constexpr bool mem = true;
    static __device__ __forceinline__ void sync() { __builtin_amdgcn_s_waitcnt(0); asm volatile("" ::: "memory"); }
};
static_assert((1 << LvLds::hbits) == LvLds::hash && (1 << LvMem::hbits) == LvMem::hash && LvMem::cap <= 65536, "table sizes");

struct LdsVec { int32_t *idx; double *val; int32_t *hkey; unsigned short *hslot; };
struct LvAcc { LdsVec v; __device__ int idx(int s) const { return v.idx[s]; } __device__ double val(int s) const { return v.val[s]; } };

#define LV_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")        // one wave: LDS operations complete in order; this orders the compiler

template <class K = LvLds> __device__ __forceinline__ unsigned lv_hash(int c) { return ((unsigned)c * 0x9E3779B1u) >> (32 - K::hbits); }
template <class K = LvLds> __device__ __forceinline__ int lv_find(const LdsVec &v, int c)
{
    unsigned h = lv_hash<K>(c);
    for (;;) {
        int k;
        if constexpr (K::mem) k = ld_agent_i32(&v.hkey[h]); else k = v.hkey[h];
        if (k == c) return v.hslot[h];
        if (k == -1) return -1;
        h = (h + 1) & (K::hash - 1);
    }
}
// c is not in the table; other lanes may be entering other indices at the same time
template <class K = LvLds> __device__ __forceinline__ void lv_enter(const LdsVec &v, int c, int slot)
{
    unsigned h = lv_hash<K>(c);
    for (;;) {
        if (atomicCAS(&v.hkey[h], -1, c) == -1) { v.hslot[h] = (unsigned short)slot; return; }
        h = (h + 1) & (K::hash - 1);
    }
}
template <class K = LvLds> __device__ __forceinline__ void lv_clear(const LdsVec &v, int lane)
{
    int4 *t = reinterpret_cast<int4 *>(v.hkey);
    for (int i = lane; i < K::hash / 4; i += 64) t[i] = make_int4(-1, -1, -1, -1);
}
// v[c] exists afterwards (operator[] inserts a zero); false: no room
template <class K = LvLds> __device__ __forceinline__ bool lv_touch(const LdsVec &v, int &nnz, int c, int lane)
{
    if (lv_find<K>(v, c) >= 0) return true;
    if (nnz >= K::cap) return false;
    if (lane == 0) { v.idx[nnz] = c; v.val[nnz] = 0.0; lv_enter<K>(v, c, nnz); }
    ++nnz;
    K::sync();
    return true;
}
// the row / column `who` of A (entries e0 .. e1 of ai / av) into an empty vector: entries whose index is alive (and is not `dead`), a
// doubly stored index keeps its last value; false: no room
// this lane's entry of the first 64 of a row / column of A with its liveness flag, fetched ahead
struct OwnPipe { int c, pc, live; double x; };
__device__ __forceinline__ void op_entries(OwnPipe &o, const int32_t *ai, const double *av, int e0, int e1, int lane)
{
    const int e = e0 + lane;
    o.c = -1; o.pc = -1; o.x = 0.0; o.live = 0;
    if (e < e1) { o.c = ai[e]; o.x = av[e]; if (e > e0) o.pc = ai[e - 1]; }
}
__device__ __forceinline__ void op_live(OwnPipe &o, const int32_t *alive) { if (o.c >= 0) o.live = alive[o.c]; }

__device__ __forceinline__ bool lv_load(const LdsVec &v, int &nnz, const int32_t *ai, const double *av, int e0, int e1, const int32_t *alive, int dead, int lane,
                                        const OwnPipe *pre = nullptr)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = e0; base < e1; base += 64) {
        const int e = base + lane;
        const bool act = e < e1;
        int c, pc, live; double x;
        if (pre && base == e0) { c = pre->c; pc = pre->pc; x = pre->x; live = pre->live; }
        else { c = act ? ai[e] : -1; pc = (act && e > e0) ? ai[e - 1] : -1; x = act ? av[e] : 0.0; live = act ? alive[c] : 0; }
        const bool ok = act && c != dead && live != 0;
        const bool first = ok && c != pc;
        const unsigned long long mask = __ballot(first);
        if (nnz + __popcll(mask) > kLvCap) return false;
        if (first) { const int s = nnz + __popcll(mask & lt); v.idx[s] = c; v.val[s] = x; lv_enter(v, c, s); }
        nnz += __popcll(mask);
        const unsigned long long dup = __ballot(ok && !first);
        if (dup) {
            LV_SYNC();
            for (int b = 0; b < 64; ++b)
                if ((dup >> b) & 1ull) { const int cb = wv_i32(c, b); const double xb = wv_f64(x, b); if (lane == 0) v.val[lv_find(v, cb)] = xb; LV_SYNC(); }
        }
    }
    LV_SYNC();
    return true;
}
// sum of |x| (mode 0) or x * x (mode 1) over the slots in order
__device__ __forceinline__ double lv_seq_sum(const LdsVec &v, int nnz, int mode, int lane)
{
    double acc = 0.0;
    for (int base = 0; base < nnz; base += 64) {
        const int s = base + lane;
        const double x = s < nnz ? v.val[s] : 0.0;
        const double t = mode == 0 ? fabs(x) : x * x;
        const int cnt = nnz - base < 64 ? nnz - base : 64;
        for (int i = 0; i < cnt; ++i) acc = acc + wv_f64(t, i);
    }
    return acc;
}

struct DpLive { int ok; };
__device__ __forceinline__ DpLive dp_live(const int32_t *alive, const DpEnt &t, const DpRow &r, int lane)
{
    DpLive l{0};
    if (r.e0 + lane < r.e1) l.ok = alive[t.c];
    return l;
}

// v -= (value of the node / pivot of its row) * (row of the other factor), for every node of the list from `start`, in list order; new
// indices are appended in entry order.  Nodes, extents, entries and liveness flags are read four nodes ahead.  false: no room.
// the read-ahead state of a list walk: four nodes, the extents of three rows, the entries of two, the liveness flags of one
struct ListPipe { DpNode n1, n2, n3, n4; DpRow a1, a2, a3; DpEnt t1, t2; DpLive l1; };
struct ListSrc { const int32_t *who; const double *nval; const int32_t *link; const double *Dinv; const int32_t *ptr; const int32_t *eidx; const double *eval; const int32_t *alive; };
// the four levels of loads that fill a pipe; each level needs what the level before it brought (the caller interleaves the levels of
// several pipes so that their round trips overlap)
__device__ __forceinline__ void lp_level1(ListPipe &p, const ListSrc &S, int start) { p.n1 = dp_node(S.who, S.nval, S.link, start); }
__device__ __forceinline__ void lp_level2(ListPipe &p, const ListSrc &S) { p.a1 = dp_row(S.Dinv, S.ptr, p.n1); p.n2 = dp_node(S.who, S.nval, S.link, p.n1.link); }
__device__ __forceinline__ void lp_level3(ListPipe &p, const ListSrc &S, int lane)
{ p.t1 = dp_ent(S.eidx, S.eval, p.a1, lane); p.a2 = dp_row(S.Dinv, S.ptr, p.n2); p.n3 = dp_node(S.who, S.nval, S.link, p.n2.link); }
__device__ __forceinline__ void lp_level4(ListPipe &p, const ListSrc &S, int lane)
{ p.l1 = dp_live(S.alive, p.t1, p.a1, lane); p.t2 = dp_ent(S.eidx, S.eval, p.a2, lane); p.a3 = dp_row(S.Dinv, S.ptr, p.n3); p.n4 = dp_node(S.who, S.nval, S.link, p.n3.link); }

__device__ __forceinline__ bool lv_subtract_pipe(const LdsVec &v, int &nnz, ListPipe &P, const ListSrc &S, int dead, int lane);

__device__ __forceinline__ bool lv_subtract_list(const LdsVec &v, int &nnz, int start, const int32_t *who, const double *nval, const int32_t *link,
                                                 const double *Dinv, const int32_t *ptr, const int32_t *eidx, const double *eval, const int32_t *alive, int dead,
                                                 int lane)
{
    const ListSrc S{who, nval, link, Dinv, ptr, eidx, eval, alive};
    ListPipe P;
    lp_level1(P, S, start); lp_level2(P, S); lp_level3(P, S, lane); lp_level4(P, S, lane);
    return lv_subtract_pipe(v, nnz, P, S, dead, lane);
}

__device__ __forceinline__ bool lv_subtract_pipe(const LdsVec &v, int &nnz, ListPipe &P, const ListSrc &S, int dead, int lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int32_t *who = S.who, *link = S.link, *ptr = S.ptr, *eidx = S.eidx, *alive = S.alive;
    const double *nval = S.nval, *Dinv = S.Dinv, *eval = S.eval;
    DpNode n1 = P.n1, n2 = P.n2, n3 = P.n3, n4 = P.n4;
    DpRow a1 = P.a1, a2 = P.a2, a3 = P.a3;
    DpEnt t1 = P.t1, t2 = P.t2;
    DpLive l1 = P.l1;
    while (n1.at != -1) {
        const DpLive l2 = dp_live(alive, t2, a2, lane);
        const DpEnt t3 = dp_ent(eidx, eval, a3, lane);
        const DpRow a4 = dp_row(Dinv, ptr, n4);
        const DpNode n5 = dp_node(who, nval, link, n4.link);
        const double f = n1.v / a1.dinv;
        for (int base = a1.e0; base < a1.e1; base += 64) {
            const int e = base + lane;
            const bool act = e < a1.e1;
            int c; double ev; int live;
            if (base == a1.e0) { c = t1.c; ev = t1.v; live = l1.ok; }
            else { c = act ? eidx[e] : 0; ev = act ? eval[e] : 0.0; live = act ? alive[c] : 0; }
            const bool ok = act && live != 0 && c != dead;
            const int slot = ok ? lv_find(v, c) : -1;
            const bool isnew = ok && slot < 0;
            const unsigned long long mask = __ballot(isnew);
            if (nnz + __popcll(mask) > kLvCap) return false;
            if (ok) {
                const double prod = f * ev;
                if (isnew) { const int s = nnz + __popcll(mask & lt); v.idx[s] = c; v.val[s] = 0.0 - prod; lv_enter(v, c, s); }
                else v.val[slot] = v.val[slot] - prod;
            }
            nnz += __popcll(mask);
            LV_SYNC();
        }
        n1 = n2; a1 = a2; t1 = t2; l1 = l2; n2 = n3; a2 = a3; t2 = t3; n3 = n4; a3 = a4; n4 = n5;
    }
    return true;
}

// the slots that pass the dropping rule, in insertion order, at most `limit` of them (the largest keys, by the reference's selection);
// afterwards sortk[0 .. return) = (index << 32 | slot) ascending by index
template <class K = LvLds> __device__ int lv_take(const LdsVec &v, int nnz, bool single, double weight, double thr, int limit, double *key, int32_t *cand, unsigned long long *sortk, int lane,
                       int skip = -1)         // skip: an index that is outside the range the candidates are taken from (partialILUC: [k + 1, n))
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    int cnt = 0;
    for (int base = 0; base < nnz; base += 64) {
        const int s = base + lane;
        const bool act = s < nnz;
        const double x = act ? v.val[s] : 0.0;
        const double kx = single ? weight * fabs(x) : fabs(x);
        const bool ok = act && (single ? kx >= thr : kx > thr) && (skip < 0 || v.idx[s] != skip);
        const unsigned long long mask = __ballot(ok);
        if (ok) { const int p = cnt + __popcll(mask & lt); cand[p] = s; key[p] = kx; }
        cnt += __popcll(mask);
    }
    int off = 0;
    K::sync();
    if (cnt > limit) {
        if (lane == 0 && limit > 0) select_largest(key, cand, 0, cnt - 1, limit);
        off = cnt - limit;
        K::sync();
    }
    const int nk = cnt - off;
    auto keyof = [&](int i) { const int s = cand[off + i]; return ((unsigned long long)(unsigned)v.idx[s] << 32) | (unsigned)s; };
    if (nk <= 64) {
        const unsigned long long sorted = dp_sort_n(lane < nk ? keyof(lane) : ~0ull, nk, lane);
        if (lane < nk) sortk[lane] = sorted;
        K::sync();
        return nk;
    }
    int N = 64;
    while (N < nk) N *= 2;
    for (int i = lane; i < N; i += 64) sortk[i] = i < nk ? keyof(i) : ~0ull;
    K::sync();
    wave_sort_u64<K::mem>(sortk, N, lane);
    K::sync();
    return nk;
}

__device__ __forceinline__ void dp_chain_lds(const DpArgs &A)
{
    __shared__ __attribute__((aligned(16))) int32_t s_zidx[kLvCap], s_widx[kLvCap], s_zh[kLvHash], s_wh[kLvHash], s_cand[kLvCap], s_pnum[kPnL];
    __shared__ __attribute__((aligned(16))) double s_zval[kLvCap], s_wval[kLvCap], s_key[kLvCap];
    __shared__ __attribute__((aligned(16))) unsigned long long s_sort[kLvCap];
    __shared__ unsigned short s_zs[kLvHash], s_ws[kLvHash];
    const int lane = threadIdx.x;
    const int n = A.n;
    const LdsVec z{s_zidx, s_zval, s_zh, s_zs}, w{s_widx, s_wval, s_wh, s_ws};
    int znnz = 0, wnnz = 0;
    bool eliminate = A.ctrl[4] != 0, end_level_now = false;
    double piv_tol = A.dctrl[1], threshold = A.dctrl[0];
    int last = A.ctrl[1], nA = A.ctrl[2], zero_piv = A.ctrl[3], pos_pivot = -1;
    int pU = A.ctrl[8], pL = A.ctrl[9], pS = A.ctrl[10];
    const double nnzA = (double)A.Cp[n];
    const int row_max = (A.max_fill < n ? A.max_fill : n) + 1;
    const int npn = n + 2 < kPnL ? n + 2 : kPnL;
    for (int i = lane; i < npn; i += 64) s_pnum[i] = A.pnum[i];
    LV_SYNC();
#define DPL_STOP(code) do { __builtin_amdgcn_s_waitcnt(0); if (lane == 0) { A.ctrl[0] = (code); A.ctrl[1] = last; A.ctrl[2] = nA; A.ctrl[3] = zero_piv; A.ctrl[4] = eliminate ? 1 : 0; \
                                             A.ctrl[5] = k; A.ctrl[6] = 0; A.ctrl[7] = 0; A.ctrl[8] = pU; A.ctrl[9] = pL; A.ctrl[10] = pS; A.ctrl[11] = -1; \
                                             A.dctrl[0] = threshold; A.dctrl[1] = piv_tol; } return; } while (0)
#ifdef DP_PROF
    long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int k = A.ctrl[5]; k < n; ++k) {
        if ((long)pU + row_max > (long)A.capU) DPL_STOP(1);
        if ((long)pL + row_max > (long)A.capL) DPL_STOP(2);
        if (!eliminate && (long)pS + row_max > (long)A.capS) DPL_STOP(3);
        const double piv_tol_step = (A.begin_total_piv && k == A.bp) ? 1.0 : piv_tol;        // :448 (kept only when the step is)
        const int sel = A.prow[k];                                                  // (2.) :453-466
        const int nk_at_k = A.numb[k];
        const int perm_k = A.perm[k];
        lv_clear(z, lane); lv_clear(w, lane);
        znnz = wnnz = 0;
        // everything the step reads about its row -- and, on the guess that the diagonal will be the pivot, about its column -- is asked for
        // level by level for both at once: the row / column of A with liveness flags, the first four nodes of both multiplier lists
        const ListSrc SL{A.colL, A.Lval, A.linkL, A.Dinv, A.Uptr, A.Uidx, A.Uval, A.nonpiv}, SU{A.rowU, A.Uval, A.linkU, A.Dinv, A.Lptr, A.Lidx, A.Lval, A.unused};
        const int r0 = A.Ap[sel], r1 = A.Ap[sel + 1], head = A.startL[sel];
        const int sel_alive = A.nonpiv[sel];
        const int gc0 = A.Cp[sel], gc1 = A.Cp[sel + 1], gheadU = A.startU[sel];
        OwnPipe zo, wo;
        ListPipe zl, wl;
        op_entries(zo, A.Ai, A.Av, r0, r1, lane); lp_level1(zl, SL, head);
        op_entries(wo, A.Ci, A.Cv, gc0, gc1, lane); lp_level1(wl, SU, gheadU);
        op_live(zo, A.nonpiv); lp_level2(zl, SL);
        op_live(wo, A.unused); lp_level2(wl, SU);
        lp_level3(zl, SL, lane); lp_level3(wl, SU, lane);
        lp_level4(zl, SL, lane); lp_level4(wl, SU, lane);
        LV_SYNC();
        if (!lv_load(z, znnz, A.Ai, A.Av, r0, r1, A.nonpiv, -1, lane, &zo)) DPL_STOP(4);
        DP_T(0);
        // (3.) :472-487: the rows of U this row has multipliers for
        if (!lv_subtract_pipe(z, znnz, zl, SL, -1, lane)) DPL_STOP(4);
        DP_T(1);
        double pivot = 0.0;
        int pslot = -1;
        bool elim_step = eliminate, end_here = false;
        if (elim_step) {                                                            // the pivot, :540-558
            double best = 0.0;
            int bslot = 0x7fffffff;
            for (int s = lane; s < znnz; s += 64) { const double v = fabs(z.val[s]); if (v > best) { best = v; bslot = s; } }
            bslot = wv_argmax_first(best, bslot);
            pos_pivot = bslot == 0x7fffffff ? -1 : z.idx[bslot];
            pslot = bslot == 0x7fffffff ? -1 : bslot;
            const double val_larg_el = pos_pivot >= 0 ? z.val[bslot] : 0.0;
            if (sel_alive != 0) {
                if (!lv_touch(z, znnz, sel, lane)) DPL_STOP(4);
                const int ss = lv_find(z, sel);
                const double zs = z.val[ss];
                if (fabs(val_larg_el * piv_tol_step) > fabs(zs) && pos_pivot >= 0 && A.piv_tol > 0) pivot = val_larg_el;
                else { pos_pivot = sel; pslot = ss; pivot = zs; }
            } else {
                if (fabs(val_larg_el) > 0.0 && pos_pivot >= 0) pivot = val_larg_el;
                else { pos_pivot = perm_k; if (!lv_touch(z, znnz, pos_pivot, lane)) DPL_STOP(4); pslot = lv_find(z, pos_pivot); pivot = z.val[pslot]; }
            }
            if (!A.force_finish && (double)k > A.min_elim_factor * (double)n && A.small_pivot_terminates && fabs(pivot) < A.min_pivot) {   // :595-612
                elim_step = false;
                end_here = true;
            }
        }
        DP_T(2);
        double dinv = 1.0;
        if (elim_step) {                                                            // :613-651, the column of L first (nothing is written before both vectors stand)
            dinv = 1.0 / pivot;
            const int c = pos_pivot;
            if (c == sel) {                                                         // the guess held: the column is here already
                if (!lv_load(w, wnnz, A.Ci, A.Cv, gc0, gc1, A.unused, sel, lane, &wo)) DPL_STOP(4);
                DP_T(3);
                if (!lv_subtract_pipe(w, wnnz, wl, SU, sel, lane)) DPL_STOP(4);
            } else {
                const int c0 = A.Cp[c], c1 = A.Cp[c + 1], headU = A.startU[c];
                if (!lv_load(w, wnnz, A.Ci, A.Cv, c0, c1, A.unused, sel, lane)) DPL_STOP(4);
                DP_T(3);
                if (!lv_subtract_list(w, wnnz, headU, A.rowU, A.Uval, A.linkU, A.Dinv, A.Lptr, A.Lidx, A.Lval, A.unused, sel, lane)) DPL_STOP(4);
            }
        }
        DP_T(4);
        // ---- the step stands: its effects ----
        piv_tol = piv_tol_step;
        if (lane == 0) { A.unused[sel] = 0; A.wrec[sel].slot = -2; }
        if (end_here) {
            eliminate = false;
            end_level_now = true;
            threshold *= A.shift_schur;
            last = k - 1;
            nA = n - k;
        }
        if (elim_step) {
            for (int s = lane; s < znnz; s += 64) z.val[s] = z.val[s] * dinv;
            for (int s = lane; s < wnnz; s += 64) w.val[s] = w.val[s] * dinv;       // :652
            LV_SYNC();
            if (lane == 0) {
                z.val[pslot] = 0.0;                                                  // (eliminated for the sorting, :619; dead as a column from here on)
                A.zrec[pos_pivot] = DpRec{0.0, -2, 0};
                const int p = A.iperm[pos_pivot];
                const int t = A.iperm[perm_k]; A.iperm[perm_k] = p; A.iperm[pos_pivot] = t;
                const int u = A.perm[p]; A.perm[k] = u; A.perm[p] = perm_k;
                A.nonpiv[pos_pivot] = 0;
                A.Dinv[k] = dinv;
            }
            LV_SYNC();
        }
        double wtdU = 0.0, wtdL = 0.0;
        if (A.wts) {                                                                // weighted dropping: :629-631, :670-674
            wtdU = dp_accumulate_weights(LvAcc{z}, znnz, A.wts, elim_step ? pos_pivot : 0, elim_step, lane);
            if (elim_step) wtdL = dp_accumulate_weights(LvAcc{w}, wnnz, A.wts + n, sel, false, lane);
        }
        DP_T(5);
        // ---- dropping in the row, :714-759 ----
        int nU;
        double n1z = 0.0;
        if (!elim_step) {
            const double norm = sqrt(lv_seq_sum(z, znnz, 1, lane));
            nU = lv_take(z, znnz, false, 0.0, norm * threshold, A.max_fill, s_key, s_cand, s_sort, lane);
        } else {
            const double n2z = (A.rules & PILUC_DROP_STANDARD) ? sqrt(lv_seq_sum(z, znnz, 1, lane)) : 0.0;
            const double n1w = (A.rules & (PILUC_DROP_ERR_PROP | PILUC_DROP_ERR_PROP2)) ? lv_seq_sum(w, wnnz, 0, lane) : 0.0;
            n1z = (A.rules & (PILUC_DROP_ERR_PROP | PILUC_DROP_ERR_PROP2)) ? lv_seq_sum(z, znnz, 0, lane) : 0.0;
            double invU = 0.0;
            if (A.rules & PILUC_DROP_INVERSE)
                invU = dp_inverse_update(LvAcc{z}, znnz, k, pos_pivot, A.inv, A.inv + n, A.inv + 2 * (size_t)n, A.inv + 3 * (size_t)n, lane);
            const double weightU = dp_weight(A, n2z, n1w, dinv, invU, wtdU);
            nU = lv_take(z, znnz, true, weightU, threshold, A.max_fill - 1, s_key, s_cand, s_sort, lane);
        }
        DP_T(6);
        if (elim_step) {                                                            // :761-797: the 1 at the pivot's column, then the list backwards
            const int p0 = pU;
            pU += nU + 1;
            for (int j = lane; j < nU; j += 64) {
                const unsigned long long ks = s_sort[nU - 1 - j];
                const int pos = p0 + 1 + j, c = (int)(ks >> 32), sl = (int)(unsigned)ks;
                A.Uval[pos] = z.val[sl]; A.Uidx[pos] = c;
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
            for (int j = lane; j < nU; j += 64) {
                const unsigned long long ks = s_sort[nU - 1 - j];
                A.Sval[q0 + j] = z.val[(int)(unsigned)ks]; A.Sidx[q0 + j] = (int)(ks >> 32);
            }
            if (lane == 0) {
                A.Uval[p0] = 1.0; A.Uidx[p0] = perm_k; A.Uptr[k + 1] = p0 + 1; A.Dinv[k] = 1.0;
                A.Sptr[kA + 1] = q0 + nU;
            }
        }
        LV_SYNC();
        DP_T(7);
        // ---- the column of L, :849-1005 ----
        if (elim_step) {
            const double n2w = (A.rules & PILUC_DROP_STANDARD) ? sqrt(lv_seq_sum(w, wnnz, 1, lane)) : 0.0;
            double invL = 0.0;
            if (A.rules & PILUC_DROP_INVERSE)
                invL = dp_inverse_update(LvAcc{w}, wnnz, k, sel, A.inv + 4 * (size_t)n, A.inv + 5 * (size_t)n, A.inv + 6 * (size_t)n, A.inv + 7 * (size_t)n, lane);
            const double weightL = dp_weight(A, n2w, n1z, dinv, invL, wtdL);
            const int nL = lv_take(w, wnnz, true, weightL, threshold, A.max_fill, s_key, s_cand, s_sort, lane);
            const int p0 = pL;
            pL += nL + 1;
            for (int j = lane; j < nL; j += 64) {
                const unsigned long long ks = s_sort[j];
                const int pos = p0 + 1 + j, b = (int)(ks >> 32);
                A.Lval[pos] = w.val[(int)(unsigned)ks]; A.Lidx[pos] = b;
                A.linkL[pos] = A.startL[b]; A.startL[b] = pos; A.colL[pos] = k;
            }
            if (lane == 0) { A.Lval[p0] = 1.0; A.Lidx[p0] = sel; A.Lptr[k + 1] = p0 + nL + 1; }
            DP_T(8);
            // the rows by their number of entries in L, one move per new entry and in the order of the entries (:964-970): what the moves
            // read is fetched for 64 entries at once; if no move's boundary row is another move's row they do not interact and are applied
            // at once, otherwise (or with a count beyond the cached boundaries) one lane applies them in order
            for (int base = 0; base < nL; base += 64) {
                const int j = base + lane;
                const int b0 = j < nL ? (int)(s_sort[j] >> 32) : -1;
                const bool inr = j < nL && b0 >= A.bpr && b0 <= A.epr;
                const int b = inr ? A.iprow[b0] : -1;
                const int cntb = inr ? A.numb[b] + 1 : 0;
                const int cnt_chunk = nL - base < 64 ? nL - base : 64;
                const bool far = __ballot(inr && cntb >= kPnL) != 0ull;
                int rank = 0;
                for (int i = 0; i < cnt_chunk; ++i) { const int ci = wv_i32(cntb, i); const int ii = wv_i32((int)inr, i); if (i < lane && ii && ci == cntb) ++rank; }
                const int a = (inr && !far) ? s_pnum[cntb] - 1 - rank : -1;
                const int ra = (inr && !far) ? A.prow[a] : -1;
                bool clash = false;
                for (int i = 0; i < cnt_chunk; ++i) { const int bi = wv_i32(b0, i); const int ii = wv_i32((int)inr, i); if (ii && i != lane && inr && ra == bi) clash = true; }
                if (!far && __ballot(clash) == 0ull) {
                    if (inr) {
                        atomicSub(&s_pnum[cntb], 1);
                        if (a != b) {
                            A.iprow[ra] = b; A.iprow[b0] = a;
                            A.prow[a] = b0; A.prow[b] = ra;
                            A.numb[a] = cntb; A.numb[b] = cntb - 1;
                        } else A.numb[b] = cntb;
                    }
                    LV_SYNC();
                    if (inr) A.pnum[cntb] = s_pnum[cntb];
                } else {
                    __builtin_amdgcn_s_waitcnt(0);
                    if (lane == 0) {
                        for (int jj = base; jj < base + cnt_chunk; ++jj) {
                            const int r = (int)(s_sort[jj] >> 32);
                            if (r < A.bpr || r > A.epr) continue;
                            const int bb = A.iprow[r];
                            const int cb = A.numb[bb] + 1;
                            const int aa = (cb < kPnL ? s_pnum[cb] : A.pnum[cb]) - 1;
                            if (cb < kPnL) s_pnum[cb] = aa;
                            A.pnum[cb] = aa;
                            if (aa == bb) { A.numb[bb] = cb; continue; }
                            const int rra = A.prow[aa], rrb = A.prow[bb];
                            A.iprow[rra] = bb; A.iprow[rrb] = aa;
                            A.prow[aa] = rrb; A.prow[bb] = rra;
                            const int na = A.numb[aa];
                            A.numb[aa] = cb; A.numb[bb] = na;
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0);
                LV_SYNC();
            }
            DP_T(9);
            // a new group of rows with equally many entries begins behind this step: by increasing row index (:980-981)
            const int nk = nk_at_k;
            const int g0 = nk + 1 < kPnL ? s_pnum[nk + 1] : A.pnum[nk + 1];
            if (g0 == k + 1) {
                __builtin_amdgcn_s_waitcnt(0);
                const int g1 = (nk + 2 < kPnL ? s_pnum[nk + 2] : A.pnum[nk + 2]) - 1;
                const int len = g1 - g0 + 1;
                if (len > 1 && len <= 64) {
                    const unsigned long long sorted = dp_sort_n(lane < len ? (unsigned long long)(unsigned)A.prow[g0 + lane] : ~0ull, len, lane);
                    if (lane < len) { const int r = (int)(unsigned)sorted; A.prow[g0 + lane] = r; A.iprow[r] = g0 + lane; }
                } else if (len > 1) {
                    int N = 64;
                    while (N < len) N *= 2;
                    for (int i = lane; i < N; i += 64) A.sortk[i] = i < len ? (unsigned long long)(unsigned)A.prow[g0 + i] : ~0ull;
                    DP_SYNC();
                    wave_sort_u64<true>(A.sortk, N, lane);
                    for (int i = lane; i < len; i += 64) { const int r = (int)(unsigned)A.sortk[i]; A.prow[g0 + i] = r; A.iprow[r] = g0 + i; }
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
                case 7: end_level_now = sqrt(lv_seq_sum(z, znnz, 1, lane)) > A.row_u_max; break;
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
        }
        DP_T(10);
        __builtin_amdgcn_s_waitcnt(0);                                              // (what this step stored is what the next one reads)
        DP_T(11);
    }
#ifdef DP_PROF
    if (lane == 0) for (int i = 0; i < 12; ++i) A.prof[i] += pt[i];
#endif
    (void)end_level_now;
    if (lane == 0) { A.ctrl[0] = 0; A.ctrl[1] = last; A.ctrl[2] = nA; A.ctrl[3] = zero_piv; A.ctrl[4] = eliminate ? 1 : 0; A.ctrl[5] = n; }
#undef DPL_STOP
}

__global__ void __launch_bounds__(64) k_pilucdp_lds(DpArgs A) { dp_chain_lds(A); }
// several chains, one workgroup each (the batched construction: BASELINE config 5's "many matrices" shape)
__global__ void __launch_bounds__(64) k_pilucdp_lds_batch(const DpArgs *__restrict__ args)
{
    const DpArgs A = args[blockIdx.x];
    dp_chain_lds(A);
}

// =====================================================================================================================================
// partialILUC -- the factorisation WITHOUT pivoting (ILUCDP.hpp:1405-2231) -- as a chain (k_piluc_chain).  The dataflow kernel of
// piluc_df.hip runs its steps side by side; when the fill makes nearly every step depend on the one before it, and the working rows
// outgrow its LDS classes, it degenerates into a sequential walk through global-memory slots (700 us per step on the critical path).
// This kernel IS a sequential walk, with the working vectors in LDS as above: Crout's three lists (the columns of A, the rows of L, the
// columns of U: ILUC.hpp:31-101 -- initialize / update_sparse_matrix_fields, update_triangular_fields) are kept exactly as the
// reference keeps them, because the ORDER in which they name the contributors is the order of the subtractions.  It also serves the
// dropping rules whose estimates are recurrences over all steps (inverse-based, weighted), which a dataflow kernel cannot run.
// Aliases of DpArgs: perm = listA, iperm = headA, prow = firstA, iprow = listL, numb = firstL, pnum = listU, nonpiv = firstU.
// (LvLds::nodes / LvMem::nodes: the contributors of one step -- the nodes of its lists -- kept for the list update)

// The lists after step k (update_triangular_fields, ILUC.hpp:31-63; update_sparse_matrix_fields, :86-101): every node of list k (in `nodes`,
// in list order) -- for a triangular factor first of all k itself -- advances its `first` and, if it has an entry left, is put at the HEAD
// of the list of that entry's index j: list[h] = tgt[j]; tgt[j] = h (tgt = list for a triangular factor, head for the columns of A).
// The reference does that one node after the other; what depends on the order is only what several nodes with the same j see: the
// first one the old head, every later one the node before it, and the last one becomes the head.  64 nodes at a time, one per lane.
__device__ void pc_relink(int k, bool with_k, const int32_t *ptr, const int32_t *idx, int32_t *list, int32_t *tgt, int32_t *first, const int32_t *nodes, int nn,
                          int lane)
{
    const int total = nn + (with_k ? 1 : 0);
    for (int base = 0; base < total; base += 64) {
        const int t = base + lane;
        const bool act = t < total;
        const int h = !act ? -1 : (with_k ? (t == 0 ? k : nodes[t - 1]) : nodes[t]);
        int f = 0, j = -1, old = -1;
        if (act) {
            f = (with_k && t == 0) ? ptr[k] + 1 : first[h] + 1;
            first[h] = f;
            if (f < ptr[h + 1]) { j = idx[f]; old = tgt[j]; }
        }
        int prev = -1;
        bool last = j >= 0;
        const int cnt = total - base < 64 ? total - base : 64;
        for (int i = 0; i < cnt; ++i) {
            const int ji = wv_i32(j, i), hi = wv_i32(h, i);
            if (j >= 0 && ji == j) { if (i < lane) prev = hi; else if (i > lane) last = false; }
        }
        if (j >= 0) {
            list[h] = prev >= 0 ? prev : old;
            if (last) tgt[j] = h;
        }
        __builtin_amdgcn_s_waitcnt(0);                 // (the next 64 see these heads)
    }
}

// v -= (mval[fmul[h]] / Dinv[h]) * (the entries [fent[h], eptr[h + 1]) of eidx / eval) for every node h of the list from `start`, in list
// order; the nodes go to nodes[base ...].  A node costs one dependent trip (the link to the next one): its fields and the next node's are
// asked for one node ahead, its multiplier and first 64 entries while the node before it is subtracted.  false: out of room.
struct PcNode { int h, next, fm, e0, e1; double dinv; };
struct PcHead { double m; int c; double ev; };
__device__ __forceinline__ PcNode pc_node(int h, const int32_t *list, const int32_t *fmul, const int32_t *fent, const int32_t *eptr, const double *Dinv)
{
    PcNode n{h, -1, 0, 0, 0, 1.0};
    if (h != -1) { n.next = list[h]; n.fm = fmul[h]; n.e0 = fent[h]; n.e1 = eptr[h + 1]; n.dinv = Dinv[h]; }
    return n;
}
__device__ __forceinline__ PcHead pc_head(const PcNode &n, const double *mval, const int32_t *eidx, const double *eval, int lane)
{
    PcHead t{0.0, 0, 0.0};
    if (n.h != -1) { t.m = mval[n.fm]; if (n.e0 + lane < n.e1) { t.c = eidx[n.e0 + lane]; t.ev = eval[n.e0 + lane]; } }
    return t;
}
template <class K>
__device__ __forceinline__ bool pc_subtract_walk(const LdsVec &v, int &nnz, int start, const int32_t *list, const int32_t *fmul, const double *mval,
                                                 const int32_t *fent, const int32_t *eptr, const int32_t *eidx, const double *eval, const double *Dinv,
                                                 int32_t *nodes, int base_node, int &nn, int lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    PcNode a = pc_node(start, list, fmul, fent, eptr, Dinv);
    PcNode b = pc_node(a.next, list, fmul, fent, eptr, Dinv);
    PcHead a2 = pc_head(a, mval, eidx, eval, lane);
    while (a.h != -1) {
        const PcNode c = pc_node(b.next, list, fmul, fent, eptr, Dinv);
        const PcHead b2 = pc_head(b, mval, eidx, eval, lane);
        if (base_node + nn >= K::nodes) return false;
        if (lane == 0) nodes[base_node + nn] = a.h;
        ++nn;
        const double f = a2.m / a.dinv;
        for (int base = a.e0; base < a.e1; base += 64) {
            const int e = base + lane;
            const bool act = e < a.e1;
            int cc; double ev;
            if (base == a.e0) { cc = a2.c; ev = a2.ev; }
            else { cc = act ? eidx[e] : 0; ev = act ? eval[e] : 0.0; }
            const int slot = act ? lv_find<K>(v, cc) : -1;
            const bool isnew = act && slot < 0;
            const unsigned long long mask = __ballot(isnew);
            if (nnz + __popcll(mask) > K::cap) return false;
            if (act) {
                const double prod = f * ev;
                if (isnew) { const int sl = nnz + __popcll(mask & lt); v.idx[sl] = cc; v.val[sl] = 0.0 - prod; lv_enter<K>(v, cc, sl); }
                else v.val[slot] = v.val[slot] - prod;
            }
            nnz += __popcll(mask);
            K::sync();
        }
        a = b; a2 = b2; b = c;
    }
    return true;
}

template <class K>
__device__ __forceinline__ void pc_chain(const DpArgs &A)
{
    int32_t *s_zidx, *s_widx, *s_zh, *s_wh, *s_cand, *s_nodes;
    double *s_zval, *s_wval, *s_key;
    unsigned long long *s_sort;
    unsigned short *s_zs, *s_ws;
    if constexpr (K::mem) {
        char *q = A.lvmem;
        auto take = [&q](size_t bytes) { char *r = q; q += (bytes + 15) & ~(size_t)15; return r; };
        s_zidx = (int32_t *)take(4 * (size_t)K::cap); s_widx = (int32_t *)take(4 * (size_t)K::cap); s_cand = (int32_t *)take(4 * (size_t)K::cap);
        s_zh = (int32_t *)take(4 * (size_t)K::hash); s_wh = (int32_t *)take(4 * (size_t)K::hash); s_nodes = (int32_t *)take(4 * (size_t)K::nodes);
        s_zval = (double *)take(8 * (size_t)K::cap); s_wval = (double *)take(8 * (size_t)K::cap); s_key = (double *)take(8 * (size_t)K::cap);
        s_sort = (unsigned long long *)take(8 * (size_t)K::cap);
        s_zs = (unsigned short *)take(2 * (size_t)K::hash); s_ws = (unsigned short *)take(2 * (size_t)K::hash);
    } else {
        __shared__ __attribute__((aligned(16))) int32_t l_zidx[K::cap], l_widx[K::cap], l_zh[K::hash], l_wh[K::hash], l_cand[K::cap], l_nodes[K::nodes];
        __shared__ __attribute__((aligned(16))) double l_zval[K::cap], l_wval[K::cap], l_key[K::cap];
        __shared__ __attribute__((aligned(16))) unsigned long long l_sort[K::cap];
        __shared__ unsigned short l_zs[K::hash], l_ws[K::hash];
        s_zidx = l_zidx; s_widx = l_widx; s_zh = l_zh; s_wh = l_wh; s_cand = l_cand; s_nodes = l_nodes;
        s_zval = l_zval; s_wval = l_wval; s_key = l_key; s_sort = l_sort; s_zs = l_zs; s_ws = l_ws;
    }
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n = A.n;
    int32_t *const listA = A.perm, *const headA = A.iperm, *const firstA = A.prow, *const listL = A.iprow, *const firstL = A.numb, *const listU = A.pnum,
            *const firstU = A.nonpiv;
    const LdsVec z{s_zidx, s_zval, s_zh, s_zs}, w{s_widx, s_wval, s_wh, s_ws};
    bool eliminate = A.ctrl[4] != 0;
    double threshold = A.dctrl[0];
    int last = A.ctrl[1], nA = A.ctrl[2], zero_piv = A.ctrl[3];
    int pU = A.ctrl[8], pL = A.ctrl[9], pS = A.ctrl[10];
    const int row_max = (A.max_fill < n ? A.max_fill : n) + 1;
#define PC_STOP(code) do { __builtin_amdgcn_s_waitcnt(0); if (lane == 0) { A.ctrl[0] = (code); A.ctrl[1] = last; A.ctrl[2] = nA; A.ctrl[3] = zero_piv; A.ctrl[4] = eliminate ? 1 : 0; \
                                            A.ctrl[5] = k; A.ctrl[8] = pU; A.ctrl[9] = pL; A.ctrl[10] = pS; A.dctrl[0] = threshold; } return; } while (0)
    for (int k = A.ctrl[5]; k < n; ++k) {
        if ((long)pU + row_max > (long)A.capU) PC_STOP(1);
        if ((long)pL + row_max > (long)A.capL) PC_STOP(2);
        if (!eliminate && (long)pS + row_max > (long)A.capS) PC_STOP(3);
        lv_clear<K>(z, lane); lv_clear<K>(w, lane);
        int znnz = 0, wnnz = 0;
        K::sync();
        // (2.) :1575-1583: the row of A from its first entry right of the eliminated columns
        {
            const int e0 = firstA[k], e1 = A.Ap[k + 1];
            for (int base = e0; base < e1; base += 64) {
                const int e = base + lane;
                const bool act = e < e1;
                const int c = act ? A.Ai[e] : -1;
                const int pc = (act && e > e0) ? A.Ai[e - 1] : -1;
                const double x = act ? A.Av[e] : 0.0;
                const bool first = act && c != pc;
                const unsigned long long mask = __ballot(first);
                if (znnz + __popcll(mask) > K::cap) PC_STOP(4);
                if (first) { const int s = znnz + __popcll(mask & lt); z.idx[s] = c; z.val[s] = x; lv_enter<K>(z, c, s); }
                znnz += __popcll(mask);
                const unsigned long long dup = __ballot(act && !first);
                if (dup) {
                    K::sync();
                    for (int b = 0; b < 64; ++b)
                        if ((dup >> b) & 1ull) { const int cb = wv_i32(c, b); const double xb = wv_f64(x, b); if (lane == 0) z.val[lv_find<K>(z, cb)] = xb; K::sync(); }
                }
            }
            K::sync();
        }
        // (3.) :1589-1602: the rows of U this row has multipliers for, in the order of the list
        int nnL = 0;
        if (!pc_subtract_walk<K>(z, znnz, listL[k], listL, firstL, A.Lval, firstU, A.Uptr, A.Uidx, A.Uval, A.Dinv, s_nodes, 0, nnL, lane)) PC_STOP(4);
        // (the reference's z[k] inserts the slot -- in the test below, which it only reaches while eliminating, or in the elimination itself)
        if (eliminate && !lv_touch<K>(z, znnz, k, lane)) PC_STOP(4);
        const int kslot = eliminate ? lv_find<K>(z, k) : -1;
        const double zk = eliminate ? z.val[kslot] : 0.0;
        if (eliminate && !A.force_finish && (double)k > A.min_elim_factor * (double)n && A.small_pivot_terminates && fabs(zk) < A.min_pivot) {
            eliminate = false;
            threshold *= A.shift_schur;
            last = k - 1;
            nA = n - k;
        }
        double pivot = 0.0, dinv = 1.0;
        if (eliminate) {                                                            // :1637-1642
            pivot = zk;
            dinv = 1.0 / pivot;
            for (int s = lane; s < znnz; s += 64) z.val[s] = z.val[s] * dinv;
            K::sync();
            if (lane == 0) z.val[kslot] = 0.0;
            K::sync();
        }
        double wtdU = 0.0, wtdL = 0.0;
        // (8.) :1651-1675: the column of L
        int nnU = 0, nnA = 0;
        if (eliminate) {
            // the rows of A that have an entry in column k, in the order of the list
            for (int h = headA[k]; h != -1;) {
                const int next = listA[h];
                const double x = A.Av[firstA[h]];
                if (nnA >= K::cap) PC_STOP(4);
                if (lane == 0) s_cand[nnA] = h;                                  // (s_cand is free until the dropping: the nodes of A's list)
                ++nnA;
                if (h > k) {
                    if (wnnz >= K::cap) PC_STOP(4);
                    const int slot = lv_find<K>(w, h);
                    if (lane == 0) { if (slot < 0) { w.idx[wnnz] = h; w.val[wnnz] = x; lv_enter<K>(w, h, wnnz); } else w.val[slot] = x; }
                    if (slot < 0) ++wnnz;
                    K::sync();
                }
                h = next;
            }
            if (!pc_subtract_walk<K>(w, wnnz, listU[k], listU, firstU, A.Uval, firstL, A.Lptr, A.Lidx, A.Lval, A.Dinv, s_nodes, nnL, nnU, lane)) PC_STOP(4);
            for (int s = lane; s < wnnz; s += 64) w.val[s] = w.val[s] * dinv;       // w.scale(Dinv[k]), :1665
            K::sync();
        }
        // (status 4 -- a vector or the node list out of room -- leaves the step untouched: every side effect in memory comes after this point,
        // so that the memory flavour can take the chain over AT this step)
        if (eliminate && nnL + nnU + nnA > K::nodes) PC_STOP(4);
        if (A.wts) wtdU = dp_accumulate_weights(LvAcc{z}, znnz, A.wts, k, true, lane);            // :1634-1636 (independent of the column: w)
        if (A.wts && eliminate) {                                                   // :1670-1675 (w may hold row k itself)
            const int ks = lv_find<K>(w, k);
            wtdL = dp_accumulate_weights(LvAcc{w}, wnnz, A.wts + n, k, ks >= 0, lane, ks >= 0 ? fabs(w.val[ks]) : 0.0);
        }
        double invU = 0.0;
        if (eliminate && (A.rules & PILUC_DROP_INVERSE))                            // :1676-1710
            invU = dp_inverse_update(LvAcc{z}, znnz, k, k, A.inv, A.inv + n, A.inv + 2 * (size_t)n, A.inv + 3 * (size_t)n, lane);
        // the nodes of A's list move from s_cand to the tail of s_nodes before the dropping takes s_cand
        if (eliminate) {
            for (int i = lane; i < nnA; i += 64) s_nodes[nnL + nnU + i] = s_cand[i];
            K::sync();
        }
        // ---- dropping, :1716-1764 ----
        int nU;
        double n1z = 0.0;
        if (!eliminate) {
            const double norm = sqrt(lv_seq_sum(z, znnz, 1, lane));
            nU = lv_take<K>(z, znnz, false, 0.0, norm * threshold, A.max_fill, s_key, s_cand, s_sort, lane);
        } else {
            const double n2z = (A.rules & PILUC_DROP_STANDARD) ? sqrt(lv_seq_sum(z, znnz, 1, lane)) : 0.0;
            const double n1w = (A.rules & (PILUC_DROP_ERR_PROP | PILUC_DROP_ERR_PROP2)) ? lv_seq_sum(w, wnnz, 0, lane) : 0.0;
            n1z = (A.rules & (PILUC_DROP_ERR_PROP | PILUC_DROP_ERR_PROP2)) ? lv_seq_sum(z, znnz, 0, lane) : 0.0;
            const double weightU = dp_weight(A, n2z, n1w, dinv, invU, wtdU);
            nU = lv_take<K>(z, znnz, true, weightU, threshold, A.max_fill - 1, s_key, s_cand, s_sort, lane, k);
        }
        double dinv_store = dinv;
        if (eliminate) {                                                            // :1769-1792
            const int p0 = pU;
            pU += nU + 1;
            for (int j = lane; j < nU; j += 64) { const unsigned long long ks = s_sort[j]; A.Uval[p0 + 1 + j] = z.val[(int)(unsigned)ks]; A.Uidx[p0 + 1 + j] = (int)(ks >> 32); }
            if (pivot == 0.0) { ++zero_piv; dinv_store = 1.0; }
            if (lane == 0) { A.Uval[p0] = 1.0; A.Uidx[p0] = k; A.Uptr[k + 1] = p0 + nU + 1; A.Dinv[k] = dinv_store; }
        } else {                                                                    // :1793-1850
            const int kA = k - last - 1;
            const int p0 = pU, q0 = pS;
            pU += 1; pS += nU;
            for (int j = lane; j < nU; j += 64) { const unsigned long long ks = s_sort[j]; A.Sval[q0 + j] = z.val[(int)(unsigned)ks]; A.Sidx[q0 + j] = (int)(ks >> 32); }
            if (lane == 0) { A.Uval[p0] = 1.0; A.Uidx[p0] = k; A.Uptr[k + 1] = p0 + 1; A.Dinv[k] = 1.0; A.Sptr[kA + 1] = q0 + nU; }
        }
        K::sync();
        // (12.) L, :1855-1975
        if (eliminate) {
            double invL = 0.0;
            if (A.rules & PILUC_DROP_INVERSE)
                invL = dp_inverse_update(LvAcc{w}, wnnz, k, k, A.inv + 4 * (size_t)n, A.inv + 5 * (size_t)n, A.inv + 6 * (size_t)n, A.inv + 7 * (size_t)n, lane);
            const double n2w = (A.rules & PILUC_DROP_STANDARD) ? sqrt(lv_seq_sum(w, wnnz, 1, lane)) : 0.0;
            const double weightL = dp_weight(A, n2w, n1z, dinv_store, invL, wtdL);
            const int nL = lv_take<K>(w, wnnz, true, weightL, threshold, A.max_fill - 1, s_key, s_cand, s_sort, lane, k);     // (w may hold row k itself: out of [k + 1, n))
            const int p0 = pL;
            pL += nL + 1;
            for (int j = lane; j < nL; j += 64) { const unsigned long long ks = s_sort[j]; A.Lval[p0 + 1 + j] = w.val[(int)(unsigned)ks]; A.Lidx[p0 + 1 + j] = (int)(ks >> 32); }
            if (lane == 0) { A.Lval[p0] = 1.0; A.Lidx[p0] = k; A.Lptr[k + 1] = p0 + nL + 1; }
        } else {
            const int p0 = pL;
            pL += 1;
            if (lane == 0) { A.Lval[p0] = 1.0; A.Lidx[p0] = k; A.Lptr[k + 1] = p0 + 1; }
        }
        __builtin_amdgcn_s_waitcnt(0);                                              // (the new row and column are what the lists are moved along)
        K::sync();
        // :1977-1983: the three lists move on
        if (eliminate) {
            pc_relink(k, false, A.Ap, A.Ai, listA, headA, firstA, s_nodes + nnL + nnU, nnA, lane);
            pc_relink(k, true, A.Uptr, A.Uidx, listU, listU, firstU, s_nodes + nnL, nnU, lane);
        }
        pc_relink(k, true, A.Lptr, A.Lidx, listL, listL, firstL, s_nodes, nnL, lane);
        __builtin_amdgcn_s_waitcnt(0);
        K::sync();
    }
    if (lane == 0) { A.ctrl[0] = 0; A.ctrl[1] = last; A.ctrl[2] = nA; A.ctrl[3] = zero_piv; A.ctrl[4] = eliminate ? 1 : 0; A.ctrl[5] = n; }
#undef PC_STOP
}
__global__ void __launch_bounds__(64) k_piluc_chain(DpArgs A) { pc_chain<LvLds>(A); }
__global__ void __launch_bounds__(64) k_piluc_chain_mem(DpArgs A) { pc_chain<LvMem>(A); }
__global__ void __launch_bounds__(64) k_piluc_chain_batch(const DpArgs *__restrict__ args) { pc_chain<LvLds>(args[blockIdx.x]); }
__global__ void __launch_bounds__(64) k_piluc_chain_mem_batch(const DpArgs *__restrict__ args) { pc_chain<LvMem>(args[blockIdx.x]); }
constexpr size_t kLvMemBytes = 3 * 4 * (size_t)LvMem::cap + 2 * 4 * (size_t)LvMem::hash + 4 * (size_t)LvMem::nodes + 4 * 8 * (size_t)LvMem::cap + 2 * 2 * (size_t)LvMem::hash + 256;

// ---- launching the chains of a batch together ----
struct ChainBatch {
    std::mutex mu;
    std::condition_variable cv;
    int live = 0;                                   // workers that may still hand in a launch
    struct Item { const DpArgs *args; hipEvent_t before; float ms; bool done; int rc; int kind; };
    std::vector<Item *> waiting;
    hipStream_t stream = nullptr;
    std::string fire_msg;                           // why the last combined launch failed (read by every worker it failed for; under `mu`)
};
static thread_local ChainBatch *t_batch = nullptr;

ChainBatch *chain_batch_create(int workers)
{
    ChainBatch *b = new ChainBatch();
    b->live = workers;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { delete b; return nullptr; }
    return b;
}
void chain_batch_destroy(ChainBatch *b)
{
    if (!b) return;
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}
void chain_batch_enter(ChainBatch *b) { t_batch = b; }

// the kinds of chain a batch combines (the chain with pivoting in LDS; partialILUC in LDS / with its vectors in memory)
enum { CHAIN_DP_LDS = 0, CHAIN_PC_LDS = 1, CHAIN_PC_MEM = 2, CHAIN_KINDS = 3 };

// (mu held) every live worker waits here: their chains as one launch per kind (all launches in the queue before the one wait)
static void chain_batch_fire(ChainBatch *b)
{
    const int cnt = (int)b->waiting.size();
    int rc = ILUPP_OK;
    float ms = 0.f;
    try {
        // the arguments sorted by kind; `off` = where a kind starts
        std::vector<DpArgs> host;
        host.reserve((size_t)cnt);
        int off[CHAIN_KINDS + 1] = {0};
        for (int kind = 0; kind < CHAIN_KINDS; ++kind) {
            for (int i = 0; i < cnt; ++i) if (b->waiting[(size_t)i]->kind == kind) host.push_back(*b->waiting[(size_t)i]->args);
            off[kind + 1] = (int)host.size();
        }
        DpArgs *dev = nullptr;
        ILUPP_HIP(hipMalloc(reinterpret_cast<void **>(&dev), sizeof(DpArgs) * (size_t)cnt));
        struct Free { DpArgs *p; ~Free() { (void)hipFree(p); } } guard{dev};
        ILUPP_HIP(hipMemcpyAsync(dev, host.data(), sizeof(DpArgs) * (size_t)cnt, hipMemcpyHostToDevice, b->stream));
        for (int i = 0; i < cnt; ++i) ILUPP_HIP(hipStreamWaitEvent(b->stream, b->waiting[(size_t)i]->before, 0));
        EventPair ev;
        ILUPP_HIP(ev.create());
        ILUPP_HIP(hipEventRecord(ev.a, b->stream));
        if (off[1] > off[0]) hipLaunchKernelGGL(k_pilucdp_lds_batch, dim3((unsigned)(off[1] - off[0])), dim3(64), 0, b->stream, (const DpArgs *)(dev + off[0]));
        if (off[2] > off[1]) hipLaunchKernelGGL(k_piluc_chain_batch, dim3((unsigned)(off[2] - off[1])), dim3(64), 0, b->stream, (const DpArgs *)(dev + off[1]));
        if (off[3] > off[2]) hipLaunchKernelGGL(k_piluc_chain_mem_batch, dim3((unsigned)(off[3] - off[2])), dim3(64), 0, b->stream, (const DpArgs *)(dev + off[2]));
        ILUPP_HIP(hipEventRecord(ev.b, b->stream));
        ILUPP_HIP(hipStreamSynchronize(b->stream));
        ILUPP_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
        if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] chains of a batch: %d with pivoting, %d + %d of partialILUC (LDS / memory), %.2f ms\n", off[1] - off[0], off[2] - off[1], off[3] - off[2], ms);
    } catch (const HipError &e) { b->fire_msg = std::string("HIP error in the batched chain launch: ") + hipGetErrorString(e.code); rc = ILUPP_ERR_HIP; }
    catch (const std::bad_alloc &) { b->fire_msg = "out of host memory in the batched chain launch"; rc = ILUPP_ERR_MEMORY; }
    catch (...) { b->fire_msg = "unexpected exception in the batched chain launch"; rc = ILUPP_ERR_HIP; }
    // (every waiting worker is released whatever happened above -- one left behind would block its thread, and with it the batch, for good)
    for (ChainBatch::Item *it : b->waiting) { it->ms = ms; it->rc = rc; it->done = true; }
    b->waiting.clear();
    b->cv.notify_all();
}
void chain_batch_leave(ChainBatch *b)
{
    t_batch = nullptr;
    std::unique_lock<std::mutex> lk(b->mu);
    --b->live;
    if (b->live > 0 && (int)b->waiting.size() == b->live) chain_batch_fire(b);
}

// one chain launch: directly, or together with the other chains of the batch this thread works for
static int chain_launch(hipStream_t st, const DpArgs &a, int kind, float *ms)         // kind: CHAIN_*, or -1: the chain with pivoting on global memory (never combined)
{
    ChainBatch *b = t_batch;
    *ms = 0.f;
    if (!b || kind < 0) {
        EventPair ev;
        ILUPP_HIP(ev.create());
        ILUPP_HIP(hipEventRecord(ev.a, st));
        if (kind == CHAIN_DP_LDS) hipLaunchKernelGGL(k_pilucdp_lds, dim3(1), dim3(64), 0, st, a);
        else if (kind == CHAIN_PC_LDS) hipLaunchKernelGGL(k_piluc_chain, dim3(1), dim3(64), 0, st, a);
        else if (kind == CHAIN_PC_MEM) hipLaunchKernelGGL(k_piluc_chain_mem, dim3(1), dim3(64), 0, st, a);
        else hipLaunchKernelGGL(k_pilucdp, dim3(1), dim3(64), 0, st, a);
        ILUPP_HIP(hipEventRecord(ev.b, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        ILUPP_HIP(hipEventElapsedTime(ms, ev.a, ev.b));
        return ILUPP_OK;
    }
    ChainBatch::Item it{&a, nullptr, 0.f, false, ILUPP_OK, kind};
    ILUPP_HIP(hipEventCreateWithFlags(&it.before, hipEventDisableTiming));
    struct DropEvent { hipEvent_t e; ~DropEvent() { (void)hipEventDestroy(e); } } drop{it.before};
    ILUPP_HIP(hipEventRecord(it.before, st));                  // (what this thread queued for the chain: initialisation, enlarged stores)
    {
        std::unique_lock<std::mutex> lk(b->mu);
        b->waiting.push_back(&it);
        if ((int)b->waiting.size() == b->live) chain_batch_fire(b);
        else b->cv.wait(lk, [&] { return it.done; });
    }
    *ms = it.ms;
    if (it.rc != ILUPP_OK) set_error(b->fire_msg);             // (in THIS worker's thread: the message is thread-local, the launch may have been another worker's)
    return it.rc;                                              // (the batch's stream has been synchronised: the chain's results are there)
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

// a store of factor entries that grows: contents copied into a block twice as large (the reference's enlarge_fields_keep_data, :763-769)
struct DpStore {
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
};

__global__ void k_pc_first_cols(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t *__restrict__ fc)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) fc[k] = ptr[k] < ptr[k + 1] ? idx[ptr[k]] : 0;
}
__global__ void k_pc_ones(int32_t n, double *__restrict__ d)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) d[k] = 1.0;
}

// partialILUC as a chain (k_piluc_chain): same results as piluc_level (piluc_df.hip).  +1: a working vector or a list outgrew the kernel's
// LDS capacity -- the caller takes the dataflow kernel's largest class.
int piluc_chain_level(hipStream_t st, const DevMat &Arow, const PilucParams &P, bool force_finish, double tau, DevMat *L, DevMat *U, double **Dinv_out,
                      DevMat *Anew, int32_t *kterm, float *kernel_ms, bool mem_ok)
{
    const int32_t n = Arow.n;
    const int64_t nnz = Arow.nnz;
    const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    int32_t max_fill = P.max_fill_in > 0 ? P.max_fill_in : n;                        // :1440-1447
    if (max_fill < 1) max_fill = 1;
    if (max_fill > n) max_fill = n;
    PoolBlock b_i, b_sort, b_ctrl, b_dctrl, b_inv, b_wts, b_id;
    const size_t slot = ((size_t)n + 64) & ~(size_t)15;
    ILUPP_HIP(b_i.alloc(sizeof(int32_t) * slot * 10));          // listA headA firstA listL firstL listU firstU Uptr Lptr Sptr
    ILUPP_HIP(b_sort.alloc(64));
    ILUPP_HIP(b_ctrl.alloc(64));
    ILUPP_HIP(b_dctrl.alloc(64));
    ILUPP_HIP(b_id.alloc(sizeof(int32_t) * (size_t)(n + 1)));
    int32_t *I = b_i.as<int32_t>();
    auto iarr = [&](int q) { return I + slot * (size_t)q; };
    double *Dinv = nullptr;
    ILUPP_HIP(pool_malloc(&Dinv, sizeof(double) * (size_t)n));
    struct DinvGuard { double **p; bool keep = false; ~DinvGuard() { if (!keep && *p) { (void)pool_free(*p); *p = nullptr; } } } gd{&Dinv};
    DpStore SU, SL, SS;
    SU.lists = SL.lists = SS.lists = false;
    int64_t cap0 = 2 * nnz + 8 * (int64_t)n + 1024, capS0 = nnz + 2 * (int64_t)n + 1024;
    if (const char *e = getenv("ILUPP_DP_STORE")) { cap0 = capS0 = (int64_t)n + 2 + atol(e); }
    { int rc = SU.grow(st, cap0, 0); if (rc) return rc; rc = SL.grow(st, cap0, 0); if (rc) return rc; rc = SS.grow(st, capS0, 0); if (rc) return rc; }
    DpArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    a.Ap = Arow.ptr; a.Ai = Arow.idx; a.Av = Arow.val;
    a.threshold = tau; a.shift_schur = P.threshold_shift_schur; a.min_pivot = P.min_pivot; a.min_elim_factor = P.min_elim_factor;
    a.small_pivot_terminates = P.small_pivot_terminates ? 1 : 0; a.force_finish = force_finish ? 1 : 0; a.max_fill = max_fill;
    a.rules = P.rules; a.combine = P.combine; a.scale_invdiag = P.scale_invdiag ? 1 : 0;
    for (int q = 0; q < 7; ++q) a.wgt[q] = P.wgt[q];
    a.neutral = P.neutral; a.min_weight = P.min_weight;
    if (P.rules & (PILUC_DROP_WEIGHTED | PILUC_DROP_WEIGHTED2)) {
        ILUPP_HIP(b_wts.alloc(sizeof(double) * 2 * (size_t)n));
        a.wts = b_wts.as<double>();
        unsigned long long bits; memcpy(&bits, &P.init_weights_lu, sizeof(bits));
        fill_u64(st, reinterpret_cast<unsigned long long *>(a.wts), 2 * (int64_t)n, bits);
    }
    if (P.rules & PILUC_DROP_INVERSE) {
        ILUPP_HIP(b_inv.alloc(sizeof(double) * 8 * (size_t)n));
        ILUPP_HIP(hipMemsetAsync(b_inv.p, 0, sizeof(double) * 8 * (size_t)n, st));
        a.inv = b_inv.as<double>();
    }
    a.perm = iarr(0); a.iperm = iarr(1); a.prow = iarr(2); a.iprow = iarr(3); a.numb = iarr(4); a.pnum = iarr(5); a.nonpiv = iarr(6);
    a.Uptr = iarr(7); a.Lptr = iarr(8); a.Sptr = iarr(9);
    a.Dinv = Dinv;
    a.ctrl = b_ctrl.as<int32_t>();
    a.dctrl = b_dctrl.as<double>();
    {
        // initialize_sparse_matrix_fields (ILUC.hpp:65-84): every row into the list of its first column, rows in ascending order each at the
        // head -- a sequential statement about n integers: on the host
        std::vector<int32_t> hp((size_t)n + 1), hc((size_t)n), lA((size_t)n, -1), hA((size_t)n, -1);
        PoolBlock b_fc;
        ILUPP_HIP(b_fc.alloc(sizeof(int32_t) * (size_t)n));
        hipLaunchKernelGGL(k_pc_first_cols, dim3((n + 255) / 256), dim3(256), 0, st, n, Arow.ptr, Arow.idx, b_fc.as<int32_t>());
        ILUPP_HIP(hipMemcpyAsync(hp.data(), Arow.ptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipMemcpyAsync(hc.data(), b_fc.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        for (int32_t k = 0; k < n; ++k)
            if (hp[(size_t)k] < hp[(size_t)k + 1]) { const int32_t c = hc[(size_t)k]; lA[(size_t)k] = hA[(size_t)c]; hA[(size_t)c] = k; }
        ILUPP_HIP(hipMemcpyAsync(a.perm, lA.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
        ILUPP_HIP(hipMemcpyAsync(a.iperm, hA.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
        ILUPP_HIP(hipMemcpyAsync(a.prow, Arow.ptr, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));      // firstA = pointer
        ILUPP_HIP(hipMemsetAsync(a.iprow, 0xff, sizeof(int32_t) * (size_t)n, st));          // listL = -1
        ILUPP_HIP(hipMemsetAsync(a.pnum, 0xff, sizeof(int32_t) * (size_t)n, st));           // listU = -1
        ILUPP_HIP(hipMemsetAsync(a.numb, 0, sizeof(int32_t) * (size_t)n, st));
        ILUPP_HIP(hipMemsetAsync(a.nonpiv, 0, sizeof(int32_t) * (size_t)n, st));
        hipLaunchKernelGGL(k_pc_ones, dim3((n + 255) / 256), dim3(256), 0, st, n, Dinv);
        iota_i32(st, b_id.as<int32_t>(), n + 1);
        int32_t c0[16] = {0};
        c0[1] = n - 1; c0[4] = 1;
        const double d0[2] = {tau, 0.0};
        ILUPP_HIP(hipMemcpyAsync(a.ctrl, c0, sizeof(c0), hipMemcpyHostToDevice, st));
        ILUPP_HIP(hipMemcpyAsync(a.dctrl, d0, sizeof(d0), hipMemcpyHostToDevice, st));
        ILUPP_HIP(hipMemsetAsync(a.Uptr, 0, sizeof(int32_t), st));
        ILUPP_HIP(hipMemsetAsync(a.Lptr, 0, sizeof(int32_t), st));
        ILUPP_HIP(hipMemsetAsync(a.Sptr, 0, sizeof(int32_t), st));
        ILUPP_HIP(hipStreamSynchronize(st));                   // (the host vectors above are done with)
    }
    int32_t ctrl[16] = {0};
    PoolBlock b_lvmem;
    bool in_mem = mem_ok && getenv("ILUPP_PILUC_CHAIN_MEM") != nullptr;              // (tests: the memory flavour from the first step)
    for (int launch = 0;; ++launch) {
        if (in_mem && !a.lvmem) { ILUPP_HIP(b_lvmem.alloc(kLvMemBytes)); a.lvmem = b_lvmem.as<char>(); }
        a.Uidx = SU.idx.as<int32_t>(); a.Uval = SU.val.as<double>(); a.capU = (int32_t)SU.cap;
        a.Lidx = SL.idx.as<int32_t>(); a.Lval = SL.val.as<double>(); a.capL = (int32_t)SL.cap;
        a.Sidx = SS.idx.as<int32_t>(); a.Sval = SS.val.as<double>(); a.capS = (int32_t)SS.cap;
        float ms = 0.f;
        { const int rc = chain_launch(st, a, in_mem ? CHAIN_PC_MEM : CHAIN_PC_LDS, &ms); if (rc) return rc; }      // (alone, or with the other chains of a batch)
        ILUPP_HIP(hipMemcpyAsync(ctrl, a.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        if (kernel_ms) *kernel_ms += ms;
        if (dbg) fprintf(stderr, "[ilupp] piluc (chain%s): n %d, launch %d (stores of %lld / %lld / %lld): status %d at step %d, %.2f ms\n", in_mem ? ", vectors in memory" : "",
                         n, launch, (long long)SU.cap, (long long)SL.cap, (long long)SS.cap, ctrl[0], ctrl[5], ms);
        if (ctrl[0] == 0) break;
        if (ctrl[0] == 4) {
            // a working vector (or the node list) of this step does not fit: the step is untouched.  The memory flavour takes the chain over at
            // this step (and keeps it: rows that long do not get shorter); beyond ITS capacity the level is not built
            if (in_mem || !mem_ok) return 1;
            in_mem = true;
            continue;
        }
        DpStore &S = ctrl[0] == 1 ? SU : ctrl[0] == 2 ? SL : SS;
        const int64_t used = ctrl[0] == 1 ? ctrl[8] : ctrl[0] == 2 ? ctrl[9] : ctrl[10];
        if (S.cap >= 0x7ffffff0ll || (launch > 40 && !getenv("ILUPP_DP_STORE"))) { set_error("partialILUC: the factors of a level outgrow 2^31 entries"); return ILUPP_ERR_MEMORY; }
        int64_t ncap = getenv("ILUPP_DP_STORE") ? S.cap + (int64_t)n + 2 + atol(getenv("ILUPP_DP_STORE")) : 2 * S.cap + (int64_t)n + 1024;
        if (ncap > 0x7ffffff0ll) ncap = 0x7ffffff0ll;
        { const int rc = S.grow(st, ncap, used); if (rc) return rc; }
    }
    {
        const int32_t last = ctrl[1], nA = ctrl[2];
        const bool to_the_end = ctrl[4] != 0;
        const int32_t *ident = b_id.as<int32_t>();
        { const int rc = seg_compress_sort(st, n, a.Lptr, a.Lidx, a.Lval, ident, 0, false, L); if (rc) return rc; }      // compress(), :2053-2054
        { const int rc = seg_compress_sort(st, n, a.Uptr, a.Uidx, a.Uval, ident, 0, true, U); if (rc) return rc; }
        if (to_the_end) {                                                           // :2055
            Anew->release();
            Anew->n = 0; Anew->nnz = 0; Anew->is_csr = true; Anew->owns = true;
            ILUPP_HIP(pool_malloc(&Anew->ptr, sizeof(int32_t)));
            ILUPP_HIP(hipMemsetAsync(Anew->ptr, 0, sizeof(int32_t), st));
            ILUPP_HIP(pool_malloc(&Anew->idx, sizeof(int32_t)));
            ILUPP_HIP(pool_malloc(&Anew->val, sizeof(double)));
            *kterm = n;
        } else {
            const int rc = seg_compress_sort(st, nA, a.Sptr, a.Sidx, a.Sval, ident, last + 1, true, Anew);       // :2057-2065
            if (rc) return rc;
            *kterm = last + 1;
        }
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    gd.keep = true;
    *Dinv_out = Dinv;
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
    DpStore SU, SL, SS;
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
    for (int q = 0; q < 7; ++q) a.wgt[q] = P.wgt[q];
    PoolBlock b_inv, b_wts;
    a.inv = nullptr; a.wts = nullptr;
    if (P.rules & (PILUC_DROP_WEIGHTED | PILUC_DROP_WEIGHTED2)) {                   // :426-428: weightsU, weightsL
        ILUPP_HIP(b_wts.alloc(sizeof(double) * 2 * (size_t)n));
        a.wts = b_wts.as<double>();
        { unsigned long long bits; memcpy(&bits, &P.init_weights_lu, sizeof(bits)); fill_u64(st, reinterpret_cast<unsigned long long *>(a.wts), 2 * (int64_t)n, bits); }
    }
    if (P.rules & PILUC_DROP_INVERSE) {                                             // :423-425: eight vectors of zeros
        ILUPP_HIP(b_inv.alloc(sizeof(double) * 8 * (size_t)n));
        ILUPP_HIP(hipMemsetAsync(b_inv.p, 0, sizeof(double) * 8 * (size_t)n, st));
        a.inv = b_inv.as<double>();
    }
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
#ifdef DP_PROF
    PoolBlock b_prof;
    ILUPP_HIP(b_prof.alloc(sizeof(long long) * 16));
    ILUPP_HIP(hipMemsetAsync(b_prof.p, 0, sizeof(long long) * 16, st));
    a.prof = b_prof.as<long long>();
#endif
    int32_t ctrl[16] = {0};
    bool in_lds = getenv("ILUPP_NO_DPLDS") == nullptr;         // the working vectors in LDS while they fit (status 4: from that step on in memory)
    for (int launch = 0;; ++launch) {
        a.Uidx = SU.idx.as<int32_t>(); a.linkU = SU.link.as<int32_t>(); a.rowU = SU.who.as<int32_t>(); a.Uval = SU.val.as<double>(); a.capU = (int32_t)SU.cap;
        a.Lidx = SL.idx.as<int32_t>(); a.linkL = SL.link.as<int32_t>(); a.colL = SL.who.as<int32_t>(); a.Lval = SL.val.as<double>(); a.capL = (int32_t)SL.cap;
        a.Sidx = SS.idx.as<int32_t>(); a.Sval = SS.val.as<double>(); a.capS = (int32_t)SS.cap;
        float ms = 0.f;
        { const int rc = chain_launch(st, a, in_lds ? CHAIN_DP_LDS : -1, &ms); if (rc) return rc; }
        ILUPP_HIP(hipMemcpyAsync(ctrl, a.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        if (kernel_ms) *kernel_ms += ms;
        if (dbg) fprintf(stderr, "[ilupp] pilucdp: n %d, launch %d (%s; stores of %lld / %lld / %lld): status %d at step %d, %.2f ms\n", n, launch,
                         in_lds ? "vectors in LDS" : "vectors in memory", (long long)SU.cap, (long long)SL.cap, (long long)SS.cap, ctrl[0], ctrl[5], ms);
        if (ctrl[0] == 0) break;
        if (ctrl[0] == 4) { in_lds = false; continue; }
        DpStore &S = ctrl[0] == 1 ? SU : ctrl[0] == 2 ? SL : SS;
        const int64_t used = ctrl[0] == 1 ? ctrl[8] : ctrl[0] == 2 ? ctrl[9] : ctrl[10];
        if (S.cap >= 0x7ffffff0ll || (launch > 40 && !getenv("ILUPP_DP_STORE"))) { set_error("ILU++ with pivoting: the factors of a level outgrow 2^31 entries"); return ILUPP_ERR_UNSUPPORTED; }
        int64_t ncap = getenv("ILUPP_DP_STORE") ? S.cap + (int64_t)n + 2 + atol(getenv("ILUPP_DP_STORE")) : 2 * S.cap + (int64_t)n + 1024;
        if (ncap > 0x7ffffff0ll) ncap = 0x7ffffff0ll;
        { const int rc = S.grow(st, ncap, used); if (rc) return rc; }
    }
#ifdef DP_PROF
    {
        long long hp[16];
        ILUPP_HIP(hipMemcpy(hp, a.prof, sizeof(hp), hipMemcpyDeviceToHost));
        static const char *names_mem[12] = {"clear + load row", "U rows subtracted", "pivot search", "scale z + swaps", "load column", "L columns subtracted", "scale w",
                                            "norms + take z", "write U", "take w + write L", "bucket moves", "group sort + level end"};
        static const char *names_lds[12] = {"start + load row", "U rows subtracted", "pivot search", "load column", "L columns subtracted", "commit + scale",
                                            "norms + take z", "write U", "take w + write L", "bucket moves", "group sort + level end", "stores complete"};
        const char **names = getenv("ILUPP_NO_DPLDS") ? names_mem : names_lds;
        long long tot = 0;
        for (int i = 0; i < 12; ++i) tot += hp[i];
        for (int i = 0; i < 12; ++i) fprintf(stderr, "[ilupp] pilucdp phase %-24s %10.3f Mticks  %5.1f %%\n", names[i], 1e-6 * (double)hp[i], 100.0 * (double)hp[i] / (double)(tot > 0 ? tot : 1));
    }
#endif
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
