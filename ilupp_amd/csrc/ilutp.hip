// ilupp_amd/csrc/ilutp.hip -- ILUTP: ILUT with column pivoting (gfx950).  SURVEY section 8 (f4).
//
// Reference: ILUTP2, ILUTP.hpp:13-140 (ILUTPPreconditioner, preconditioner_implementation.h:1050-1078; binding.cpp:313-326).  Row i of the
// factors comes from row i of the matrix: its entries are taken in the order of their POSITION under the column permutation of the
// moment (a std::map from position to entry that the reference walks while entries are added behind the current one), every entry
// left of position i is dropped if small against the norm of the row's original L part, else divided by the pivot of that row of U
// and that row subtracted; then the row is split at position i, each part thresholded against its own norm and cut to its budget by
// the reference's partial sort, and the largest kept entry of the U part becomes the pivot -- unless the diagonal entry beats the
// largest by piv_tol, in which case the diagonal is given the norm as its magnitude and wins (sparse_implementation.h:2036-2160).
// The pivot's column is swapped to position i.  Later rows see that permutation: the rows form a chain, and as for the other two
// pivoting factorisations (pilucdp.hip, ilucp.hip) ONE WAVE walks it, its lanes working inside the row:
//   * "the entry of the smallest position not yet visited" is a wave-wide minimum over the row's slots;
//   * subtracting a row of U: an entry per lane, new slots appended in entry order (ballot / prefix count = insertion order);
//   * norms are summed in slot order (64 values per pass, added one after the other), candidates collected in slot order, the cut by
//     the reference's own partial sort (select_largest) on one lane, the largest kept entry found as "first maximum in order".
// The working row lives in slot arrays (value, index, position, state) + an index -> slot map in HBM; a dropped entry leaves a dead
// zero slot behind as it does in the reference (its index may come back in a new slot).  The stores have the size the reference
// reserves (min(max_fill_in * n, (Integer) mem_factor * nnz), :37); running out of them is its error.
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "piluc_dev.h"
#include "dp_dev.h"

namespace ilupp {

struct TpArgs {
    int32_t n;
    const int32_t *Ap, *Ai; const double *Av;            // the matrix by its major slices (rows of the reference's A)
    double threshold, piv_tol;
    int32_t bp, max_fill, reserved, cap;
    int32_t *perm, *iperm, *occ;                          // occ: index -> slot of the working row, -1
    double *sval; int32_t *sidx, *skey, *sstate;          // slots: value, index, position at insertion, state (0 waiting, 1 visited, 2 dropped)
    int32_t *Uptr, *Uidx; double *Uval;
    int32_t *Lptr, *Lidx; double *Lval;
    double *keyL, *keyU; int32_t *listL, *listU;
    int32_t *ctrl;                                        // [0] status (0 done, 3 memory, 1 zero pivot, 12 slots exhausted), [1] zero pivots, [5] the row
};

// first position of the largest key in [lo, hi) (strictly greater wins: the first of equals), by the whole wave
__device__ int tp_first_max(const double *key, int lo, int hi, int lane)
{
    double mx = -1.0;
    int pos = 0x7fffffff;
    for (int i = lo + lane; i < hi; i += 64) { const double a = key[i]; if (pos == 0x7fffffff || a > mx) { if (pos == 0x7fffffff || a > mx) { mx = a; pos = i; } } }
    pos = wv_argmax_first(mx, pos);
    return pos == 0x7fffffff ? lo : pos;
}

// the two selections of ILUTP2 (sparse_implementation.h:1943-2033 / :2036-2160) over the slots 0 .. ns; lists of INDICES in listL / listU
__device__ void tp_select(const TpArgs &A, int ns, int n_L, int n_U, double tau_L, double tau_U, int mid, bool with_piv, double piv_tol, int &nL, int &nU,
                          int lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    // norms of the two parts in slot order, the largest magnitude of the U part, the diagonal entry
    double accL = 0.0, accU = 0.0, larg = 0.0, potpiv = 0.0;
    int pos_pot = -1;
    for (int base = 0; base < ns; base += 64) {
        const int s = base + lane;
        const bool act = s < ns;
        const double x = act ? A.sval[s] : 0.0;
        const int key = act ? A.iperm[A.sidx[s]] : 0;
        const int cls = !act ? 0 : (key < mid ? 1 : 2);
        const double t = x * x;
        if (cls == 2) { const double a = fabs(x); if (a > larg) larg = a; }
        const unsigned long long dm = __ballot(cls == 2 && key == mid);
        if (dm) { const int src = 63 - __builtin_clzll(dm); potpiv = fabs(wv_f64(x, src)); pos_pot = base + src; }
        const int cnt = ns - base < 64 ? ns - base : 64;
        for (int i = 0; i < cnt; ++i) {
            const double ti = wv_f64(t, i);
            const int ci = wv_i32(cls, i);
            if (ci == 1) accL = accL + ti; else if (ci == 2) accU = accU + ti;
        }
    }
    larg = wv_max_f64(larg);
    const double nrmL = sqrt(accL), nrmU = sqrt(accU);
    int keep_diag = -1;
    if (with_piv && !((larg * piv_tol >= potpiv) || (pos_pot < 0))) keep_diag = pos_pot;
    int cL = 0, cU = 0;
    for (int base = 0; base < ns; base += 64) {
        const int s = base + lane;
        const bool act = s < ns;
        const int idx = act ? A.sidx[s] : 0;
        const int key = act ? A.iperm[idx] : 0;
        const double a = !act ? 0.0 : (s == keep_diag ? nrmU : fabs(A.sval[s]));
        const bool isL = act && key < mid && a > nrmL * tau_L;
        const bool isU = act && key >= mid && a > nrmU * tau_U;
        const unsigned long long mL = __ballot(isL), mU = __ballot(isU);
        if (isL) { const int p = cL + __popcll(mL & lt); A.listL[p] = idx; A.keyL[p] = a; }
        if (isU) { const int p = cU + __popcll(mU & lt); A.listU[p] = idx; A.keyU[p] = a; }
        cL += __popcll(mL); cU += __popcll(mU);
    }
    DP_SYNC();
    int offL = 0, offU = 0;
    if (cL > n_L) {
        if (lane == 0 && n_L > 0) select_largest(A.keyL, A.listL, 0, cL - 1, n_L);
        offL = cL - n_L;
    }
    if (cU > n_U) {
        if (lane == 0 && n_U > 0) select_largest(A.keyU, A.listU, 0, cU - 1, n_U);
        offU = cU - n_U;
    }
    DP_SYNC();
    if (cU > 0) {
        const int pos = tp_first_max(A.keyU, offU, cU, lane);
        if (lane == 0) { const int t = A.listU[pos]; A.listU[pos] = A.listU[cU - 1]; A.listU[cU - 1] = t; }
    }
    DP_SYNC();
    // the kept parts to the front
    nL = cL - offL; nU = cU - offU;
    if (offL > 0) {
        for (int base = 0; base < nL; base += 64) { const int i = base + lane; const int v = i < nL ? A.listL[offL + i] : 0; DP_SYNC(); if (i < nL) A.listL[i] = v; }
    }
    if (offU > 0) {
        for (int base = 0; base < nU; base += 64) { const int i = base + lane; const int v = i < nU ? A.listU[offU + i] : 0; DP_SYNC(); if (i < nU) A.listU[i] = v; }
    }
    DP_SYNC();
}

__global__ void __launch_bounds__(64) k_ilutp(TpArgs A)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n = A.n;
    int zero_piv = 0, pU = 0, pL = 0;
    double piv_tol = A.piv_tol;
#define TP_FAIL(code) do { if (lane == 0) { A.ctrl[0] = (code); A.ctrl[1] = zero_piv; A.ctrl[5] = i; } return; } while (0)

    for (int i = 0; i < n; ++i) {
        if (i == A.bp) piv_tol = 1.0;                                               // :43-44
        int ns = 0;
        // ---- row i of the matrix with the positions of its columns (:46-52) ----
        const int r0 = A.Ap[i], r1 = A.Ap[i + 1];
        if (r1 - r0 > A.cap) TP_FAIL(12);
        double acc = 0.0;
        for (int base = r0; base < r1; base += 64) {
            const int e = base + lane;
            const bool act = e < r1;
            const int c = act ? A.Ai[e] : 0;
            const double v = act ? A.Av[e] : 0.0;
            const int key = act ? A.iperm[c] : 0;
            const int pc = (act && e > r0) ? A.Ai[e - 1] : -1;
            const bool first = act && c != pc;                                      // (a column stored twice: one slot, the last value)
            const unsigned long long mask = __ballot(first);
            const int s = ns + __popcll(mask & lt) - (first ? 0 : 1);
            if (first) { A.sidx[s] = c; A.skey[s] = key; A.sstate[s] = 0; A.occ[c] = s; }
            if (act && (e + 1 >= r1 || A.Ai[e + 1] != c)) A.sval[s] = v;
            ns += __popcll(mask);
            const double t = v * v;
            const int cnt = r1 - base < 64 ? r1 - base : 64;
            for (int q = 0; q < cnt; ++q) { const double tq = wv_f64(t, q); const int kq = wv_i32(key, q); if (kq < i) acc = acc + tq; }
        }
        const double norm_wL = sqrt(acc);
        DP_SYNC();
        // ---- the entries left of position i, by position (:54-67) ----
        for (;;) {
            int kmin = 0x7fffffff, smin = -1;
            for (int s = lane; s < ns; s += 64) { const int key = A.skey[s]; if (A.sstate[s] == 0 && key < i && key < kmin) { kmin = key; smin = s; } }
            {   // (the positions of the waiting slots are all different: the lane that holds the smallest names its slot)
                const int km = wv_min_i32(kmin);
                const unsigned long long who = __ballot(kmin == km && smin >= 0);
                smin = who ? wv_i32(smin, __builtin_ctzll(who)) : -1;
                kmin = km;
            }
            if (smin < 0) break;
            const double cur = A.sval[smin];
            if (fabs(cur) < A.threshold * norm_wL) {                                // current_zero_set (sparse_implementation.h:2376-2384)
                if (lane == 0) { A.sval[smin] = 0.0; A.occ[A.sidx[smin]] = -1; A.sstate[smin] = 2; }
                DP_SYNC();
                continue;
            }
            const int u0 = A.Uptr[kmin], u1 = A.Uptr[kmin + 1];
            const double wk = cur / A.Uval[u0];
            if (lane == 0) { A.sval[smin] = wk; A.sstate[smin] = 1; }
            for (int base = u0 + 1; base < u1; base += 64) {
                const int e = base + lane;
                const bool act = e < u1;
                const int c = act ? A.Uidx[e] : 0;
                int sl = act ? A.occ[c] : 0;
                const bool isnew = act && sl < 0;
                const unsigned long long mask = __ballot(isnew);
                if (ns + __popcll(mask) > A.cap) TP_FAIL(12);
                double curv = 0.0;
                if (isnew) { sl = ns + __popcll(mask & lt); A.sidx[sl] = c; A.skey[sl] = A.iperm[c]; A.sstate[sl] = 0; A.occ[c] = sl; }
                else if (act) curv = A.sval[sl];
                if (act) { const double prod = wk * A.Uval[e]; A.sval[sl] = curv - prod; }
                ns += __popcll(mask);
            }
            DP_SYNC();
        }
        // ---- split, threshold, cut, pivot (:69-84) ----
        int nL = 0, nU = 0;
        tp_select(A, ns, A.max_fill - 1, A.max_fill, A.threshold, A.threshold, i, true, piv_tol, nL, nU, lane);
        if (nU == 0) {
            if (A.threshold > 0.0) tp_select(A, ns, A.max_fill - 1, A.max_fill, A.threshold, 0.0, i, false, 0.0, nL, nU, lane);
            if (nU == 0) {
                ++zero_piv;
                const int c = A.perm[i];
                if (A.occ[c] < 0) {
                    if (ns + 1 > A.cap) TP_FAIL(12);
                    if (lane == 0) { A.sidx[ns] = c; A.skey[ns] = i; A.sstate[ns] = 0; A.occ[c] = ns; A.sval[ns] = 1.0; }
                    ++ns;
                } else if (lane == 0) A.sval[A.occ[c]] = 1.0;
                if (lane == 0) A.listU[0] = c;
                nU = 1;
                DP_SYNC();
            }
        }
        // ---- the rows of L (its 1 last, positions as column indices) and of U (pivot first, original column indices), :86-108 ----
        if ((long)pL + nL + 1 > (long)A.reserved) TP_FAIL(3);
        for (int j = lane; j < nL; j += 64) { const int c = A.listL[nL - 1 - j]; A.Lval[pL + j] = A.sval[A.occ[c]]; A.Lidx[pL + j] = A.iperm[c]; }
        if (lane == 0) { A.Lval[pL + nL] = 1.0; A.Lidx[pL + nL] = i; A.Lptr[i + 1] = pL + nL + 1; }
        if ((long)pU + nU > (long)A.reserved) TP_FAIL(3);
        for (int j = lane; j < nU; j += 64) { const int c = A.listU[nU - 1 - j]; A.Uval[pU + j] = A.sval[A.occ[c]]; A.Uidx[pU + j] = c; }
        DP_SYNC();
        const int c0 = A.Uidx[pU];
        const double piv = A.Uval[pU];
        if (lane == 0) {
            A.Uptr[i + 1] = pU + nU;
            const int pi = A.perm[i], p = A.iperm[c0];
            const int t = A.iperm[pi]; A.iperm[pi] = A.iperm[c0]; A.iperm[c0] = t;
            const int u = A.perm[i]; A.perm[i] = A.perm[p]; A.perm[p] = u;
        }
        if (piv == 0) TP_FAIL(1);                                                   // "encountered zero pivot in row", :117-118
        for (int s = lane; s < ns; s += 64) A.occ[A.sidx[s]] = -1;                  // zero_reset
        pL += nL + 1; pU += nU;
        DP_SYNC();
    }
    if (lane == 0) { A.ctrl[0] = 0; A.ctrl[1] = zero_piv; A.ctrl[5] = n; }
#undef TP_FAIL
}

__global__ void k_tp_init(int32_t n, int32_t *perm, int32_t *iperm, int32_t *occ)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { perm[i] = iperm[i] = i; occ[i] = -1; }
}
__global__ void k_tp_gather_i32(int64_t nnz, const int32_t *__restrict__ idx, const int32_t *__restrict__ map, int32_t *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nnz) out[j] = map[idx[j]];
}

// A: the matrix by rows; L by rows (1 last, permuted numbering, sorted), Up: U by rows in the PERMUTED numbering (pivot first, sorted),
// Uorig: the same rows with the original column indices (what the reference stores), perm (device, n)
int ilutp_factor(hipStream_t st, const DevMat &A, int32_t max_fill_in, double threshold, double piv_tol, int32_t bp, double mem_factor,
                 DevMat *L, DevMat *Up, DevMat *Uorig, int32_t *perm_out, int32_t *zero_pivots, float *kernel_ms)
{
    const int32_t n = A.n;
    const int64_t nnz = A.nnz;
    if (max_fill_in < 1) max_fill_in = 1;
    if (max_fill_in > n) max_fill_in = n;
    int64_t reserved;
    {
        const int64_t a = (int64_t)max_fill_in * (int64_t)n, b = (int64_t)((int32_t)mem_factor) * nnz;      // (Integer) mem_factor * A.non_zeroes(), :37
        reserved = a < b ? a : b;
        if (reserved < 0) reserved = 0;
        if (reserved > 0x7ffffff0ll) { set_error("ILUTP: the memory to reserve exceeds 2^31 entries"); return ILUPP_ERR_UNSUPPORTED; }
    }
    const size_t slot = ((size_t)n + 64) & ~(size_t)15;
    const size_t capw = 4 * slot;                                                   // slots of the working row (dead ones included)
    PoolBlock b_i, b_d, b_ctrl, b_ui, b_uv, b_li, b_lv;
    ILUPP_HIP(b_i.alloc(sizeof(int32_t) * (slot * 5 + capw * 5)));
    ILUPP_HIP(b_d.alloc(sizeof(double) * (capw * 3)));
    ILUPP_HIP(b_ctrl.alloc(64));
    const size_t cap = (size_t)reserved + 1;
    ILUPP_HIP(b_ui.alloc(sizeof(int32_t) * cap)); ILUPP_HIP(b_uv.alloc(sizeof(double) * cap));
    ILUPP_HIP(b_li.alloc(sizeof(int32_t) * cap)); ILUPP_HIP(b_lv.alloc(sizeof(double) * cap));
    int32_t *I = b_i.as<int32_t>();
    TpArgs a;
    a.n = n; a.Ap = A.ptr; a.Ai = A.idx; a.Av = A.val;
    a.threshold = threshold; a.piv_tol = piv_tol; a.bp = bp; a.max_fill = max_fill_in; a.reserved = (int32_t)reserved; a.cap = (int32_t)capw;
    a.perm = I; a.iperm = I + slot; a.occ = I + 2 * slot; a.Uptr = I + 3 * slot; a.Lptr = I + 4 * slot;
    int32_t *W = I + 5 * slot;
    a.sidx = W; a.skey = W + capw; a.sstate = W + 2 * capw; a.listL = W + 3 * capw; a.listU = W + 4 * capw;
    a.sval = b_d.as<double>(); a.keyL = a.sval + capw; a.keyU = a.keyL + capw;
    a.Uidx = b_ui.as<int32_t>(); a.Uval = b_uv.as<double>(); a.Lidx = b_li.as<int32_t>(); a.Lval = b_lv.as<double>();
    a.ctrl = b_ctrl.as<int32_t>();
    ILUPP_HIP(hipMemsetAsync(a.ctrl, 0, 64, st));
    ILUPP_HIP(hipMemsetAsync(a.Uptr, 0, sizeof(int32_t), st));
    ILUPP_HIP(hipMemsetAsync(a.Lptr, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_tp_init, dim3((n + 255) / 256), dim3(256), 0, st, n, a.perm, a.iperm, a.occ);
    EventPair ev;
    ILUPP_HIP(ev.create());
    ILUPP_HIP(hipEventRecord(ev.a, st));
    hipLaunchKernelGGL(k_ilutp, dim3(1), dim3(64), 0, st, a);
    ILUPP_HIP(hipEventRecord(ev.b, st));
    int32_t ctrl[8] = {0};
    ILUPP_HIP(hipMemcpyAsync(ctrl, a.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    ILUPP_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (kernel_ms) *kernel_ms = ms;
    if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] ilutp: n %d, stores of %lld: status %d at row %d, %.2f ms\n", n, (long long)reserved, ctrl[0], ctrl[5], ms);
    if (ctrl[0] == 3) { set_error("ILUTP2: memory reserved was insufficient."); return ILUPP_ERR_MEMORY; }
    if (ctrl[0] == 1) { set_error("matrix_sparse::ILUTP2: encountered zero pivot in row " + std::to_string(ctrl[5])); return ILUPP_ERR_ZERO_PIVOT; }
    if (ctrl[0] != 0) { set_error("ILUTP: the working row of row " + std::to_string(ctrl[5]) + " outgrows its slots"); return ILUPP_ERR_INTERNAL; }
    if (zero_pivots) *zero_pivots = ctrl[1];
    // compress() (:131-132); L.normal_order(); U.reorder(inverse_perm) = every row by the permuted position of its columns (:134-135)
    PoolBlock b_id;
    ILUPP_HIP(b_id.alloc(sizeof(int32_t) * (size_t)n));
    iota_i32(st, b_id.as<int32_t>(), n);
    { const int rc = seg_compress_sort(st, n, a.Lptr, a.Lidx, a.Lval, b_id.as<int32_t>(), 0, true, L); if (rc) return rc; }
    { const int rc = seg_compress_sort(st, n, a.Uptr, a.Uidx, a.Uval, a.iperm, 0, true, Up); if (rc) return rc; }
    Uorig->release();
    Uorig->n = n; Uorig->nnz = Up->nnz; Uorig->is_csr = true; Uorig->owns = true;
    ILUPP_HIP(pool_malloc(&Uorig->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(pool_malloc(&Uorig->idx, sizeof(int32_t) * (size_t)(Up->nnz > 0 ? Up->nnz : 1)));
    ILUPP_HIP(pool_malloc(&Uorig->val, sizeof(double) * (size_t)(Up->nnz > 0 ? Up->nnz : 1)));
    ILUPP_HIP(hipMemcpyAsync(Uorig->ptr, Up->ptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, st));
    if (Up->nnz > 0) {
        ILUPP_HIP(hipMemcpyAsync(Uorig->val, Up->val, sizeof(double) * (size_t)Up->nnz, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_tp_gather_i32, dim3((unsigned)((Up->nnz + 255) / 256)), dim3(256), 0, st, Up->nnz, Up->idx, a.perm, Uorig->idx);
    }
    ILUPP_HIP(hipMemcpyAsync(perm_out, a.perm, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    return ILUPP_OK;
}

}  // namespace ilupp
