// ilupp_amd/csrc/piluc_df.hip -- one level of the multilevel ILU++ preconditioner without pivoting: matrix_sparse::partialILUC
// (reference ILUCDP.hpp:1405-2231 with the list helpers ILUC.hpp:31-101) for the precon_parameter 10 family
// (parameters_implementation.h:927-934: no row or column permutation inside the factorisation, error-propagation dropping, the
// level ends at the first pivot smaller than MIN_PIVOT), as the DATAFLOW computation of iluc_df.hip: one wave per step k (row k
// of U and column k of L together), a step starts when everything that reaches it has finished, results bit-identical to the
// reference's sequential loop.
//
// What differs from ILUC2 (iluc_df.hip), all of it arithmetic that has to be reproduced exactly:
//   * A ~ L D U with UNIT L and U: the contributors are subtracted as (l(k,h) / Dinv[h]) * u(h,j) (:1591-1593) and
//     (u(h,k) / Dinv[h]) * l(i,h) (:1668-1670); the whole working row and column are multiplied by Dinv[k] = 1 / pivot before
//     anything is dropped (:1639-1641, :1676); a missing diagonal is a pivot 0 (operator[] inserts the slot);
//   * dropping by error propagation (:1716-1726, :1896-1906 with sparse_implementation.h:1360-1415, WEIGHTED_DROPPING): an entry of
//     row k of U stays if  |column k of L|_1 * |u| >= tau,  an entry of column k of L if  |row k of U|_1 * |l| >= tau  -- both
//     1-norms over ALL slots of the scaled working vectors in insertion order (the column's includes the slot of index k that the
//     contributors' tails bring, which is why the tails here start AT the diagonal index); no limit on the number of entries
//     (MAX_FILLIN_IS_INF), hence no fixed slabs: rows and columns are appended to two stores through atomic cursors;
//   * the level ends at the first k > MIN_ELIM_FACTOR n with |pivot| < MIN_PIVOT (:1619-1636): rows k.. become the Schur
//     complement.  "First" is a property of the sequential order; here a step that meets a small pivot PARKS (records its index,
//     finishes nothing, releases nothing), every step above the smallest parked index is dropped when it is drawn, and the
//     kernel ends when no step is queued or running.  All steps below the smallest parked index have then finished -- nothing
//     they wait for can lie above them -- and what steps above it did before it was known is ignored.
//   * the Schur complement (:1589-1602 with eliminate == false, :1716-1718, :1822-1850): row k >= kterm is A's part right of
//     kterm minus the contributions of the columns h < kterm with a stored l(k,h), in the order of the reference's re-threaded
//     list -- which keeps being re-threaded through the Schur rows (update_triangular_fields runs for them too), so the rows
//     hand the chains on exactly as steps do.  A second launch of the same kernel in Schur mode does that: contributors from
//     the touch records of the steps below kterm, tails = the part of their U rows right of kterm.
#include <stdlib.h>

#include <chrono>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "iluc_common.h"
#include "piluc_dev.h"

#ifndef PILUC_WSGB
#define PILUC_WSGB 16
#endif

namespace ilupp {

// ctrl: [2] error (11..17 = a capacity, 2 = timeout), [3] smallest parked step, [4] finished steps (every wave adds its count when it leaves),
//       [6] cursor of the U store (Schur mode: of the store of the Schur rows), [7] cursor of the L store -- both moved a CHUNK at a time, a wave
//       appends its rows to its own chunk --, [9] progress mark, then the queues: queue q has its head at kCuQBase + 64 q, its tail 32 words later
//       and, 48 words later, a 64-bit word {pushes so far : steps queued or running}.
// One address that every step updates is a bottleneck of its own (a million steps, 100+ ms, whatever the number of waves): hence the chunks, the
// per-queue counts and the sparse progress mark.
__host__ __device__ inline size_t piluc_ws_bytes(int ne, int ns, int tm)
{
    const size_t dbl = (size_t)ne + 2 * (size_t)ns + 2 * (size_t)tm;
    const size_t ints = 2 * (size_t)ne + 6 * (size_t)ns + 2 + 4 * (size_t)ns + 11 * (size_t)tm + 1 + 4 * (size_t)ns + 2 + tm;
    return ((dbl * 8 + ints * 4) + 255) & ~(size_t)255;
}

struct PilucArgs {
    unsigned char *gws;                       // global class: the waves' working arrays
    int32_t gNE, gNS, gTM;
    int32_t n;
    const int32_t *ptr, *idx;                 // the level's matrix, ROW storage
    const double *val;
    const int32_t *colptr, *colord;           // its sub-diagonal part by columns, CSR positions in the reference's traversal order
    const int32_t *rowof;
    double tau;                               // dropping threshold of this launch
    double min_pivot;
    int32_t park_from;                        // steps k > park_from may end the level (0x7fffffff: none may)
    int32_t rules, combine, scale_invdiag;    // the dropping rules that make up a row's weight (PILUC_DROP_*), COMBINE_FACTOR, SCALE_WEIGHT_INVDIAG
    double wgt[5];                            // WEIGHT_STANDARD_DROP, _DROP2, WEIGHT_ERR_PROP_DROP, _DROP2, WEIGHT_PIVOT_DROP
    double neutral, min_weight;               // NEUTRAL_ELEMENT, MIN_WEIGHT
    int32_t budget;                           // entries a row of U / a column of L may keep besides the 1 (max_fill_in - 1); Schur mode: entries of a row (max_fill_in)
    int32_t schur, kterm;                     // Schur mode: rows kterm.. of the Schur complement
    int32_t T, nq;
    int32_t *Uidx, *Lidx;                     // the stores: a row of U = (k, 1) then the kept entries by increasing index; same for L
    double *Uval, *Lval;
    int32_t capU, capL;
    int32_t chunk;                            // entries a wave takes from a store at a time
    int32_t *Ustart, *Ulen, *Lstart, *Llen;   // per step (Schur mode: Ustart / Ulen of the steps below kterm are read)
    double *Dinv;
    int32_t *Sidx;                            // Schur mode: store of the Schur rows (column indices as in the level's matrix)
    double *Sval;
    int32_t capS;
    int32_t *Sstart, *Slen;                   // per Schur row k - kterm
    int32_t *cntL, *cntU;                     // touch records of a step: stored l(k, .) / stored u(., k)
    unsigned long long *recL, *recU;          // [n][T][4]
    int32_t *pending, *rq, *ctrl;
};

// the weight of a row of U / a column of L (ILUCDP.hpp:1719-1728, :1896-1906; combine(): parameters_implementation.h:526-534 with std::max):
// n2own = 2-norm of the vector that is dropped, n1other = 1-norm of the other factor's vector, dinv = Dinv[k] of that moment
__device__ __forceinline__ double piluc_weight(const PilucArgs &A, double n2own, double n1other, double dinv)
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

// first position p in [lo, hi] (1-based list `slots`, keys key[slots[p]]) whose key is >= v; hi + 1 if none
__device__ __forceinline__ int lower_pos(const int *slots, const int *key, int lo, int hi, int v)
{
    int a = lo, b = hi + 1;
    while (a < b) { const int mid = (a + b) >> 1; if (key[slots[mid]] < v) a = mid + 1; else b = mid; }
    return a;
}

template <int kLNE, int kLNS, int kLTM, bool kGlobal>
__global__ void __launch_bounds__(64)
k_piluc_df(PilucArgs A)
{
    __shared__ int s_erow[kLNE], s_eslot[kLNE];
    __shared__ double s_eval[kLNE];
    __shared__ int s_srow[kLNS], s_scnt[kLNS], s_srank[kLNS], s_sridx[kLNS + 1], s_crank[kLNS], s_keptslot[kLNS + 1];
    __shared__ double s_sval[kLNS];
    __shared__ int s_hslot[4 * kLNS];
    __shared__ int s_tx[kLTM], s_tt[kLTM], s_trem[kLTM], s_tnxt[kLTM], s_tseq[kLTM], s_tw[kLTM];
    __shared__ double s_tv[kLTM];
    __shared__ int s_cx[kLTM], s_crem[kLTM], s_cbase[kLTM + 1], s_cnxt[kLTM], s_cw[kLTM];
    __shared__ double s_cv[kLTM];
    // the row of U waits (scaled, not dropped yet) until the column of L is known: its slots, how often each was announced, then its
    // kept slots by increasing index and where their touch records went
    __shared__ int s_zcol[kLNS], s_zcnt[kLNS], s_zkept[kLNS + 1], s_zridx[kLNS + 1];
    __shared__ double s_zval[kLNS];
    const int kNE = kGlobal ? A.gNE : kLNE, kNS = kGlobal ? A.gNS : kLNS, kTM = kGlobal ? A.gTM : kLTM;
    int *erow = s_erow, *eslot = s_eslot, *srow = s_srow, *scnt = s_scnt, *srank = s_srank, *sridx = s_sridx, *crank = s_crank,
        *keptslot = s_keptslot, *hslot = s_hslot, *tx = s_tx, *tt = s_tt, *trem = s_trem, *tnxt = s_tnxt, *tseq = s_tseq, *tw = s_tw,
        *cx = s_cx, *crem = s_crem, *cbase = s_cbase, *cnxt = s_cnxt, *cw = s_cw, *zcol = s_zcol, *zcnt = s_zcnt, *zkept = s_zkept,
        *zridx = s_zridx;
    double *eval = s_eval, *sval = s_sval, *tv = s_tv, *cv = s_cv, *zval = s_zval;
    if (kGlobal) {
        double *d = reinterpret_cast<double *>(A.gws + (size_t)blockIdx.x * piluc_ws_bytes(A.gNE, A.gNS, A.gTM));
        eval = d; d += kNE; sval = d; d += kNS; zval = d; d += kNS; tv = d; d += kTM; cv = d; d += kTM;
        int *q = reinterpret_cast<int *>(d);
        erow = q; q += kNE; eslot = q; q += kNE;
        srow = q; q += kNS; scnt = q; q += kNS; srank = q; q += kNS; sridx = q; q += kNS + 1; crank = q; q += kNS; keptslot = q; q += kNS + 1;
        hslot = q; q += 4 * kNS;
        tx = q; q += kTM; tt = q; q += kTM; trem = q; q += kTM; tnxt = q; q += kTM; tseq = q; q += kTM; tw = q; q += kTM;
        cx = q; q += kTM; crem = q; q += kTM; cbase = q; q += kTM + 1; cnxt = q; q += kTM; cw = q; q += kTM;
        zcol = q; q += kNS; zcnt = q; q += kNS; zkept = q; q += kNS + 1; zridx = q; q += kNS + 1;
    }

    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int m = A.n, T = A.T, nq = A.nq;
    const bool schur = A.schur != 0;
    const int kterm = A.kterm;
#define PU_FAIL(code) do { if (lane == 0) atomicCAS(&A.ctrl[2], 0, (code)); return; } while (0)
    // (one wave per block: __syncthreads() orders LDS; with the arrays in global memory the wave's stores must have landed too)
#define PU_SYNC() do { if (kGlobal) __builtin_amdgcn_s_waitcnt(0); __syncthreads(); } while (0)
    // a step that is done (finished, parked or dropped): one less queued or running -- after everything it pushed
#define PU_OUT(q_) reinterpret_cast<unsigned long long *>(&A.ctrl[kCuQBase + 64 * (q_) + 48])
#define PU_RETIRE() do { __builtin_amdgcn_s_waitcnt(0); if (lane == 0) atomicAdd(PU_OUT(k % nq), ~0ull); } while (0)
#define PU_PUSH(x_) do { const int xq_ = (x_) % nq; atomicAdd(PU_OUT(xq_), (1ull << 32) + 1ull); const int pos_ = atomicAdd(&A.ctrl[kCuQBase + 64 * xq_ + 32], 1); \
                         st_agent_i32(&A.rq[(size_t)xq_ * qcap + pos_], (x_)); } while (0)

    int ndone = 0;                           // steps this wave finished
    int uoff = 0, uend = 0, loff = 0, lend = 0;      // the wave's chunks of the two stores
    const int myq = (int)(blockIdx.x % nq);
    const int qcap = (m + nq - 1) / nq;
    const int qtotal = (m - myq + nq - 1) / nq;
    int32_t *const qhead = &A.ctrl[kCuQBase + 64 * myq];
    const int32_t *const myrq = A.rq + (size_t)myq * qcap;
    for (;;) {
        int tkt = 0;
        if (lane == 0) tkt = atomicAdd(qhead, 1);
        tkt = __builtin_amdgcn_readfirstlane(tkt);
        if (tkt >= qtotal) break;
        int k;
        unsigned spins = 0;
        int seen = -1;
        for (;;) {
            k = ld_agent_i32(&myrq[tkt]);
            if (k >= 0) break;
            if ((++spins & 63u) == 0) {
                if (ld_agent_i32(&A.ctrl[2]) != 0) return;
                // nothing queued or running anywhere: whoever could have filled this slot is gone (a step retires after its pushes).  The
                // counts are per queue; two looks with no push in between (the push counters) make the sum a consistent one
                {
                    unsigned long long first = 1ull, again = 1ull;
                    for (int look = 0; look < 2; ++look) {
                        const unsigned long long w = lane < nq ? __hip_atomic_load(PU_OUT(lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                        unsigned long long busy = __ballot((unsigned)w != 0u) != 0ull ? 1ull : 0ull;
                        unsigned long long pushes = w >> 32;
                        for (int off = 32; off > 0; off >>= 1) pushes += __shfl_xor(pushes, off);
                        const unsigned long long v = (pushes << 1) | busy;
                        if (look == 0) first = v; else again = v;
                        if (busy) break;
                    }
                    if ((first & 1ull) == 0ull && first == again) {
                        k = ld_agent_i32(&myrq[tkt]);
                        if (k >= 0) break;
                        if (lane == 0 && ndone) atomicAdd(&A.ctrl[4], ndone);
                        return;
                    }
                }
                const int done = ld_agent_i32(&A.ctrl[9]);
                if (done != seen) { seen = done; spins = 0; }
                if (spins > kCuSpinLimit) PU_FAIL(2);
            }
            __builtin_amdgcn_s_sleep(4);
        }
        k = __builtin_amdgcn_readfirstlane(k);
        // above a parked step: the level has ended before this one
        if (!schur && k > __builtin_amdgcn_readfirstlane(ld_agent_i32(&A.ctrl[3]))) { PU_RETIRE(); continue; }

        int nzs = 0;                    // z: slots
        int dslot = -1;
        double dinv = 1.0, dinv_store = 1.0;
        bool parked = false;
        // ================================ the two halves: z (row k of U), then w (column k of L) ================================
        for (int half = 0; half < (schur ? 1 : 2); ++half) {
            const bool Z = half == 0;
            // ---- A's part, in the reference's insertion order ----
            int na = 0;
            if (Z) {
                const int lo = schur ? kterm : k;
                const int r0 = __builtin_amdgcn_readfirstlane(A.ptr[k]), r1 = __builtin_amdgcn_readfirstlane(A.ptr[k + 1]);
                int first = r1;                                        // first entry with column >= lo (firstA[k], ILUC.hpp:86-101)
                for (int base = r0; base < r1; base += 64) {
                    const int q = base + lane;
                    const unsigned long long mk = __ballot(q < r1 && A.idx[q] >= lo);
                    if (mk) { first = base + __ffsll((long long)mk) - 1; break; }
                }
                first = __builtin_amdgcn_readfirstlane(first);
                na = r1 - first;
                if (na > kNS) PU_FAIL(11);
                for (int e = lane; e < na; e += 64) { erow[e] = A.idx[first + e]; eval[e] = A.val[first + e]; eslot[e] = e; srow[e] = erow[e]; }
            } else {
                const int b = __builtin_amdgcn_readfirstlane(A.colptr[k]);
                na = __builtin_amdgcn_readfirstlane(A.colptr[k + 1]) - b;
                if (na > kNS) PU_FAIL(11);
                for (int e = lane; e < na; e += 64) { const int q = A.colord[b + e]; erow[e] = A.rowof[q]; eval[e] = A.val[q]; eslot[e] = e; srow[e] = erow[e]; }
            }
            // ---- the touch records of this kind (Schur mode: only those written by steps below kterm) ----
            const int ntraw = __builtin_amdgcn_readfirstlane(ld_agent_i32(Z ? &A.cntL[k] : &A.cntU[k]));
            if (ntraw > T) PU_FAIL(12);
            const unsigned long long *recs = (Z ? A.recL : A.recU) + (size_t)k * T * 4;
            int nt = 0;
            for (int base = 0; base < ntraw; base += 64) {
                const int q = base + lane;
                Rec32 rr;
                rr.w[0] = rr.w[1] = rr.w[2] = rr.w[3] = 0ull;
                if (q < ntraw) rr = ld_agent_rec32(recs + (size_t)q * 4);
                const int writer = (int)(unsigned)rr.w[0];
                const bool take = q < ntraw && (!schur || writer < kterm);
                const unsigned long long mk = __ballot(take);
                const int dst = nt + __popcll(mk & lt_mask);
                if (take && dst < kTM) {
                    tw[dst] = writer; tx[dst] = (int)(unsigned)(rr.w[0] >> 32);
                    tt[dst] = (int)(unsigned)rr.w[1]; trem[dst] = (int)(unsigned)(rr.w[1] >> 32);
                    tv[dst] = __longlong_as_double((long long)rr.w[2]);
                    tnxt[dst] = (int)(unsigned)rr.w[3]; tseq[dst] = (int)(unsigned)(rr.w[3] >> 32);
                }
                nt += __popcll(mk);
            }
            nt = __builtin_amdgcn_readfirstlane(nt);
            if (nt > kTM) PU_FAIL(12);
            PU_SYNC();
            // ---- contributors in the reference's linked-list order: (previous stored index desc, seq desc) ----
            if (nt <= 64) {
                for (int q = lane; q < nt; q += 64) {
                    const int t0 = tt[q], s0 = tseq[q];
                    int p = 0;
                    for (int q2 = 0; q2 < nt; ++q2) { const int t2 = tt[q2], s2 = tseq[q2]; p += (t2 > t0 || (t2 == t0 && s2 > s0)) ? 1 : 0; }
                    cx[p] = tx[q]; crem[p] = trem[q]; cv[p] = tv[q]; cnxt[p] = tnxt[q]; cw[p] = tw[q];
                }
            } else {
                // many contributors: sort the keys (t descending, seq descending; q rides in the low bits) -- scratch behind A's part of the entries
                unsigned long long *keys = reinterpret_cast<unsigned long long *>(eval + kNS);
                int N = 64;
                while (N < nt) N *= 2;
                for (int q = lane; q < N; q += 64)
                    keys[q] = q < nt ? (((unsigned long long)(unsigned)(0x7fffffff - tt[q]) << 32) | ((unsigned long long)(unsigned)(0xfffff - tseq[q]) << 12) | (unsigned)q)
                                     : ~0ull;
                PU_SYNC();
                wave_sort_u64<kGlobal>(keys, N, lane);
                for (int p = lane; p < nt; p += 64) {
                    const int q = (int)(keys[p] & 0xfffull);
                    cx[p] = tx[q]; crem[p] = trem[q]; cv[p] = tv[q]; cnxt[p] = tnxt[q]; cw[p] = tw[q];
                }
            }
            PU_SYNC();
            {
                // the chains go on to the contributors' next stored indices: tell those records where the chain stood in THIS list, and let
                // their steps go as far as this hand-over is concerned -- at once: the order is all they need from here, and rows of a dense
                // Schur complement, each handing every chain to the next, would otherwise run strictly one after the other
                unsigned long long *rb = Z ? A.recL : A.recU;
                for (int p = lane; p < nt; p += 64)
                    if (cnxt[p] >= 0) st_agent_i32(reinterpret_cast<int *>(rb + (size_t)cnxt[p] * 4 + 3) + 1, p + 1);
                drain_stores();
                PU_SYNC();
                for (int p = lane; p < nt; p += 64) {
                    const int nx = cnxt[p];
                    if (nx < 0) continue;
                    const int x = nx / T;
                    if (atomicAdd(&A.pending[x], -1) - 1 == 0) PU_PUSH(x);
                }
            }
            if (schur) {
                // the tails of the Schur rows: the part of the contributor's U row right of kterm (firstU stands there since the last
                // elimination step, ILUC.hpp:37-45)
                for (int p = lane; p < nt; p += 64) {
                    const int h = cw[p];
                    const int us = A.Ustart[h], ul = A.Ulen[h];
                    int a = 1, b = ul;                                    // first position in [1, ul) with column >= kterm
                    while (a < b) { const int mid = (a + b) >> 1; if (A.Uidx[us + mid] < kterm) a = mid + 1; else b = mid; }
                    cx[p] = us + a; crem[p] = ul - a;
                }
                PU_SYNC();
            }
            {
                int run = 0;
                for (int base = 0; base < nt; base += 64) {
                    const int p = base + lane;
                    const int v = p < nt ? crem[p] : 0;
                    int incl = v;
                    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
                    if (p < nt) cbase[p] = run + incl - v;
                    run += __shfl(incl, 63);
                }
                if (lane == 0) cbase[nt] = run;
            }
            PU_SYNC();
            const int net = __builtin_amdgcn_readfirstlane(cbase[nt]);
            const int ne = na + net;
            if (ne > kNE) PU_FAIL(13);
            // ---- tails: z takes (l(k,h) / Dinv[h]) * u(h, j >= k) from the contributor's U row, w takes (u(h,k) / Dinv[h]) * l(i >= k, h) ----
            {
                const int32_t *Oidx = Z ? A.Uidx : A.Lidx;
                const double *Oval = Z ? A.Uval : A.Lval;
                for (int e = lane; e < net; e += 64) {
                    int p = 0;
                    if (nt > 8) { int a = 0, b = nt; while (b - a > 1) { const int mid = (a + b) >> 1; if (cbase[mid] <= e) a = mid; else b = mid; } p = a; }
                    else while (cbase[p + 1] <= e) ++p;
                    const int xx = cx[p] + (e - cbase[p]);
                    erow[na + e] = ld_agent_i32(&Oidx[xx]);
                    eval[na + e] = cv[p] * ld_agent_f64(&Oval[xx]);
                }
            }
            PU_SYNC();
            // ---- slots in insertion order (hash: index -> slot + 1; the table only as large as this step's entries need) ----
            int hsize = 64;
            while (hsize < 2 * (ne + 1) && hsize < 4 * kNS) hsize *= 2;
            const unsigned hmask = (unsigned)(hsize - 1);
            for (int h = lane; h < hsize; h += 64) hslot[h] = 0;
            PU_SYNC();
            for (int e = lane; e < na; e += 64) {
                unsigned h = ((unsigned)erow[e] * 0x9E3779B1u) >> 7;
                for (;;) { h &= hmask; if (atomicCAS(&hslot[h], 0, e + 1) == 0) break; ++h; }
            }
            PU_SYNC();
            int ns = na;
            for (int base = na; base < ne; base += 64) {
                const int e = base + lane;
                const bool act = e < ne;
                const int r = act ? erow[e] : -1;
                int sl = -1;
                if (act) {
                    unsigned h = ((unsigned)r * 0x9E3779B1u) >> 7;
                    for (;;) { h &= hmask; const int v = hslot[h]; if (v == 0) break; if (srow[v - 1] == r) { sl = v - 1; break; } ++h; }
                }
                // the new indices of this batch, numbered in the order of their first entries: every lane that needs a slot claims the
                // hash position of its index (or finds the lane that did), the groups are ranked by their lowest lanes
                const bool need = act && sl < 0;
                int owner = lane, hpos = -1;
                if (lane < 64) crank[lane] = 64;
                PU_SYNC();
                if (need) {
                    unsigned h = ((unsigned)r * 0x9E3779B1u) >> 7;
                    for (;;) {
                        h &= hmask;
                        int v = hslot[h];
                        if (v == 0) {
                            v = atomicCAS(&hslot[h], 0, -(lane + 1));
                            if (v == 0) { hpos = (int)h; break; }
                        }
                        if (v < 0 && erow[base + (-v - 1)] == r) { owner = -v - 1; break; }
                        ++h;
                    }
                    atomicMin(&crank[owner], lane);
                }
                PU_SYNC();
                const int gfirst = need ? crank[owner] : 64;
                const unsigned long long firsts = __ballot(need && gfirst == lane);
                if (need) sl = ns + __popcll(firsts & ((1ull << gfirst) - 1ull));
                if (hpos >= 0) { hslot[hpos] = sl + 1; if (sl < kNS) srow[sl] = r; }
                ns += __popcll(firsts);
                PU_SYNC();
                if (act) eslot[e] = sl;
                if (ns > kNS) break;
            }
            ns = __builtin_amdgcn_readfirstlane(ns);
            if (ns > kNS) PU_FAIL(14);
            PU_SYNC();
            // ---- accumulate every slot sequentially over the entries (batches of 64 in order, inside a batch lowest lane first) ----
            for (int sl = lane; sl < ns; sl += 64) {
                sval[sl] = sl < na ? eval[sl] : 0.0;
                scnt[sl] = sl < na ? 1 : 0; srank[sl] = -1; crank[sl] = 64;
            }
            PU_SYNC();
            for (int base = na; base < ne; base += 64) {
                const int e = base + lane;
                bool rem = e < ne;
                const int sl = rem ? eslot[e] : 0;
                const double ev = rem ? eval[e] : 0.0;
                while (__ballot(rem) != 0ull) {
                    if (rem) atomicMin(&crank[sl], lane);
                    PU_SYNC();
                    const bool go = rem && crank[sl] == lane;
                    if (go) { sval[sl] = sval[sl] - ev; scnt[sl] += 1; }
                    PU_SYNC();
                    if (go) { crank[sl] = 64; rem = false; }
                    PU_SYNC();
                }
            }
            if (schur) {
                // ---- a row of the Schur complement: take_largest_elements_by_abs_value_with_threshold(list, max_fill_in, tau, kterm, n),
                // sparse_implementation.h:1322-1357 (every slot lies in the range): |v| > |row|_2 * tau, by increasing index ----
                double zz = 0.0;
                for (int s = 0; s < ns; ++s) { const double v = sval[s]; const double sq = v * v; zz = zz + sq; }
                const double thr = sqrt(zz) * A.tau;
                for (int s = lane; s < ns; s += 64) srank[s] = fabs(sval[s]) > thr ? 0 : -1;
                PU_SYNC();
                int nk = 0;
                for (int base = 0; base < ns; base += 64) { const int s = base + lane; nk += __popcll(__ballot(s < ns && srank[s] >= 0)); }
                nk = __builtin_amdgcn_readfirstlane(nk);
                if (nk > A.budget) {
                    // more than max_fill_in candidates: the largest by the reference's selection (:1341-1350)
                    if (lane == 0) {
                        int *ids = erow;
                        int c = 0;
                        for (int s = 0; s < ns; ++s) if (srank[s] >= 0) { eval[c] = fabs(sval[s]); ids[c] = s; ++c; }
                        if (A.budget > 0) select_largest(eval, ids, 0, c - 1, A.budget);
                        for (int q = 0; q < c - A.budget; ++q) srank[ids[q]] = -1;
                    }
                    nk = A.budget;
                    PU_SYNC();
                }
                if (uoff + nk > uend) {
                    const int take = nk > A.chunk ? nk : A.chunk;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&A.ctrl[6], take);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base < 0 || (long)base + take > (long)A.capS) PU_FAIL(16);
                    uoff = base; uend = base + take;
                }
                const int pos = uoff;
                uoff += nk;
                if (ns <= 128) {
                    for (int s = lane; s < ns; s += 64) {
                        if (srank[s] < 0) continue;
                        const int r0 = srow[s];
                        int r = 0;
                        for (int s2 = 0; s2 < ns; ++s2) r += (srank[s2] >= 0 && srow[s2] < r0) ? 1 : 0;
                        A.Sidx[pos + r] = r0; A.Sval[pos + r] = sval[s];
                    }
                } else {
                    unsigned long long *keys = reinterpret_cast<unsigned long long *>(eval);
                    int N = 64;
                    while (N < ns) N *= 2;
                    for (int s = lane; s < N; s += 64) keys[s] = (s < ns && srank[s] >= 0) ? (((unsigned long long)(unsigned)srow[s] << 32) | (unsigned)s) : ~0ull;
                    PU_SYNC();
                    wave_sort_u64<kGlobal>(keys, N, lane);
                    for (int r = lane; r < nk; r += 64) { const int s = (int)(unsigned)keys[r]; A.Sidx[pos + r] = srow[s]; A.Sval[pos + r] = sval[s]; }
                }
                if (lane == 0) { A.Sstart[k - kterm] = pos; A.Slen[k - kterm] = nk; }
                ++ndone;
                if (lane == 0 && ((k & 255) == 0 || ne > 2048)) atomicAdd(&A.ctrl[9], 1);      // progress mark
                PU_SYNC();
                break;
            }
            if (Z) {
                // ---- the pivot: its slot is A's first entry or was created by an update; z[k] inserts it otherwise (:1619, :1638) ----
                for (int base = 0; base < ns; base += 64) {
                    const int s = base + lane;
                    const unsigned long long mk = __ballot(s < ns && srow[s] == k);
                    if (mk) { dslot = base + __ffsll((long long)mk) - 1; break; }
                }
                dslot = __builtin_amdgcn_readfirstlane(dslot);
                if (dslot < 0) {
                    if (ns >= kNS) PU_FAIL(14);
                    if (lane == 0) { srow[ns] = k; sval[ns] = 0.0; scnt[ns] = 0; srank[ns] = -1; }
                    dslot = ns; ++ns;
                    PU_SYNC();
                }
                const double piv = sval[dslot];
                // ---- the level ends here? (:1619-1636) ----
                if (k > A.park_from && fabs(piv) < A.min_pivot) {
                    if (lane == 0) atomicMin(&A.ctrl[3], k);
                    parked = true;
                    break;
                }
                dinv = 1.0 / piv;
                dinv_store = piv == 0.0 ? 1.0 : dinv;                    // "zero pivot: setting diagonal to 1" (:1786-1792) -- for everyone after this step
                PU_SYNC();
                for (int s = lane; s < ns; s += 64) { const double v = sval[s] * dinv; zval[s] = s == dslot ? 0.0 : v; zcol[s] = srow[s]; zcnt[s] = s == dslot ? 0 : scnt[s]; }
                nzs = ns;
                PU_SYNC();
            } else {
                // ================= both working vectors are known: norms, dropping, the stores, touch records, counters =================
                for (int s = lane; s < ns; s += 64) sval[s] = sval[s] * dinv;                               // w.scale(Dinv[k]), :1676
                PU_SYNC();
                double n1w = 0.0, n1z = 0.0, n2w = 0.0, n2z = 0.0;                                             // vector_sparse_dynamic::norm1 / norm2, in slot order
                if (A.rules & (PILUC_DROP_ERR_PROP | PILUC_DROP_ERR_PROP2)) {
                    for (int s = 0; s < ns; ++s) n1w = n1w + fabs(sval[s]);
                    for (int s = 0; s < nzs; ++s) n1z = n1z + fabs(zval[s]);
                }
                if (A.rules & PILUC_DROP_STANDARD) {
                    for (int s = 0; s < ns; ++s) { const double sq = sval[s] * sval[s]; n2w = n2w + sq; }
                    for (int s = 0; s < nzs; ++s) { const double sq = zval[s] * zval[s]; n2z = n2z + sq; }
                    n2w = sqrt(n2w); n2z = sqrt(n2z);
                }
                const double weightU = piluc_weight(A, n2z, n1w, dinv), weightL = piluc_weight(A, n2w, n1z, dinv_store);
                // kept slots (index in [k+1, n), weight * |v| >= tau), by increasing index
                for (int s = lane; s < nzs; s += 64) { const double pr = weightU * fabs(zval[s]); zcnt[s] = (zcnt[s] & 0x3fffffff) | ((zcol[s] > k && pr >= A.tau) ? 0x40000000 : 0); }
                for (int s = lane; s < ns; s += 64) { const double pr = weightL * fabs(sval[s]); srank[s] = (srow[s] > k && pr >= A.tau) ? 0 : -1; }
                PU_SYNC();
                int nkz = 0, nkw = 0;
                for (int base = 0; base < nzs; base += 64) { const int s = base + lane; nkz += __popcll(__ballot(s < nzs && (zcnt[s] & 0x40000000))); }
                for (int base = 0; base < ns; base += 64) { const int s = base + lane; nkw += __popcll(__ballot(s < ns && srank[s] >= 0)); }
                nkz = __builtin_amdgcn_readfirstlane(nkz); nkw = __builtin_amdgcn_readfirstlane(nkw);
                if (nkz > A.budget || nkw > A.budget) {
                    // bounded fill: the max_fill_in - 1 largest products by the reference's selection (:1398-1405), candidates in slot order
                    if (lane == 0) {
                        int *ids = erow;
                        if (nkz > A.budget) {
                            int c = 0;
                            for (int s = 0; s < nzs; ++s) if (zcnt[s] & 0x40000000) { eval[c] = weightU * fabs(zval[s]); ids[c] = s; ++c; }
                            if (A.budget > 0) select_largest(eval, ids, 0, c - 1, A.budget);
                            for (int q = 0; q < c - A.budget; ++q) zcnt[ids[q]] &= 0x3fffffff;
                        }
                        if (nkw > A.budget) {
                            int c = 0;
                            for (int s = 0; s < ns; ++s) if (srank[s] >= 0) { eval[c] = weightL * fabs(sval[s]); ids[c] = s; ++c; }
                            if (A.budget > 0) select_largest(eval, ids, 0, c - 1, A.budget);
                            for (int q = 0; q < c - A.budget; ++q) srank[ids[q]] = -1;
                        }
                    }
                    if (nkz > A.budget) nkz = A.budget;
                    if (nkw > A.budget) nkw = A.budget;
                    PU_SYNC();
                }
                if (nzs <= 128) {
                    for (int s = lane; s < nzs; s += 64) {
                        if (!(zcnt[s] & 0x40000000)) continue;
                        const int c0 = zcol[s];
                        int r = 0;
                        for (int s2 = 0; s2 < nzs; ++s2) r += ((zcnt[s2] & 0x40000000) && zcol[s2] < c0) ? 1 : 0;
                        zkept[r + 1] = s;
                    }
                } else {
                    unsigned long long *keys = reinterpret_cast<unsigned long long *>(eval);
                    int N = 64;
                    while (N < nzs) N *= 2;
                    for (int s = lane; s < N; s += 64) keys[s] = (s < nzs && (zcnt[s] & 0x40000000)) ? (((unsigned long long)(unsigned)zcol[s] << 32) | (unsigned)s) : ~0ull;
                    PU_SYNC();
                    wave_sort_u64<kGlobal>(keys, N, lane);
                    for (int r = lane; r < nkz; r += 64) zkept[r + 1] = (int)(unsigned)keys[r];
                    PU_SYNC();
                }
                if (ns <= 128) {
                    for (int s = lane; s < ns; s += 64) {
                        if (srank[s] < 0) continue;
                        const int r0 = srow[s];
                        int r = 0;
                        for (int s2 = 0; s2 < ns; ++s2) r += (srank[s2] >= 0 && srow[s2] < r0) ? 1 : 0;
                        keptslot[r + 1] = s;
                    }
                } else {
                    unsigned long long *keys = reinterpret_cast<unsigned long long *>(eval);
                    int N = 64;
                    while (N < ns) N *= 2;
                    for (int s = lane; s < N; s += 64) keys[s] = (s < ns && srank[s] >= 0) ? (((unsigned long long)(unsigned)srow[s] << 32) | (unsigned)s) : ~0ull;
                    PU_SYNC();
                    wave_sort_u64<kGlobal>(keys, N, lane);
                    for (int r = lane; r < nkw; r += 64) keptslot[r + 1] = (int)(unsigned)keys[r];
                }
                PU_SYNC();
                // ---- the stores: row k of U = (k, 1), kept entries; column k of L likewise (:1779-1785, :1951-1958) ----
                if (uoff + nkz + 1 > uend) {
                    const int take = nkz + 1 > A.chunk ? nkz + 1 : A.chunk;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&A.ctrl[6], take);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base < 0 || (long)base + take > (long)A.capU) PU_FAIL(16);
                    uoff = base; uend = base + take;
                }
                if (loff + nkw + 1 > lend) {
                    const int take = nkw + 1 > A.chunk ? nkw + 1 : A.chunk;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&A.ctrl[7], take);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base < 0 || (long)base + take > (long)A.capL) PU_FAIL(16);
                    loff = base; lend = base + take;
                }
                const int posU = uoff, posL = loff;
                uoff += nkz + 1; loff += nkw + 1;
                if (lane == 0) {
                    st_agent_i32(&A.Uidx[posU], k); st_agent_f64(&A.Uval[posU], 1.0);
                    st_agent_i32(&A.Lidx[posL], k); st_agent_f64(&A.Lval[posL], 1.0);
                    A.Ustart[k] = posU; A.Ulen[k] = nkz + 1; A.Lstart[k] = posL; A.Llen[k] = nkw + 1;
                    A.Dinv[k] = dinv_store;
                }
                for (int r = 1 + lane; r <= nkz; r += 64) { const int s = zkept[r]; st_agent_i32(&A.Uidx[posU + r], zcol[s]); st_agent_f64(&A.Uval[posU + r], zval[s]); }
                for (int r = 1 + lane; r <= nkw; r += 64) { const int s = keptslot[r]; st_agent_i32(&A.Lidx[posL + r], srow[s]); st_agent_f64(&A.Lval[posL + r], sval[s]); }
                // ========================= touch records, then the counters =========================
                const int nws = ns;
                bool ovf = false;
                for (int r = 1 + lane; r <= nkw; r += 64) {
                    const int i = srow[keptslot[r]];
                    const int pos = atomicAdd(&A.cntL[i], 1);
                    if (pos >= T) ovf = true;
                    sridx[r] = i * T + pos;
                }
                for (int r = 1 + lane; r <= nkz; r += 64) {
                    const int j = zcol[zkept[r]];
                    const int pos = atomicAdd(&A.cntU[j], 1);
                    if (pos >= T) ovf = true;
                    zridx[r] = j * T + pos;
                }
                if (__ballot(ovf) != 0ull) PU_FAIL(15);
                PU_SYNC();
                // kind L, stored row i at position r: the step of row i will take (l(i,k) / Dinv[k]) times the tail u(k, j >= i) of THIS row of U
                for (int r = 1 + lane; r <= nkw; r += 64) {
                    const int s = keptslot[r];
                    const int i = srow[s];
                    const int lb = lower_pos(zkept, zcol, 1, nkz, i);                    // first kept column >= i
                    const int tprev = r > 1 ? srow[keptslot[r - 1]] : k;
                    const int nxt = r < nkw ? sridx[r + 1] : -1;
                    const double mult = sval[s] / dinv_store;
                    unsigned long long *rp = A.recL + (size_t)sridx[r] * 4;
                    st_agent_rec32(rp, cu_pack2(k, posU + lb), cu_pack2(tprev, nkz + 1 - lb), (unsigned long long)__double_as_longlong(mult), cu_pack2(nxt, 0));
                }
                // kind U, stored column j at position r: the step of column j will take (u(k,j) / Dinv[k]) times the tail l(i >= j, k) of THIS column of L
                for (int r = 1 + lane; r <= nkz; r += 64) {
                    const int s = zkept[r];
                    const int j = zcol[s];
                    const int ub = lower_pos(keptslot, srow, 1, nkw, j);                 // first kept row >= j
                    const int tprev = r > 1 ? zcol[zkept[r - 1]] : k;
                    const int nxt = r < nkz ? zridx[r + 1] : -1;
                    const double mult = zval[s] / dinv_store;
                    unsigned long long *rp = A.recU + (size_t)zridx[r] * 4;
                    st_agent_rec32(rp, cu_pack2(k, posL + ub), cu_pack2(tprev, nkw + 1 - ub), (unsigned long long)__double_as_longlong(mult), cu_pack2(nxt, 0));
                }
                drain_stores();
                // announcements first: x in R gets the kept columns below x, x in C the kept rows below x
                for (int r = 1 + lane; r <= nkw; r += 64) {
                    const int i = srow[keptslot[r]];
                    const int c = lower_pos(zkept, zcol, 1, nkz, i) - 1;                 // kept columns below i
                    if (c) atomicAdd(&A.pending[i], c);
                }
                for (int r = 1 + lane; r <= nkz; r += 64) {
                    const int j = zcol[zkept[r]];
                    const int c = lower_pos(keptslot, srow, 1, nkw, j) - 1;              // kept rows below j
                    if (c) atomicAdd(&A.pending[j], c);
                }
                // ... and every stored row (column) but the first waits for the step of the one before it in this column (row), which
                // hands the chain on and tells the record where in ITS list the chain stood
                for (int r = 2 + lane; r <= nkw; r += 64) atomicAdd(&A.pending[srow[keptslot[r]]], 1);
                for (int r = 2 + lane; r <= nkz; r += 64) atomicAdd(&A.pending[zcol[zkept[r]]], 1);
                __builtin_amdgcn_s_waitcnt(0);
                PU_SYNC();
                // then what this step decided: every slot of w and of z, as many times as it was announced (the slots of index k
                // themselves are this step's own business)
                for (int half2 = 0; half2 < 2; ++half2) {
                    const int cntn = half2 == 0 ? nws : nzs;
                    for (int s = lane; s < cntn; s += 64) {
                        const int x = half2 == 0 ? srow[s] : zcol[s];
                        const int d = half2 == 0 ? (x == k ? 0 : scnt[s]) : (zcnt[s] & 0x3fffffff);
                        if (d == 0) continue;
                        if (atomicAdd(&A.pending[x], -d) - d == 0) PU_PUSH(x);
                    }
                }
                ++ndone;
                if (lane == 0 && ((k & 255) == 0 || ne > 2048)) atomicAdd(&A.ctrl[9], 1);      // progress mark
                PU_SYNC();
            }
        }
        (void)parked;
        PU_RETIRE();
    }
    if (lane == 0 && ndone) atomicAdd(&A.ctrl[4], ndone);
#undef PU_FAIL
#undef PU_SYNC
#undef PU_RETIRE
#undef PU_OUT
#undef PU_PUSH
}

// ---- Schur mode: a row waits for the rows >= kterm that hand it a chain ----
__global__ void k_piluc_schur_pending(int32_t n, int32_t kterm, int32_t T, const int32_t *__restrict__ cntL, const unsigned long long *__restrict__ recL,
                                      int32_t *__restrict__ pendS)
{
    const int k = kterm + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int c = 0;
    const int nt = cntL[k] < T ? cntL[k] : T;
    for (int q = 0; q < nt; ++q) {
        const unsigned long long *r = recL + ((size_t)k * T + q) * 4;
        const int writer = (int)(unsigned)r[0], tprev = (int)(unsigned)r[1];
        if (writer < kterm && tprev >= kterm) ++c;
    }
    pendS[k] = c;
}
__global__ void k_piluc_schur_seed(int32_t n, int32_t kterm, int32_t nq, const int32_t *__restrict__ pendS, int32_t *rq, int32_t *ctrl)
{
    const int k = kterm + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (pendS[k] == 0) {
        const int q = k % nq;
        const int qcap = (n + nq - 1) / nq;
        atomicAdd(reinterpret_cast<unsigned long long *>(&ctrl[kCuQBase + 64 * q + 48]), (1ull << 32) + 1ull);
        rq[(size_t)q * qcap + atomicAdd(&ctrl[kCuQBase + 64 * q + 32], 1)] = k;
    }
}
__global__ void k_piluc_count_seeds(int32_t m, int32_t nq, const int32_t *__restrict__ pending, int32_t *ctrl)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m && pending[j] == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&ctrl[kCuQBase + 64 * (j % nq) + 48]), (1ull << 32) + 1ull);
}

// ---- the factors as the reference returns them: steps below kterm from the stores, (k, 1) for the others; compress() (:2053-2054,
// sparse_implementation.h:3696-3722) keeps |x| > 0 ----
__global__ void k_piluc_count(int32_t n, int32_t kterm, const int32_t *__restrict__ start, const int32_t *__restrict__ len, const double *__restrict__ val,
                              int32_t *__restrict__ cnt, int32_t *err)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n) return;
    if (k == n) { cnt[n] = 0; return; }
    if (k >= kterm) { cnt[k] = 1; return; }
    const int l = len[k];
    if (l <= 0) { atomicCAS(err, 0, 21); cnt[k] = 0; return; }      // a step below kterm that did not finish: must not happen
    int c = 0;
    const int s = start[k];
    for (int q = 0; q < l; ++q) c += fabs(val[s + q]) > 0.0 ? 1 : 0;
    cnt[k] = c;
}
__global__ void k_piluc_write(int32_t n, int32_t kterm, const int32_t *__restrict__ start, const int32_t *__restrict__ len, const int32_t *__restrict__ sidx,
                              const double *__restrict__ sval, const int32_t *__restrict__ optr, int32_t *__restrict__ oidx, double *__restrict__ oval)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int d = optr[k];
    if (k >= kterm) { oidx[d] = k; oval[d] = 1.0; return; }
    const int s = start[k], l = len[k];
    for (int q = 0; q < l; ++q) {
        const double v = sval[s + q];
        if (fabs(v) > 0.0) { oidx[d] = sidx[s + q]; oval[d] = v; ++d; }
    }
}
__global__ void k_piluc_fix_dinv(int32_t n, int32_t kterm, double *Dinv)
{
    const int k = kterm + blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) Dinv[k] = 1.0;
}
// the Schur complement (:2057-2065): rows in index order, columns shifted by kterm (compress() has nothing left to drop: a kept entry
// has |v| > |row| tau >= 0)
__global__ void k_piluc_schur_write(int32_t ns, int32_t kterm, const int32_t *__restrict__ start, const int32_t *__restrict__ len,
                                    const int32_t *__restrict__ sidx, const double *__restrict__ sval, const int32_t *__restrict__ optr,
                                    int32_t *__restrict__ oidx, double *__restrict__ oval)
{
    const int j = blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8;
    if (j >= ns) return;
    const int s = start[j], d = optr[j], l = len[j];
    for (int q = threadIdx.x % 8; q < l; q += 8) { oidx[d + q] = sidx[s + q] - kterm; oval[d + q] = sval[s + q]; }
}

static int scan_i32(hipStream_t st, const int32_t *in, int32_t *out, int count)
{
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, count, st));
    PoolBlock tmp;
    ILUPP_HIP(tmp.alloc(tb > 0 ? tb : 1));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, in, out, count, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    return ILUPP_OK;
}

struct HostLap {                               // wall-clock laps of an attempt (ILUPP_DEBUG)
    bool on; std::chrono::steady_clock::time_point t;
    explicit HostLap(bool o) : on(o), t(std::chrono::steady_clock::now()) {}
    void lap(hipStream_t st, const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(st);
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[ilupp] piluc:   %-28s %9.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

static void launch_class(int cls, int waves, hipStream_t st, const PilucArgs &a)
{
    if (cls == 0) hipLaunchKernelGGL((k_piluc_df<256, 128, 32, false>), dim3(waves), dim3(64), 0, st, a);
    else if (cls == 1) hipLaunchKernelGGL((k_piluc_df<768, 256, 64, false>), dim3(waves), dim3(64), 0, st, a);
    else if (cls == 2) hipLaunchKernelGGL((k_piluc_df<1024, 512, 128, false>), dim3(waves), dim3(64), 0, st, a);
    else if (cls == 3) hipLaunchKernelGGL((k_piluc_df<768, 256, 512, false>), dim3(waves), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_piluc_df<1, 1, 1, true>), dim3(waves), dim3(64), 0, st, a);
}

// one attempt with one capacity class and one store size; ILUPP_OK / an error / +1 = "outside this class" / +2 = "stores too small"
static int piluc_attempt(hipStream_t st, const DevMat &Av, const PilucParams &P, bool force_finish, double tau, DevMat *L, DevMat *U, double **Dinv_out,
                         DevMat *Anew, int32_t *kterm_out, float *kernel_ms, int cls, long store, int gns, long gne, int *which_capacity)
{
    const int32_t m = Av.n;
    int T = cls == 0 ? 32 : (cls == 1 ? 64 : (cls == 2 ? 128 : (cls == 3 ? 512 : 16)));
    if (cls == 4) { while (T < 4096 && T < m) T *= 2; }
    while (T > 16 && ((long)m * T > 0x7fffffffL || (size_t)m * T * 64 > ((size_t)64 << 30))) T /= 2;
    if ((long)m * T > 0x7fffffffL) return 1;
    if (store > 0x7ffffff0L) store = 0x7ffffff0L;

    HostLap laps(getenv("ILUPP_DEBUG") != nullptr && getenv("ILUPP_PILUC_LAPS") != nullptr);
    PoolBlock b_pending, b_colcnt, b_colptr, b_fillc, b_colpos, b_colord, b_rowof, b_Uidx, b_Lidx, b_Uval, b_Lval, b_Ustart, b_Ulen, b_Lstart, b_Llen,
              b_cntL, b_cntU, b_recL, b_recU, b_rq, b_ctrl, b_gws, b_dinv;
    const size_t nz = (size_t)(Av.nnz > 0 ? Av.nnz : 1);
    ILUPP_HIP(b_pending.alloc(sizeof(int32_t) * (size_t)m));
    ILUPP_HIP(b_colcnt.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_colptr.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_fillc.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_colpos.alloc(sizeof(int32_t) * nz));
    ILUPP_HIP(b_colord.alloc(sizeof(int32_t) * nz));
    ILUPP_HIP(b_rowof.alloc(sizeof(int32_t) * nz));
    ILUPP_HIP(b_Uidx.alloc(sizeof(int32_t) * (size_t)store));
    ILUPP_HIP(b_Lidx.alloc(sizeof(int32_t) * (size_t)store));
    ILUPP_HIP(b_Uval.alloc(sizeof(double) * (size_t)store));
    ILUPP_HIP(b_Lval.alloc(sizeof(double) * (size_t)store));
    ILUPP_HIP(b_Ustart.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_Ulen.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_Lstart.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_Llen.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_cntL.alloc(sizeof(int32_t) * (size_t)m));
    ILUPP_HIP(b_cntU.alloc(sizeof(int32_t) * (size_t)m));
    ILUPP_HIP(b_recL.alloc((size_t)m * T * 32));
    ILUPP_HIP(b_recU.alloc((size_t)m * T * 32));
    const size_t rq_len = (size_t)kCuQ * (size_t)((m + kCuQ - 1) / kCuQ) + (size_t)m;
    ILUPP_HIP(b_rq.alloc(sizeof(int32_t) * rq_len));
    const size_t ctrl_bytes = sizeof(int32_t) * (size_t)(kCuQBase + 64 * kCuQ);
    ILUPP_HIP(b_ctrl.alloc(ctrl_bytes));
    ILUPP_HIP(b_dinv.alloc(sizeof(double) * (size_t)m));
    int32_t *pending = b_pending.as<int32_t>(), *colcnt = b_colcnt.as<int32_t>(), *colptr = b_colptr.as<int32_t>(), *fillc = b_fillc.as<int32_t>(),
            *colpos = b_colpos.as<int32_t>(), *colord = b_colord.as<int32_t>(), *rowof = b_rowof.as<int32_t>(), *cntL = b_cntL.as<int32_t>(),
            *cntU = b_cntU.as<int32_t>(), *rq = b_rq.as<int32_t>(), *ctrl = b_ctrl.as<int32_t>();
    unsigned long long *recL = b_recL.as<unsigned long long>(), *recU = b_recU.as<unsigned long long>();
    ILUPP_HIP(hipMemsetAsync(pending, 0, sizeof(int32_t) * (size_t)m, st));
    ILUPP_HIP(hipMemsetAsync(colcnt, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(fillc, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(cntL, 0, sizeof(int32_t) * (size_t)m, st));
    ILUPP_HIP(hipMemsetAsync(cntU, 0, sizeof(int32_t) * (size_t)m, st));
    ILUPP_HIP(hipMemsetAsync(b_Ulen.p, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(b_Llen.p, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(rq, 0xff, sizeof(int32_t) * rq_len, st));
    ILUPP_HIP(hipMemsetAsync(ctrl, 0, ctrl_bytes, st));
    const int32_t big = 0x7fffffff;
    ILUPP_HIP(hipMemcpyAsync(ctrl + 3, &big, sizeof(int32_t), hipMemcpyHostToDevice, st));
    laps.lap(st, "allocations, memsets");
    const int gb = (m + 255) / 256;
    hipLaunchKernelGGL(k_iluc_prep, dim3(gb), dim3(256), 0, st, m, Av.ptr, Av.idx, pending, colcnt);
    hipLaunchKernelGGL(k_iluc_rowof, dim3(gb), dim3(256), 0, st, m, Av.ptr, rowof);
    { const int rc = scan_i32(st, colcnt, colptr, m + 1); if (rc) return rc; }
    { const int rc = iluc_column_order(st, m, Av, colcnt, colptr, fillc, colpos, rowof, colord); if (rc) return rc; }
    int waves = device_cu_count() * (cls == 0 ? 16 : (cls == 1 ? 8 : (cls == 2 ? 2 : (cls == 3 ? 2 : 4))));
    if (waves > m) waves = m;
    int gNE = 0, gNS = 0, gTM = 0;
    if (cls == 4) {
        gTM = T; gNS = gns;                                                     // (a power of two: the slot hash masks with a power of two <= 4 gNS)
        gNE = (int)(gne > (1L << 26) ? (1L << 26) : gne);
        if (gNE < 2 * gNS) gNE = 2 * gNS;                        // (scratch of the sorts lies behind the first gNS entries)
        while (waves > 16 && (size_t)waves * piluc_ws_bytes(gNE, gNS, gTM) > ((size_t)PILUC_WSGB << 30)) waves /= 2;
        if ((size_t)waves * piluc_ws_bytes(gNE, gNS, gTM) > ((size_t)PILUC_WSGB << 30)) return ILUPP_ERR_UNSUPPORTED;
        ILUPP_HIP(b_gws.alloc((size_t)waves * piluc_ws_bytes(gNE, gNS, gTM)));
    }
    laps.lap(st, "column order of A, work space");
    const int nq = waves < kCuQ ? waves : kCuQ;
    hipLaunchKernelGGL(k_piluc_count_seeds, dim3(gb), dim3(256), 0, st, m, nq, pending, ctrl);
    hipLaunchKernelGGL(k_iluc_seed, dim3(gb), dim3(256), 0, st, m, nq, pending, rq, ctrl);
    PilucArgs a;
    a.gws = b_gws.as<unsigned char>(); a.gNE = gNE; a.gNS = gNS; a.gTM = gTM;
    a.n = m; a.ptr = Av.ptr; a.idx = Av.idx; a.val = Av.val; a.colptr = colptr; a.colord = colord; a.rowof = rowof;
    a.tau = tau; a.min_pivot = P.min_pivot;
    a.park_from = big;
    if (!force_finish && P.small_pivot_terminates) {
        // k > MIN_ELIM_FACTOR * n (:1619), k an integer: k > floor(x) for x >= 0
        const double x = P.min_elim_factor * (double)m;
        a.park_from = x < 0.0 ? -1 : (x >= 2147483647.0 ? big : (int32_t)x);
    }
    a.rules = P.rules; a.combine = P.combine; a.scale_invdiag = P.scale_invdiag ? 1 : 0;
    for (int q = 0; q < 5; ++q) a.wgt[q] = P.wgt[q];
    a.neutral = P.neutral; a.min_weight = P.min_weight;
    int32_t max_fill = P.max_fill_in > 0 ? P.max_fill_in : m;             // ILUCDP.hpp:1440-1447: MAX_FILLIN_IS_INF => n; clamped to [1, n]
    if (max_fill < 1) max_fill = 1;
    if (max_fill > m) max_fill = m;
    a.budget = max_fill - 1;
    a.schur = 0; a.kterm = big; a.T = T; a.nq = nq;
    a.Uidx = b_Uidx.as<int32_t>(); a.Lidx = b_Lidx.as<int32_t>(); a.Uval = b_Uval.as<double>(); a.Lval = b_Lval.as<double>();
    a.capU = (int32_t)store; a.capL = (int32_t)store;
    {
        long c = store / (4L * waves);                      // (the waves' open chunks together: at most a quarter of a store)
        a.chunk = (int32_t)(c < 16 ? 16 : (c > 4096 ? 4096 : c));
    }
    a.Ustart = b_Ustart.as<int32_t>(); a.Ulen = b_Ulen.as<int32_t>(); a.Lstart = b_Lstart.as<int32_t>(); a.Llen = b_Llen.as<int32_t>();
    a.Dinv = b_dinv.as<double>();
    a.Sidx = nullptr; a.Sval = nullptr; a.capS = 0; a.Sstart = nullptr; a.Slen = nullptr;
    a.cntL = cntL; a.cntU = cntU; a.recL = recL; a.recU = recU; a.pending = pending; a.rq = rq; a.ctrl = ctrl;
    struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;
    ILUPP_HIP(hipEventCreate(&ev.a));
    ILUPP_HIP(hipEventCreate(&ev.b));
    ILUPP_HIP(hipEventRecord(ev.a, st));
    launch_class(cls, waves, st, a);
    ILUPP_HIP(hipGetLastError());
    int32_t h[8];
    ILUPP_HIP(hipMemcpyAsync(h, ctrl, 32, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    laps.lap(st, "elimination kernel");
    const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    if (dbg) {
        float ms = 0.f;
        ILUPP_HIP(hipEventRecord(ev.b, st));
        ILUPP_HIP(hipEventSynchronize(ev.b));
        ILUPP_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
        fprintf(stderr, "[ilupp] piluc: n %d class %d (slots %d) T %d waves %d store %ld: status %d kterm %d finished %d left %d cursors %d %d, %.1f ms\n", m, cls,
                cls == 4 ? gNS : 0, T, waves, store, h[2], h[3], h[4], h[5], h[6], h[7], ms);
    }
    if (h[2] == 2) return ILUPP_ERR_TIMEOUT;
    if (h[2] == 16) return 2;
    if (h[2] != 0) { if (which_capacity) *which_capacity = h[2]; return 1; }
    const int32_t kterm = h[3] == big ? m : h[3];
    *kterm_out = kterm;
    // ---- the Schur complement ----
    PoolBlock b_Sidx, b_Sval, b_Sstart, b_Slen, b_pendS;
    const int32_t nS = m - kterm;
    if (nS > 0) {
        long capS = store;
        ILUPP_HIP(b_Sidx.alloc(sizeof(int32_t) * (size_t)capS));
        ILUPP_HIP(b_Sval.alloc(sizeof(double) * (size_t)capS));
        ILUPP_HIP(b_Sstart.alloc(sizeof(int32_t) * (size_t)(nS + 1)));
        ILUPP_HIP(b_Slen.alloc(sizeof(int32_t) * (size_t)(nS + 1)));
        ILUPP_HIP(b_pendS.alloc(sizeof(int32_t) * (size_t)m));
        ILUPP_HIP(hipMemsetAsync(b_Slen.p, 0, sizeof(int32_t) * (size_t)(nS + 1), st));
        ILUPP_HIP(hipMemsetAsync(rq, 0xff, sizeof(int32_t) * rq_len, st));
        ILUPP_HIP(hipMemsetAsync(ctrl, 0, ctrl_bytes, st));
        const int gs = (nS + 255) / 256;
        hipLaunchKernelGGL(k_piluc_schur_pending, dim3(gs), dim3(256), 0, st, m, kterm, T, cntL, recL, b_pendS.as<int32_t>());
        hipLaunchKernelGGL(k_piluc_schur_seed, dim3(gs), dim3(256), 0, st, m, kterm, nq, b_pendS.as<int32_t>(), rq, ctrl);
        PilucArgs s = a;
        s.schur = 1; s.kterm = kterm; s.tau = tau * P.threshold_shift_schur;              // threshold *= threshold_Schur_factor, :1622
        s.budget = max_fill;                                                              // take_largest(..., max_fill_in, ...), :1717
        s.pending = b_pendS.as<int32_t>();
        s.Sidx = b_Sidx.as<int32_t>(); s.Sval = b_Sval.as<double>(); s.capS = (int32_t)capS; s.Sstart = b_Sstart.as<int32_t>(); s.Slen = b_Slen.as<int32_t>();
        launch_class(cls, waves, st, s);
        ILUPP_HIP(hipGetLastError());
        ILUPP_HIP(hipMemcpyAsync(h, ctrl, 32, hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        if (dbg) fprintf(stderr, "[ilupp] piluc: Schur rows %d: status %d finished %d left %d cursor %d\n", nS, h[2], h[4], h[5], h[6]);
        if (h[2] == 2) return ILUPP_ERR_TIMEOUT;
        if (h[2] == 16) return 2;
        if (h[2] != 0) { if (which_capacity) *which_capacity = h[2]; return 1; }
        if (h[4] != nS) { set_error("partialILUC: Schur rows left unprocessed"); return ILUPP_ERR_INTERNAL; }
    }
    laps.lap(st, "Schur complement");
    ILUPP_HIP(hipEventRecord(ev.b, st));
    // ---- the reference's arrays ----
    PoolBlock b_cnt, b_err;
    ILUPP_HIP(b_cnt.alloc(sizeof(int32_t) * (size_t)(m + 2)));
    ILUPP_HIP(b_err.alloc(64));
    ILUPP_HIP(hipMemsetAsync(b_err.p, 0, 64, st));
    DevMat *out[2] = {L, U};
    const int32_t *starts[2] = {a.Lstart, a.Ustart}, *lens[2] = {a.Llen, a.Ulen}, *sidx[2] = {a.Lidx, a.Uidx};
    const double *svals[2] = {a.Lval, a.Uval};
    const int gb1 = (m + 1 + 255) / 256;
    for (int d = 0; d < 2; ++d) {
        DevMat *M = out[d];
        hipLaunchKernelGGL(k_piluc_count, dim3(gb1), dim3(256), 0, st, m, kterm, starts[d], lens[d], svals[d], b_cnt.as<int32_t>(), b_err.as<int32_t>());
        ILUPP_HIP(pool_malloc(&M->ptr, sizeof(int32_t) * (size_t)(m + 1)));
        M->owns = true;
        { const int rc = scan_i32(st, b_cnt.as<int32_t>(), M->ptr, m + 1); if (rc) return rc; }
        int32_t nnz = 0, e = 0;
        ILUPP_HIP(hipMemcpyAsync(&nnz, M->ptr + m, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipMemcpyAsync(&e, b_err.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        if (e != 0) { set_error("partialILUC: a step below the end of the level did not finish"); return ILUPP_ERR_INTERNAL; }
        M->n = m; M->nnz = nnz; M->is_csr = d == 1;
        ILUPP_HIP(pool_malloc(&M->idx, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
        ILUPP_HIP(pool_malloc(&M->val, sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
        hipLaunchKernelGGL(k_piluc_write, dim3(gb), dim3(256), 0, st, m, kterm, starts[d], lens[d], sidx[d], svals[d], M->ptr, M->idx, M->val);
    }
    if (nS > 0) hipLaunchKernelGGL(k_piluc_fix_dinv, dim3((nS + 255) / 256), dim3(256), 0, st, m, kterm, a.Dinv);
    Anew->n = nS; Anew->nnz = 0; Anew->is_csr = true; Anew->owns = true;
    if (nS > 0) {
        ILUPP_HIP(pool_malloc(&Anew->ptr, sizeof(int32_t) * (size_t)(nS + 1)));
        { const int rc = scan_i32(st, b_Slen.as<int32_t>(), Anew->ptr, nS + 1); if (rc) return rc; }
        int32_t nnz = 0;
        ILUPP_HIP(hipMemcpyAsync(&nnz, Anew->ptr + nS, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        Anew->nnz = nnz;
        ILUPP_HIP(pool_malloc(&Anew->idx, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
        ILUPP_HIP(pool_malloc(&Anew->val, sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
        hipLaunchKernelGGL(k_piluc_schur_write, dim3((nS + 31) / 32), dim3(256), 0, st, nS, kterm, b_Sstart.as<int32_t>(), b_Slen.as<int32_t>(),
                           b_Sidx.as<int32_t>(), b_Sval.as<double>(), Anew->ptr, Anew->idx, Anew->val);
    }
    ILUPP_HIP(hipStreamSynchronize(st));
    laps.lap(st, "factors as CSR arrays");
    if (kernel_ms) { float ms = 0.f; ILUPP_HIP(hipEventElapsedTime(&ms, ev.a, ev.b)); *kernel_ms += ms; }
    *Dinv_out = b_dinv.as<double>();
    b_dinv.p = nullptr;                                   // (the caller owns it now)
    return ILUPP_OK;
}

// One level.  Av: the level's matrix in ROW storage.  L: unit lower factor by columns (the 1 first, rows ascending), U: unit upper factor by
// rows, Dinv: 1 / pivots (1 for the rows of the Schur complement), Anew: the Schur complement (0 x 0 when the elimination ran to the end).
int piluc_level(hipStream_t st, const DevMat &Av, const PilucParams &P, bool force_finish, double tau, DevMat *L, DevMat *U, double **Dinv, DevMat *Anew,
                int32_t *kterm, float *kernel_ms)
{
    if (Av.n < 1) return ILUPP_ERR_INVALID;
    long store = 3 * (long)Av.nnz + 2 * (long)Av.n + 1024;            // (the reference starts from MEM_FACTOR = 3 times nnz and doubles, :1483-1485, :1771-1773)
    int cls = 0;
    {
        const long avg = Av.nnz / Av.n + 1;
        // the kernel's time goes like 1 / (resident waves) -- a step is a few dozen dependent memory round trips -- and the smallest class has
        // the most waves (C5: 9 / 19 / 38 ms in classes 0 / 1 / 2); an attempt in a class that turns out too small costs at most its own short run
        cls = avg <= 16 ? 0 : (avg <= 40 ? 1 : 2);
        if (const char *e = getenv("ILUPP_PILUC_CLASS")) cls = atoi(e) < 0 ? 0 : (atoi(e) > 4 ? 4 : atoi(e));     // (experiments)
    }
    int rc = 1;
    int gns = 2048;
    long gne = 65536;
    bool chain_tried = getenv("ILUPP_NO_PILUC_CHAIN") != nullptr;
    const bool chain_off = chain_tried;
    // what the largest class cannot hold either goes to the chain (vectors in memory where they must: slow, but it has room for rows of 32 768
    // entries and 65 536 contributors) before it is refused -- matrices of more than 2^18 rows, which do not try the chain first
    auto last_resort = [&](const char *why) -> int {
        if (chain_tried || chain_off) { set_error(why); return ILUPP_ERR_UNSUPPORTED; }
        chain_tried = true;
        L->release(); U->release(); Anew->release();
        const int rcc = piluc_chain_level(st, Av, P, force_finish, tau, L, U, Dinv, Anew, kterm, kernel_ms, true);
        if (rcc == 1) { set_error(why); return ILUPP_ERR_UNSUPPORTED; }
        return rcc;
    };
    while (rc == 1 || rc == 2) {
        int which = 0;
        L->release(); U->release(); Anew->release();
        if (cls == 4 && !chain_tried && Av.n <= (1 << 18)) {           // (a chain of more steps than that costs tens of seconds before it can find out that a row does not fit)
            // rows too long for the LDS classes: the steps of such a factorisation depend on one another almost one by one, and the largest
            // class walks them through global-memory slots (700 us per step on the critical path).  The chain kernel walks them in LDS, and from
            // the step whose vectors outgrow LDS on, in global memory (fill of 88 entries per row, n = 2e4: 5.3 s against 14.8 s in the largest class).
            chain_tried = true;
            const int rcc = piluc_chain_level(st, Av, P, force_finish, tau, L, U, Dinv, Anew, kterm, kernel_ms, getenv("ILUPP_NO_PILUC_MEM_CHAIN") == nullptr);
            if (rcc != 1) { rc = rcc; break; }
            L->release(); U->release(); Anew->release();
        }
        rc = piluc_attempt(st, Av, P, force_finish, tau, L, U, Dinv, Anew, kterm, kernel_ms, cls, store, gns, gne, &which);
        if (rc == 1) {
            const bool records = which == 12 || which == 15;          // more touch records per step than the class holds
            if (cls < 4) {
                // (class 3 has the records of many steps but no more entries or slots than class 2)
                cls = records ? (cls < 2 ? 2 : cls + 1) : (cls < 2 ? cls + 1 : 4);
                // (an attempt in the largest class takes seconds where the others take milliseconds, and rows that long fill the factors: it
                //  starts with stores that a "stores too small" is unlikely to send back to the start -- 16 nnz + 8 n entries per factor)
                if (cls == 4) { const long big = 16 * (long)Av.nnz + 8 * (long)Av.n + 1024; if (store < big) store = big; }
            } else if (records) {
                rc = last_resort("partialILUC: a step is reached by more than 4096 stored entries");
            } else if (which == 13) {                                  // the entries of a working row: at most (contributors) x (row length)
                if (gne >= (1L << 26)) rc = last_resort("partialILUC: a working row gathers more than 2^26 entries");
                else gne *= 4;
            } else {                                                   // its slots: at most n + 1
                if (gns > 2 * (long)Av.n + 2) rc = last_resort("partialILUC: a working row does not fit the largest capacity class");
                else { gns *= 4; if (gne < 8L * gns) gne = 8L * gns; }
            }
        } else if (rc == 2) {
            if (store >= 0x7ffffff0L) { set_error("partialILUC: the factors of a level exceed 2^31 entries"); rc = ILUPP_ERR_MEMORY; break; }
            store *= cls == 4 ? 4 : 2;
        } else if (rc == ILUPP_ERR_UNSUPPORTED) {
            set_error("partialILUC: the working arrays of the largest capacity class exceed the memory set aside for them");
        }
    }
    if (rc != ILUPP_OK) { L->release(); U->release(); Anew->release(); }
    return rc;
}

}  // namespace ilupp
