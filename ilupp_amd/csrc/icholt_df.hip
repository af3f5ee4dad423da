// ilupp_amd/csrc/icholt_df.hip -- ICholT (reference IChol.hpp:78-164, ILUC.hpp:31-63, dropping.hpp:8-34) as a
// DATAFLOW computation over columns for gfx950: one wave per column, columns start as soon as everything that
// reaches them has finished, bit-identical to the reference's strictly sequential column loop.
//
// Why this is not a level schedule.  The pattern of L depends on the values (top-k by magnitude, threshold), and a
// column j is reached not only by the columns stored in row j but by every column whose PRE-drop working column
// had row j (they all update the running diagonal D[j], IChol.hpp:135-141).  Nothing of this is known ahead.  What
// IS known: every reach of column k on row i (k < i) has a cause that is visible earlier --
//     (a) A(i,k) != 0, or
//     (b) some finished column k' stores both rows k and i (k' < k < i): then column k will pull the tail of
//         column k' and create/update its slot for row i (IChol.hpp:120-132).
// So every row carries a counter `pending[i]` of announced-but-not-yet-executed reaches: it starts at the number of
// entries left of the diagonal in row i of A (a); a finishing column k' adds, for every stored row i, the number of
// stored rows between k' and i (b), and takes off one per entry it scattered/updated itself.  An unannounced reach
// always has an announced, unfinished ancestor, so the counter reaches zero exactly once: when every column that
// reaches i has finished.  That column then goes to the ready queue.  No symbolic bound, no speculation.
//
// Order-dependent arithmetic, reproduced exactly:
//   * D[j] = ((0 - w_{k1,j}^2) - w_{k2,j}^2 ...) + A_jj in ascending column order: every reach leaves a 32-byte
//     TOUCH record {k, value, ...} in row j's record list (dropped entries too); column j sorts them by k and sums.
//   * The contributing columns are subtracted in the order of the reference's re-threaded linked list
//     (ILUC.hpp:37-63: a column moves from the list of row t to the head of the list of its next stored row when
//     column t has been processed; t itself is pushed first).  Traversal of list j = columns ordered by
//     (previous stored row t DESCENDING, then REVERSE of their order in list t, t itself last).  Column t writes the
//     position (`seq`) of each of its contributors into the touch record of that contributor's next stored row, so
//     column j orders its contributors by (t desc, seq desc) -- no list is ever built.
//   * The working column: slots in insertion order (A's column, then new rows in the order the contributors'
//     tails bring them), every slot accumulated sequentially over the contributors, separate multiply/subtract;
//     2-norm summed in slot order; candidates |w| > norm*tau; top-k by magnitude with std::sort's semantics (the
//     kept SET only depends on the sort algorithm if the k-th and (k+1)-th magnitudes are equal: then one lane runs
//     the libstdc++ algorithm, otherwise a parallel ranking); kept entries by increasing row.
//
// Columns land in fixed slabs (capacity = column length of A + add_fill_in, the top-k budget, IChol.hpp:144-145);
// a compaction pass produces the CSC arrays.  Whatever exceeds the static LDS capacity classes (long working columns, many
// reaches per row) runs in the largest class: the same kernel with one wave per CU, working columns of up to 2048 rows in LDS,
// the gathered entries (up to 262144 per column) and the touch records of a row (up to 65536) in global memory; beyond that
// the factorisation is refused.
// A column that loses its diagonal (indefinite input: the pivot's square root is NaN; a negative add_fill_in that leaves a
// budget below 1) is an error here ("not positive definite"), where the reference silently returns a factor with EMPTY columns
// (a NaN pivot makes every entry of its column NaN and none of them passes the threshold test); ILUPP_REFERENCE_NANS=1 gives that factor.
#include <stdlib.h>

#include <mutex>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include <vector>
#include "stdsort.h"

namespace ilupp {

// LDS capacities come in three classes (template parameters of the kernel): NE = entries gathered per column (A's column +
// the contributors' tails), NS = slots of the working column, TM = touch records per row.  The tiny class keeps 24 waves per CU resident, the small one 12 (the kernel is bound by the latency of its ~8 dependent memory round trips per column, so resident waves
// are throughput); a factorisation that exceeds it is run again with the large class, then by the sequential kernel.
static constexpr int kCtTmax = 128;   // touch records per row, large class (run-time T <= this)
static constexpr int kCtTsmall = 64;
static constexpr int kCtTtiny = 32;
// ready queues: kCtQ of them, column i goes to queue i % kCtQ (so the number of entries a queue will ever get is known:
// a worker whose ticket lies beyond it is done), wave w serves queue w % kCtQ.  One queue would put two atomics per
// column on two addresses -- 33 M same-address atomics at 256^3, which alone take longer than the factorisation.
static constexpr int kCtQ = 64;     // at most; fewer when there are fewer waves than that (every queue needs a worker)
static constexpr int kCtQBase = 64;   // ctrl word of queue 0's head; queue q: head at kCtQBase + 64 q, tail 32 words later
static constexpr unsigned kCtSpinLimit = 1u << 22;

// ctrl: [2] error (1 = capacity exceeded -> next class, 2 = timeout), [3] malformed column, [4] finished columns (progress), then the queues' heads/tails
__global__ void k_ict_prep(int32_t m, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, int32_t add,
                           int32_t *__restrict__ cap, int32_t *pending, int32_t *ctrl)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int c0 = Aptr[j], c1 = Aptr[j + 1];
    if (c0 >= c1 || Aidx[c0] != j) { atomicMin(&ctrl[3], j); cap[j] = 0; return; }     // IChol.hpp:105-107
    cap[j] = (c1 - c0) + add > 0 ? (c1 - c0) + add : 0;
    for (int x = c0 + 1; x < c1; ++x) atomicAdd(&pending[Aidx[x]], 1);
}

__global__ void k_ict_seed(int32_t m, int32_t nq, const int32_t *__restrict__ pending, int32_t *rq, int32_t *ctrl)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    if (pending[j] == 0) {
        const int q = j % nq;
        const int qcap = (m + nq - 1) / nq;
        rq[(size_t)q * qcap + atomicAdd(&ctrl[kCtQBase + 64 * q + 32], 1)] = j;
    }
}

// largest class: bytes of the entry / slot arrays (dynamic LDS) and of one wave's touch arrays (global workspace)
__host__ __device__ inline size_t ict_lds_bytes(int ns) { return ((size_t)ns * (36 + 16) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t ict_touch_bytes(int tm) { return ((size_t)tm * 64 + 4 + 63) & ~(size_t)63; }

__device__ __forceinline__ unsigned long long pack2(int lo, int hi)
{
    return (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32);
}

// kGlobal: the largest class, one wave per CU.  Entry and slot arrays are carved out of one dynamic LDS allocation of (almost)
// the whole 160 KB of a CU, the touch arrays live in a per-wave slice of global memory, sizes (gNE, gNS, gTM) are run-time
// values -- same code, same order of operations; what the static classes cannot hold (long working columns, rows reached
// by many columns) runs here instead of on a one-lane loop.
template <int kLdsNE, int kLdsNS, int kLdsTM, bool kGlobal>
__global__ void __launch_bounds__(64)
k_icholt_df(int32_t m, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval,
            int32_t add, double tau, int32_t T, int32_t nq, const int32_t *__restrict__ Loff,
            int32_t *Lidx, double *Lval, int32_t *Llen,
            int32_t *cnt, unsigned long long *rec, int32_t *pending, int32_t *rq, int32_t *ctrl,
            unsigned char *gws, int32_t gNE, int32_t gNS, int32_t gTM, int32_t refnans)
{
    __shared__ int s_erow[kLdsNE], s_eslot[kLdsNE];
    __shared__ double s_eval[kLdsNE];
    __shared__ int s_srow[kLdsNS], s_scnt[kLdsNS], s_srank[kLdsNS], s_sridx[kLdsNS], s_cand[kLdsNS], s_crank[kLdsNS], s_keptslot[kLdsNS];
    __shared__ double s_sval[kLdsNS];
    __shared__ int s_hslot[4 * kLdsNS];
    __shared__ int s_tk[kLdsTM], s_tx[kLdsTM], s_tt[kLdsTM], s_trem[kLdsTM], s_tnxt[kLdsTM], s_tseq[kLdsTM];
    __shared__ double s_tv[kLdsTM], s_ordv[kLdsTM];
    __shared__ int s_cx[kLdsTM], s_crem[kLdsTM], s_cbase[kLdsTM + 1], s_cnxt[kLdsTM];
    __shared__ double s_cv[kLdsTM];
    __shared__ double tie_mag[2];
    const int kCtNE = kGlobal ? gNE : kLdsNE, kCtNS = kGlobal ? gNS : kLdsNS, kCtTM = kGlobal ? gTM : kLdsTM;
    int *erow = s_erow, *eslot = s_eslot, *srow = s_srow, *scnt = s_scnt, *srank = s_srank, *sridx = s_sridx, *cand = s_cand,
        *crank = s_crank, *keptslot = s_keptslot, *tk = s_tk, *tx = s_tx, *tt = s_tt, *trem = s_trem, *tnxt = s_tnxt, *tseq = s_tseq,
        *cx = s_cx, *crem = s_crem, *cbase = s_cbase, *cnxt = s_cnxt, *hslot = s_hslot;
    double *eval = s_eval, *sval = s_sval, *tv = s_tv, *ordv = s_ordv, *cv = s_cv;
    if (kGlobal) {
        // slots and their hash: dynamic LDS (doubles first); gathered entries and the touch records of the row: this wave's
        // slices of a global workspace (streamed / read one element at a time on all lanes -- broadcast loads that pipeline --
        // and their number grows with the matrix: a row of a matrix with much fill is reached by hundreds of columns)
        extern __shared__ __attribute__((aligned(16))) unsigned char ict_dyn[];
        double *d = reinterpret_cast<double *>(ict_dyn);
        sval = d; d += gNS;
        int *q = reinterpret_cast<int *>(d);
        srow = q; q += gNS; scnt = q; q += gNS; srank = q; q += gNS; sridx = q; q += gNS; cand = q; q += gNS; crank = q; q += gNS; keptslot = q; q += gNS;
        hslot = q; q += 4 * gNS;
        // gathered entries: behind this wave's touch arrays in the global workspace
        double *ge = reinterpret_cast<double *>(gws + (size_t)gridDim.x * ict_touch_bytes(gTM) + (size_t)blockIdx.x * ((size_t)gNE * 16));
        eval = ge; ge += gNE;
        erow = reinterpret_cast<int *>(ge); eslot = erow + gNE;
        double *gd = reinterpret_cast<double *>(gws + (size_t)blockIdx.x * ict_touch_bytes(gTM));
        tv = gd; gd += gTM; ordv = gd; gd += gTM; cv = gd; gd += gTM;
        int *gq = reinterpret_cast<int *>(gd);
        tk = gq; gq += gTM; tx = gq; gq += gTM; tt = gq; gq += gTM; trem = gq; gq += gTM; tnxt = gq; gq += gTM; tseq = gq; gq += gTM;
        cx = gq; gq += gTM; crem = gq; gq += gTM; cbase = gq; gq += gTM + 1; cnxt = gq; gq += gTM;
    }

    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#define CT_FAIL(code) do { if (lane == 0) atomicCAS(&ctrl[2], 0, (code)); return; } while (0)

    const int myq = (int)(blockIdx.x % nq);
    const int qcap = (m + nq - 1) / nq;
    const int qtotal = (m - myq + nq - 1) / nq;              // columns i = myq (mod nq) below m
    int32_t *const qhead = &ctrl[kCtQBase + 64 * myq];
    const int32_t *const myrq = rq + (size_t)myq * qcap;
    for (;;) {
        int tkt = 0;
        if (lane == 0) tkt = atomicAdd(qhead, 1);
        tkt = __builtin_amdgcn_readfirstlane(tkt);
        if (tkt >= qtotal) break;
        int j;
        unsigned spins = 0;
        int last_done = -1;
        for (;;) {
            j = ld_agent_i32(&myrq[tkt]);
            if (j >= 0) break;
            if ((++spins & 63u) == 0) {
                if (ld_agent_i32(&ctrl[2]) != 0) return;
                // the limit is on the time WITHOUT PROGRESS anywhere (ctrl[4] counts finished columns): the last columns of a small
                // dense-ish matrix with a large fill budget are reached by a thousand earlier ones each and form a chain -- a hundred
                // of them took longer than a fixed limit on the wait (found by profiles/tools/fuzz_kernels.py, seed 1245)
                const int done = ld_agent_i32(&ctrl[4]);
                if (done != last_done) { last_done = done; spins = 0; }
                else if (spins > kCtSpinLimit) CT_FAIL(2);
            }
            __builtin_amdgcn_s_sleep(4);
        }
        j = __builtin_amdgcn_readfirstlane(j);

        const int c0 = __builtin_amdgcn_readfirstlane(Aptr[j]);
        const int clen = __builtin_amdgcn_readfirstlane(Aptr[j + 1]) - c0;
        const int capj = clen + add;
        const int loff = __builtin_amdgcn_readfirstlane(Loff[j]);
        const int nt = __builtin_amdgcn_readfirstlane(ld_agent_i32(&cnt[j]));
        if (clen > kCtNS || nt > T || nt > kCtTM) CT_FAIL(1);

        // ---- the column of A (slots 0..clen-1, IChol.hpp:110-112) and the touch records of row j ----
        for (int e = lane; e < clen; e += 64) { erow[e] = Aidx[c0 + e]; eval[e] = Aval[c0 + e]; eslot[e] = e; srow[e] = erow[e]; }
        for (int q = lane; q < nt; q += 64) {
            const Rec32 rr = ld_agent_rec32(rec + ((size_t)j * T + q) * 4);
            const unsigned long long w0 = rr.w[0], w1 = rr.w[1], w2 = rr.w[2], w3 = rr.w[3];
            tk[q] = (int)(unsigned)w0; tx[q] = (int)(unsigned)(w0 >> 32);
            tt[q] = (int)(unsigned)w1; trem[q] = (int)(unsigned)(w1 >> 32);
            tv[q] = __longlong_as_double((long long)w2);
            tnxt[q] = (int)(unsigned)w3; tseq[q] = (int)(unsigned)(w3 >> 32);
        }
        __syncthreads();

        // ---- D[j]: the squares of all reaches in ascending column order, then + A_jj  (:115, :139) ----
        int nc = 0;
        for (int q = lane; q < nt; q += 64) {
            const int kq = tk[q];
            int r = 0;
            for (int q2 = 0; q2 < nt; ++q2) r += (tk[q2] < kq) ? 1 : 0;
            ordv[r] = tv[q];
        }
        for (int q2 = 0; q2 < nt; ++q2) nc += (tx[q2] >= 0) ? 1 : 0;
        nc = __builtin_amdgcn_readfirstlane(nc);
        __syncthreads();
        double D = 0.0;
        for (int r = 0; r < nt; ++r) { const double v = ordv[r]; const double sq = v * v; D = D - sq; }
        D = D + eval[0];
        const double Ljj = sqrt(D);                                                  // :116

        // ---- contributors in the reference's linked-list order: (previous stored row desc, seq desc) ----
        for (int q = lane; q < nt; q += 64) {
            if (tx[q] < 0) continue;
            const int t0 = tt[q], s0 = tseq[q];
            int p = 0;
            for (int q2 = 0; q2 < nt; ++q2) {
                if (tx[q2] < 0) continue;
                const int t2 = tt[q2], s2 = tseq[q2];
                p += (t2 > t0 || (t2 == t0 && s2 > s0)) ? 1 : 0;
            }
            cx[p] = tx[q]; crem[p] = trem[q]; cv[p] = tv[q]; cnxt[p] = tnxt[q];
        }
        __syncthreads();
        for (int p = lane; p < nc; p += 64)
            if (cnxt[p] >= 0) st_agent_i32(reinterpret_cast<int *>(rec + (size_t)cnxt[p] * 4 + 3) + 1, p + 1);
        if (lane == 0) {
            int s = 0;
            for (int p = 0; p < nc; ++p) { cbase[p] = s; s += crem[p]; }
            cbase[nc] = s;
        }
        __syncthreads();
        const int net = __builtin_amdgcn_readfirstlane(cbase[nc]);
        const int ne = clen + net;
        if (ne > kCtNE) CT_FAIL(1);

        // ---- tails of the contributing columns: entry e = (row, L_ik * L_jk)  (:122-131) ----
        for (int e = lane; e < net; e += 64) {
            int p = 0;
            while (cbase[p + 1] <= e) ++p;
            const int xx = cx[p] + 1 + (e - cbase[p]);
            const int row = ld_agent_i32(&Lidx[xx]);
            const double lv = ld_agent_f64(&Lval[xx]);
            erow[clen + e] = row;
            eval[clen + e] = lv * cv[p];
        }
        __syncthreads();

        // ---- slots in insertion order: a row's slot is created by its first occurrence (hash: row -> slot + 1) ----
        const unsigned hmask = (unsigned)(4 * kCtNS - 1);
        for (int h = lane; h < 4 * kCtNS; h += 64) hslot[h] = 0;
        __syncthreads();
        for (int e = lane; e < clen; e += 64) {                        // A's column: distinct rows, slots 0..clen-1
            unsigned h = ((unsigned)erow[e] * 0x9E3779B1u) >> 7;
            for (;;) { h &= hmask; if (atomicCAS(&hslot[h], 0, e + 1) == 0) break; ++h; }
        }
        __syncthreads();
        int ns = clen;
        for (int base = clen; base < ne; base += 64) {
            const int e = base + lane;
            const bool act = e < ne;
            const int r = act ? erow[e] : -1;
            int sl = -1;
            if (act) {
                unsigned h = ((unsigned)r * 0x9E3779B1u) >> 7;
                for (;;) { h &= hmask; const int v = hslot[h]; if (v == 0) break; if (srow[v - 1] == r) { sl = v - 1; break; } ++h; }
            }
            // rows this batch sees for the first time, in lane order; equal rows share the slot their first lane creates
            bool need = act && sl < 0;
            unsigned long long todo = __ballot(need);
            while (todo != 0ull) {
                const int leader = __ffsll((long long)todo) - 1;
                const int lr = __shfl(r, leader);
                const bool same = need && r == lr;
                if (same) { sl = ns; need = false; }
                if (lane == leader && ns < kCtNS) {
                    srow[ns] = lr;
                    unsigned h = ((unsigned)lr * 0x9E3779B1u) >> 7;
                    for (;;) { h &= hmask; if (hslot[h] == 0) { hslot[h] = ns + 1; break; } ++h; }
                }
                ++ns;
                todo &= ~__ballot(same);
                __syncthreads();
            }
            if (act) eslot[e] = sl;
            if (ns > kCtNS) break;
        }
        ns = __builtin_amdgcn_readfirstlane(ns);
        if (ns > kCtNS) CT_FAIL(1);
        __syncthreads();

        // ---- accumulate every slot sequentially over the entries, scale by 1/L_jj  (:126-131, :135-141) ----
        // batches of 64 entries in order; inside a batch the entries of one slot go one after the other, lowest lane first
        for (int sl = lane; sl < ns; sl += 64) {
            sval[sl] = sl == 0 ? Ljj : (sl < clen ? eval[sl] : 0.0);                // :117
            scnt[sl] = sl < clen ? 1 : 0; srank[sl] = -1; crank[sl] = 64;
        }
        __syncthreads();
        for (int base = clen; base < ne; base += 64) {
            const int e = base + lane;
            bool rem = e < ne;
            const int sl = rem ? eslot[e] : 0;
            const double ev = rem ? eval[e] : 0.0;
            while (__ballot(rem) != 0ull) {
                if (rem) atomicMin(&crank[sl], lane);
                __syncthreads();
                const bool go = rem && crank[sl] == lane;
                if (go) { sval[sl] = sval[sl] - ev; scnt[sl] += 1; }
                __syncthreads();
                if (go) { crank[sl] = 64; rem = false; }
                __syncthreads();
            }
        }
        for (int sl = 1 + lane; sl < ns; sl += 64) sval[sl] = sval[sl] / Ljj;
        __syncthreads();

        // ---- threshold_and_drop(w, list, capj, tau, j, m)  (dropping.hpp:8-34) ----
        double z = 0.0;
        for (int s = 0; s < ns; ++s) { const double v = sval[s]; const double sq = v * v; z = z + sq; }
        const double thr = sqrt(z) * tau;
        int ncand = 0;
        for (int base = 0; base < ns; base += 64) {
            const int s = base + lane;
            const bool is = s < ns && fabs(sval[s]) > thr;
            const unsigned long long mask = __ballot(is);
            if (is) cand[ncand + __popcll(mask & lt_mask)] = s;
            ncand += __popcll(mask);
        }
        ncand = __builtin_amdgcn_readfirstlane(ncand);
        __syncthreads();
        int nk = ncand;
        if (ncand > capj) {
            nk = capj;
            for (int c = lane; c < ncand; c += 64) {
                const double a = fabs(sval[cand[c]]);
                int r = 0;
                for (int c2 = 0; c2 < ncand; ++c2) { const double a2 = fabs(sval[cand[c2]]); r += (a2 > a || (a2 == a && c2 < c)) ? 1 : 0; }
                crank[c] = r;
                if (r == capj - 1) tie_mag[0] = a;
                if (r == capj) tie_mag[1] = a;
            }
            __syncthreads();
            const bool tie = __builtin_amdgcn_readfirstlane((ncand > 16 && tie_mag[0] == tie_mag[1]) ? 1 : 0) != 0;
            if (tie) {
                // equal magnitudes across the cut: the kept set is whatever libstdc++'s introsort leaves in front
                if (lane == 0) {
                    c_sort_slots_by_abs_desc(cand, ncand, sval);
                    for (int c = 0; c < capj; ++c) srank[cand[c]] = 0;
                }
            } else {
                for (int c = lane; c < ncand; c += 64) if (crank[c] < capj) srank[cand[c]] = 0;
            }
        } else {
            for (int c = lane; c < ncand; c += 64) srank[cand[c]] = 0;
        }
        __syncthreads();
        // kept entries by increasing row (dropping.hpp:32-33); srank = position in the stored column
        for (int s = lane; s < ns; s += 64) {
            if (srank[s] < 0) continue;
            const int r0 = srow[s];
            int r = 0;
            for (int s2 = 0; s2 < ns; ++s2) r += (srank[s2] >= 0 && srow[s2] < r0) ? 1 : 0;
            crank[s] = r;                       // (crank is free again)
            keptslot[r] = s;
        }
        __syncthreads();
        for (int s = lane; s < ns; s += 64) if (srank[s] >= 0) srank[s] = crank[s];
        __syncthreads();
        // the reference assumes the diagonal leads the stored column (firstL = pointer+1, ILUC.hpp:43).  A column that lost
        // it (the pivot became negative: NaN column; a budget too small to keep it) is reported: the reference goes on and
        // returns a factor full of NaN / with misplaced "diagonals" for such an input
        //   3: the pivot is NaN (the matrix is not positive definite); 4: a finite pivot that the threshold or the top-k budget dropped
        if (nk < 1 || __builtin_amdgcn_readfirstlane(srank[0]) != 0) {
            const double piv = sval[0];
            // (ILUPP_REFERENCE_NANS=1: what the reference does with a NaN pivot -- every entry of the column is NaN, none passes
            // `|w| > norm * tau`, the column is stored EMPTY, and its NaNs reach the diagonals of the rows it touches through the
            // records below, whose columns end the same way: IChol.hpp:115-141, dropping.hpp:15-21)
            if (!(refnans != 0 && piv != piv && nk == 0)) CT_FAIL(piv != piv ? 3 : 4);
        }

        // ---- append  (sparse_implementation.h:3170-3186) ----
        for (int s = lane; s < ns; s += 64)
            if (srank[s] >= 0) { st_agent_i32(&Lidx[loff + srank[s]], srow[s]); st_agent_f64(&Lval[loff + srank[s]], sval[s]); }
        if (lane == 0) {
            Llen[j] = nk;
            if (nt > 64 || (j & 255) == 0) atomicAdd(&ctrl[4], 1);    // progress mark (not from every column: one address; the slow columns are those with many records)
        }

        // ---- one touch record per sub-diagonal slot, in the record list of its row ----
        bool ovf = false;
        for (int s = 1 + lane; s < ns; s += 64) {
            const int i = srow[s];
            const int pos = atomicAdd(&cnt[i], 1);
            if (pos >= T) ovf = true;
            sridx[s] = i * T + pos;
        }
        if (__ballot(ovf) != 0ull) CT_FAIL(1);
        __syncthreads();
        for (int s = 1 + lane; s < ns; s += 64) {
            const int r = srank[s];
            const bool stored = r >= 0;
            const int x = stored ? loff + r : -1;
            const int tprev = stored ? srow[keptslot[r - 1]] : -1;
            const int rem = stored ? nk - 1 - r : 0;
            const int nxt = (stored && r + 1 < nk) ? sridx[keptslot[r + 1]] : -1;
            unsigned long long *rp = rec + (size_t)sridx[s] * 4;
            st_agent_rec32(rp, pack2(j, x), pack2(tprev, rem), (unsigned long long)__double_as_longlong(sval[s]), pack2(nxt, 0));
        }
        drain_stores();
        // ---- announce / discharge reaches; rows whose last reach this was become ready ----
        for (int s = 1 + lane; s < ns; s += 64) {
            const int i = srow[s];
            const int r = srank[s];
            const int delta = (r >= 0 ? r - 1 : 0) - scnt[s];
            bool ready = false;
            if (delta != 0) ready = atomicAdd(&pending[i], delta) + delta == 0;
            if (ready) {
                const int q = i % nq;
                const int pos = atomicAdd(&ctrl[kCtQBase + 64 * q + 32], 1);
                st_agent_i32(&rq[(size_t)q * qcap + pos], i);
            }
        }
        __syncthreads();
    }
#undef CT_FAIL
}

__global__ void k_ict_compact(int32_t m, const int32_t *__restrict__ Loff, const int32_t *__restrict__ Llen,
                              const int32_t *__restrict__ Lptr, const int32_t *__restrict__ sidx, const double *__restrict__ sval,
                              int32_t *__restrict__ oidx, double *__restrict__ oval)
{
    const int j = blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8;
    if (j >= m) return;
    const int src = Loff[j], dst = Lptr[j], len = Llen[j];
    for (int q = threadIdx.x % 8; q < len; q += 8) { oidx[dst + q] = sidx[src + q]; oval[dst + q] = sval[src + q]; }
}

// one attempt with one capacity class; returns ILUPP_OK / an error of the reference / +1 = "outside this class"
static int icholt_attempt(hipStream_t st, const DevMat &Atri, int32_t add_fill_in, double threshold, DevMat *L, float *kernel_ms,
                          int cls)          // capacity class: 0 tiny (24 waves per CU), 1 small (12), 2 large (3)
{
    const bool small = cls <= 1;
    const int32_t m = Atri.n;
    if (m < 1) return 1;
    const long slab = (long)Atri.nnz + (long)(add_fill_in > 0 ? add_fill_in : 0) * (long)m;
    if (slab > 0x7fffffffL) return 1;
    long a = slab;                                                      // IChol.hpp:85-87
    long b = (long)((double)Atri.nnz * 10.0);
    long reserved = a < b ? a : b;
    if (reserved > 0x7fffffffL) reserved = 0x7fffffffL;

    // touch-record capacity per row: the number of reaches of a row is about the length of a pre-drop column
    const long avg = Atri.nnz / m + 1;
    int T = 16;
    const int Tlimit = cls == 3 ? (1 << 16) : kCtTmax;
    while (T < 4 * (avg + (add_fill_in > 0 ? add_fill_in : 0)) && T < Tlimit) T *= 2;
    // largest class: as many touch records per row as the matrix can need (every column may reach a row), within 8 GB
    if (cls == 3) { while (T < Tlimit && T < m && (size_t)m * T * 2 * 32 <= ((size_t)8 << 30)) T *= 2; }
    if (cls == 1 && T > kCtTsmall) return 1;
    if (cls == 0 && T > kCtTtiny) return 1;
    while (T > 16 && ((long)m * T > 0x7fffffffL || (size_t)m * T * 32 > ((size_t)96 << 30))) T /= 2;
    if ((long)m * T > 0x7fffffffL) return 1;

    int32_t *cap, *Loff, *Lidx, *Llen, *cnt, *pending, *rq, *ctrl, *Lptr;
    double *Lval;
    unsigned long long *rec;
    ILUPP_HIP(pool_malloc(&cap, sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(pool_malloc(&Loff, sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(pool_malloc(&Llen, sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(pool_malloc(&Lptr, sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(pool_malloc(&Lidx, sizeof(int32_t) * (size_t)(slab > 0 ? slab : 1)));
    ILUPP_HIP(pool_malloc(&Lval, sizeof(double) * (size_t)(slab > 0 ? slab : 1)));
    ILUPP_HIP(pool_malloc(&cnt, sizeof(int32_t) * (size_t)m));
    ILUPP_HIP(pool_malloc(&pending, sizeof(int32_t) * (size_t)m));
    const size_t rq_len = (size_t)kCtQ * (size_t)((m + kCtQ - 1) / kCtQ) + (size_t)m;   // (nq <= kCtQ queues of ceil(m / nq) entries)
    ILUPP_HIP(pool_malloc(&rq, sizeof(int32_t) * rq_len));
    const size_t ctrl_bytes = sizeof(int32_t) * (size_t)(kCtQBase + 64 * kCtQ);
    ILUPP_HIP(pool_malloc(&ctrl, ctrl_bytes));
    if (pool_malloc(&rec, (size_t)m * T * 32) != hipSuccess) {
        (void)hipGetLastError();
        for (void *q : {(void *)cap, (void *)Loff, (void *)Llen, (void *)Lptr, (void *)Lidx, (void *)Lval, (void *)cnt, (void *)pending,
                        (void *)rq, (void *)ctrl})
            ILUPP_HIP(pool_free(q));
        return 1;
    }
    ILUPP_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)m, st));
    ILUPP_HIP(hipMemsetAsync(pending, 0, sizeof(int32_t) * (size_t)m, st));
    ILUPP_HIP(hipMemsetAsync(Llen, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(rq, 0xff, sizeof(int32_t) * rq_len, st));
    ILUPP_HIP(hipMemsetAsync(ctrl, 0, ctrl_bytes, st));
    const int32_t big = 0x7fffffff;
    ILUPP_HIP(hipMemcpyAsync(ctrl + 3, &big, sizeof(int32_t), hipMemcpyHostToDevice, st));
    int waves = device_cu_count() * (cls == 0 ? 24 : (cls == 1 ? 12 : 3));
    if (waves > m) waves = m;
    // largest class: one wave per CU, working arrays in a dynamic LDS allocation
    unsigned char *gws = nullptr;
    int gNE = 0, gNS = 0, gTM = 0;
    size_t dyn_bytes = 0;
    if (cls == 3) {
        gTM = T; gNS = 2048; gNE = 1 << 18;
        dyn_bytes = ict_lds_bytes(gNS);
        waves = device_cu_count();
        if (waves > m) waves = m;
        ILUPP_HIP(pool_malloc(&gws, (ict_touch_bytes(gTM) + (size_t)gNE * 16) * (size_t)waves));
        static std::once_flag once[64];
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_icholt_df<1, 1, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        });
    }
    const int nq = waves < kCtQ ? waves : kCtQ;
    const int gb = (m + 255) / 256;
    hipLaunchKernelGGL(k_ict_prep, dim3(gb), dim3(256), 0, st, m, Atri.ptr, Atri.idx, add_fill_in, cap, pending, ctrl);
    {
        size_t tb = 0;
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cap, Loff, m, st));
        void *tmp;
        ILUPP_HIP(pool_malloc(&tmp, tb > 0 ? tb : 1));
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cap, Loff, m, st));
        ILUPP_HIP(pool_free(tmp));
    }
    hipLaunchKernelGGL(k_ict_seed, dim3(gb), dim3(256), 0, st, m, nq, pending, rq, ctrl);
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    ILUPP_HIP(hipEventRecord(e0, st));
    static const int refnans = getenv("ILUPP_REFERENCE_NANS") != nullptr ? 1 : 0;
#define ICT_ARGS m, Atri.ptr, Atri.idx, Atri.val, add_fill_in, threshold, T, nq, Loff, Lidx, Lval, Llen, cnt, rec, pending, rq, ctrl, gws, gNE, gNS, gTM, refnans
    if (cls == 0)
        hipLaunchKernelGGL((k_icholt_df<128, 64, kCtTtiny, false>), dim3(waves), dim3(64), 0, st, ICT_ARGS);
    else if (small)
        hipLaunchKernelGGL((k_icholt_df<256, 128, kCtTsmall, false>), dim3(waves), dim3(64), 0, st, ICT_ARGS);
    else if (cls == 2)
        hipLaunchKernelGGL((k_icholt_df<1024, 512, kCtTmax, false>), dim3(waves), dim3(64), 0, st, ICT_ARGS);
    else
        hipLaunchKernelGGL((k_icholt_df<1, 1, 1, true>), dim3(waves), dim3(64), dyn_bytes, st, ICT_ARGS);
#undef ICT_ARGS
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t h[4];
    ILUPP_HIP(hipMemcpyAsync(h, ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    ILUPP_HIP(hipEventDestroy(e0));
    ILUPP_HIP(hipEventDestroy(e1));
    int rc = ILUPP_OK;
    if (gws) ILUPP_HIP(pool_free(gws));
    if (h[2] == 2 && getenv("ILUPP_DEBUG")) {
        std::vector<int32_t> hp(m), hl(m + 1), hc(m);
        ILUPP_HIP(hipMemcpy(hp.data(), pending, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
        ILUPP_HIP(hipMemcpy(hl.data(), Llen, sizeof(int32_t) * (size_t)(m + 1), hipMemcpyDeviceToHost));
        ILUPP_HIP(hipMemcpy(hc.data(), cnt, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
        int unfinished = 0, firstu = -1, negp = 0;
        for (int j = 0; j < m; ++j) { if (hl[j] == 0) { ++unfinished; if (firstu < 0) firstu = j; } if (hp[j] < 0) ++negp; }
        fprintf(stderr, "[ilupp] icholt timeout: class %d, T %d, waves %d, nq %d: %d of %d columns unfinished, first %d (pending %d, touch records %d), %d negative counters\n",
                cls, T, waves, nq, unfinished, m, firstu, firstu >= 0 ? hp[firstu] : 0, firstu >= 0 ? hc[firstu] : 0, negp);
        for (int j = firstu; j >= 0 && j < m && j < firstu + 6; ++j) fprintf(stderr, "   col %d: len %d pending %d cnt %d\n", j, hl[j], hp[j], hc[j]);
    }
    if (h[3] != big) rc = ILUPP_ERR_NOT_TRIANGULAR;
    else if (h[2] == 3) rc = ILUPP_ERR_NOT_SPD;          // a column lost its diagonal: the pivot is NaN (not positive definite)
    else if (h[2] == 4) rc = ILUPP_ERR_DIAG_DROPPED;     // ... a finite pivot dropped by the threshold / the budget
    else if (h[2] == 2) rc = ILUPP_ERR_TIMEOUT;
    else if (h[2] != 0) rc = 1;
    if (rc == ILUPP_OK) {
        size_t tb = 0;
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, Llen, Lptr, m + 1, st));
        void *tmp;
        ILUPP_HIP(pool_malloc(&tmp, tb > 0 ? tb : 1));
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, Llen, Lptr, m + 1, st));
        int32_t nnz = 0;
        ILUPP_HIP(hipMemcpyAsync(&nnz, Lptr + m, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        ILUPP_HIP(pool_free(tmp));                   // (after the wait: the pool knows nothing of streams)
        if ((long)nnz > reserved) rc = ILUPP_ERR_MEMORY;             // append_row's capacity check, :3178-3179
        else {
            L->n = m; L->nnz = nnz; L->is_csr = false; L->owns = true;
            L->ptr = Lptr;
            ILUPP_HIP(pool_malloc(&L->idx, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
            ILUPP_HIP(pool_malloc(&L->val, sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
            hipLaunchKernelGGL(k_ict_compact, dim3((m + 31) / 32), dim3(256), 0, st, m, Loff, Llen, Lptr, Lidx, Lval, L->idx, L->val);
            ILUPP_HIP(hipStreamSynchronize(st));
        }
    }
    if (rc != ILUPP_OK) ILUPP_HIP(pool_free(Lptr));
    for (void *q : {(void *)cap, (void *)Loff, (void *)Llen, (void *)Lidx, (void *)Lval, (void *)cnt, (void *)pending, (void *)rq,
                    (void *)ctrl, (void *)rec})
        ILUPP_HIP(pool_free(q));
    return rc;
}

// returns ILUPP_OK / an error; the capacity classes are tried from the one with the most resident waves to the one that
// takes the whole LDS of a CU per wave; +1 when a working column does not fit even there
int icholt_factor_df(hipStream_t st, const DevMat &Atri, int32_t add_fill_in, double threshold, DevMat *L, float *kernel_ms)
{
    int rc = 1;
    const char *force = getenv("ILUPP_ICHOLT_CLASS");                  // tests: start with a given class
    const int first = force ? atoi(force) : 0;
    for (int cls = first; cls < 4 && rc == 1; ++cls) rc = icholt_attempt(st, Atri, add_fill_in, threshold, L, kernel_ms, cls);
    return rc;
}

}  // namespace ilupp
