// ilupp_amd/csrc/iluc_df.hip -- ILUC, the Crout ILU of Li / Saad / Chow (reference ILUC.hpp:112-207 with the list helpers
// :31-101 and dropping.hpp:8-34), as a DATAFLOW computation over the steps for gfx950: one wave per step k (row k of U and
// column k of L together), steps start as soon as everything that reaches them has finished, bit-identical to the reference's
// strictly sequential loop.  The design is the one of icholt_df.hip (same author, same machinery) for two coupled factors.
//
// What reaches step k: the steps h < k with a STORED l(k,h) (their U rows update z, the new row of U) and those with a
// stored u(h,k) (their L columns update w, the new column of L).  Which entries are stored depends on the values (threshold,
// top-k), so nothing is known ahead; but every entry (x,h) or (h,x), h < x, that may come to exist has a cause that is visible
// earlier: A has it, or a finished step h' stored both l(x,h') and u(h',h) (resp. l(h,h') and u(h',x)).  So every step x
// carries a counter pending[x] of announced entries in its row-left / column-upper part that are not decided yet: it starts
// at the number of such entries of A; a finishing step h with stored rows R and stored columns C adds, for every x in R, the
// number of columns in C below x, and for every x in C the number of rows in R below x (the entries its outer product
// will touch in later steps), and takes off, for every slot of its own pre-drop z and w, the number of times that slot was
// announced (once by A, once per update).  Additions first, so the counter reaches zero exactly once: when every step that
// can reach x has finished.  Then x goes to a ready queue.
//
// Order-dependent arithmetic, reproduced exactly:
//   * contributors are subtracted in the order of the reference's re-threaded linked lists listL / listU (ILUC.hpp:37-63):
//     traversal of list k = the chained columns (rows) ordered by (their previous stored index t DESCENDING, then REVERSE of
//     their order in list t, t itself last).  Step t writes the position (`seq`) of each of its contributors into the touch
//     record that contributor left at its next stored index, so step k orders its contributors by (t desc, seq desc).
//   * the column of A that initialises w is walked in the order of listA / headA (:74-101): rows ordered by (column of
//     their previous entry DESCENDING, then reverse of the order in that column's list; rows that start in this column last,
//     by descending row).  That order only depends on A's pattern: k_iluc_colorder computes it up front.
//   * working vectors: slots in insertion order (A's part, then new indices in the order the contributors' tails bring
//     them), every slot accumulated sequentially over the contributors, separate multiply / subtract; 2-norm summed in slot
//     order over the range (k, n); candidates |v| > norm * tau; top-(p-1) by magnitude with std::sort's semantics; kept
//     entries by increasing index; w / u_kk for the kept ones only.
//
// U rows and L columns land in fixed slabs of p = max_fill_in entries per step; a compaction pass produces the reference's
// arrays.  A structurally missing pivot is the reference's "zero pivot" error (smallest k reported, as the sequential loop
// would); the reservation check of append_row_with_prefix (min(p n, 10 nnz)) is made on the final lengths.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "iluc_common.h"
#include "stdsort.h"

#ifndef ILUC_W4
#define ILUC_W4 4
#endif
#ifndef ILUC_WSGB
#define ILUC_WSGB 16
#endif

namespace ilupp {

// ctrl: [2] error (1 = capacity exceeded -> next class, 2 = timeout), [3] smallest step without a pivot, [4] finished steps, then the queues
// pending[x]: entries of A left of the diagonal in row x and above it in column x
__global__ void k_iluc_prep(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t *pending,
                            int32_t *colcnt)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int left = 0;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int c = idx[q];
        if (c < r) ++left; else if (c > r) atomicAdd(&pending[c], 1);
        if (c < r) atomicAdd(&colcnt[c], 1);                       // rows below the diagonal in column c
    }
    if (left) atomicAdd(&pending[r], left);
}

// the sub-diagonal part of A by columns: for column c the CSR positions of its entries (r, c), r > c
__global__ void k_iluc_colfill(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                               const int32_t *__restrict__ colptr, int32_t *fill, int32_t *colpos)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int c = idx[q];
        if (c < r) colpos[colptr[c] + atomicAdd(&fill[c], 1)] = q;
    }
}

// position q of A (row r, some column k): does row r1 come before row r2 in the traversal of listA's chain of column k?
// (ILUC.hpp:74-101: a row enters the chain of its next column at the head, when the step of its current column re-threads
// that column's chain in traversal order; at initialisation rows enter the chains of their first columns in ascending order)
__device__ bool iluc_before(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int q1, int r1, int q2, int r2)
{
    bool flip = false;
    for (;;) {
        const int c1 = q1 > ptr[r1] ? idx[q1 - 1] : -1, c2 = q2 > ptr[r2] ? idx[q2 - 1] : -1;
        if (c1 != c2) return (c1 > c2) != flip;                   // the later arrival is nearer the head
        if (c1 < 0) return (r1 > r2) != flip;                     // both since initialisation: the larger row is nearer the head
        flip = !flip; --q1; --q2;                                 // same previous column: reverse of the order there
    }
}
__global__ void k_iluc_colorder(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                const int32_t *__restrict__ rowof, const int32_t *__restrict__ colptr, const int32_t *__restrict__ colpos,
                                int32_t *__restrict__ colord)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const int b = colptr[c], e = colptr[c + 1];
    for (int i = b; i < e; ++i) {
        const int q1 = colpos[i], r1 = rowof[q1];
        int rank = 0;
        for (int j = b; j < e; ++j) {
            if (j == i) continue;
            const int q2 = colpos[j];
            rank += iluc_before(ptr, idx, q2, rowof[q2], q1, r1) ? 1 : 0;
        }
        colord[b + rank] = q1;
    }
}
// sum over the columns of (entries below the diagonal)^2: what k_iluc_colorder's pairwise ranking costs at least
__global__ void k_iluc_colorder_cost(int32_t n, const int32_t *__restrict__ colcnt, unsigned long long *cost)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = c < n ? (unsigned long long)colcnt[c] * (unsigned long long)colcnt[c] : 0ull;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(cost, v);
}

// The order in which the reference walks column k of A (ILUC.hpp:74-101) for every k, into colord (CSR positions): the pairwise ranking of
// k_iluc_colorder where columns are short (sparse matrices: every column on its own, in parallel); where they are long -- a dense Schur
// complement of a multilevel factorisation: 549 rows took 11 s that way, the comparison of two rows walks back along everything they
// share -- the threading of the lists itself, which is sequential but linear in the entries, on the host (pattern only; such matrices are small).
// colptr must hold the exclusive scan of colcnt (both from k_iluc_prep); colpos is scratch.
int iluc_column_order(hipStream_t st, int32_t m, const DevMat &Av, const int32_t *colcnt, const int32_t *colptr, int32_t *fillc, int32_t *colpos,
                      const int32_t *rowof, int32_t *colord)
{
    const int gb = (m + 255) / 256;
    PoolBlock b_cost;
    ILUPP_HIP(b_cost.alloc(64));
    ILUPP_HIP(hipMemsetAsync(b_cost.p, 0, 64, st));
    hipLaunchKernelGGL(k_iluc_colorder_cost, dim3(gb), dim3(256), 0, st, m, colcnt, b_cost.as<unsigned long long>());
    unsigned long long cost = 0;
    ILUPP_HIP(hipMemcpyAsync(&cost, b_cost.p, sizeof(cost), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    const int64_t nnz = Av.nnz;
    // (the device kernels walk every chain in one thread: fine for the short chains of a large sparse matrix, hundreds of milliseconds for a
    //  small DENSE one -- the Schur complements of a multilevel object: 212 rows, 45 000 entries, 3 million dependent steps; there the host's
    //  O(nnz) emulation below, a millisecond, is taken)
    const bool small_dense = nnz <= 4000000 && cost > 16ull * (unsigned long long)(nnz > 0 ? nnz : 1);
    if (!small_dense && (cost <= 64ull * (unsigned long long)(nnz > 0 ? nnz : 1) + 1000000ull || nnz > 200000000)) {
        hipLaunchKernelGGL(k_iluc_colfill, dim3(gb), dim3(256), 0, st, m, Av.ptr, Av.idx, colptr, fillc, colpos);
        hipLaunchKernelGGL(k_iluc_colorder, dim3(gb), dim3(256), 0, st, m, Av.ptr, Av.idx, rowof, colptr, colpos, colord);
        return ILUPP_OK;
    }
    std::vector<int32_t> ptr((size_t)m + 1), idx((size_t)(nnz > 0 ? nnz : 1)), cp((size_t)m + 1), ord((size_t)(nnz > 0 ? nnz : 1)), first((size_t)m), list((size_t)m, -1),
                         head((size_t)m, -1), fill((size_t)m, 0);
    ILUPP_HIP(hipMemcpyAsync(ptr.data(), Av.ptr, sizeof(int32_t) * (size_t)(m + 1), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipMemcpyAsync(cp.data(), colptr, sizeof(int32_t) * (size_t)(m + 1), hipMemcpyDeviceToHost, st));
    if (nnz > 0) ILUPP_HIP(hipMemcpyAsync(idx.data(), Av.idx, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    // initialize_sparse_matrix_fields (ILUC.hpp:74-84), then for every k: the walk of chain k, and update_sparse_matrix_fields (:86-101)
    for (int32_t k = 0; k < m; ++k) {
        first[(size_t)k] = ptr[(size_t)k];
        if (ptr[(size_t)k] < ptr[(size_t)k + 1]) { const int32_t c = idx[(size_t)ptr[(size_t)k]]; list[(size_t)k] = head[(size_t)c]; head[(size_t)c] = k; }
    }
    for (int32_t k = 0; k < m; ++k) {
        for (int32_t h = head[(size_t)k]; h != -1; h = list[(size_t)h])
            if (h > k) ord[(size_t)cp[(size_t)k] + (size_t)fill[(size_t)k]++] = first[(size_t)h];
        for (int32_t h = head[(size_t)k]; h != -1; h = list[(size_t)h]) first[(size_t)h] += 1;
        int32_t h = head[(size_t)k];
        while (h != -1) {
            const int32_t i = h;
            h = list[(size_t)i];
            if (first[(size_t)i] < ptr[(size_t)i + 1]) { const int32_t c = idx[(size_t)first[(size_t)i]]; list[(size_t)i] = head[(size_t)c]; head[(size_t)c] = i; }
        }
    }
    const size_t nsub = (size_t)cp[(size_t)m];
    if (nsub) ILUPP_HIP(hipMemcpyAsync(colord, ord.data(), sizeof(int32_t) * nsub, hipMemcpyHostToDevice, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    return ILUPP_OK;
}

__global__ void k_iluc_rowof(int32_t n, const int32_t *__restrict__ ptr, int32_t *__restrict__ rowof)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) rowof[q] = r;
}

__global__ void k_iluc_seed(int32_t m, int32_t nq, const int32_t *__restrict__ pending, int32_t *rq, int32_t *ctrl)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    if (pending[j] == 0) {
        const int q = j % nq;
        const int qcap = (m + nq - 1) / nq;
        rq[(size_t)q * qcap + atomicAdd(&ctrl[kCuQBase + 64 * q + 32], 1)] = j;
    }
}


// bytes of one wave's working arrays in the global class
__host__ __device__ inline size_t iluc_ws_bytes(int ne, int ns, int tm)
{
    const size_t dbl = (size_t)ne + ns + 2 * (size_t)tm + ns + 1;
    const size_t ints = 2 * (size_t)ne + 6 * (size_t)ns + 2 * ((size_t)ns + 1) + 4 * (size_t)ns + 9 * (size_t)tm + (tm + 1) + 2 * (size_t)ns + 2 * ((size_t)ns + 1) + tm;
    return ((dbl * 8 + ints * 4) + 255) & ~(size_t)255;
}

struct IlucArgs {
    unsigned char *gws;                       // global class: the waves' working arrays
    int32_t gNE, gNS, gTM;
    int32_t n;
    const int32_t *ptr, *idx;                 // A, major-order view
    const double *val;
    const int32_t *colptr, *colord;           // sub-diagonal part by columns, CSR positions in the reference's traversal order
    const int32_t *rowof;
    int32_t p;                                // entries a stored row / column may have besides the diagonal: max_fill_in - 1
    int32_t cap;                              // slab length: p + 1
    double tau;
    int32_t T, nq;
    int32_t *Uidx, *Lidx, *Ulen, *Llen;       // slabs [n][cap]: diagonal first, then the kept entries by increasing index
    double *Uval, *Lval;
    int32_t *cntL, *cntU;                     // touch records of a step: stored l(k, .) / stored u(., k)
    unsigned long long *recL, *recU;          // [n][T][4]
    int32_t *pending, *rq, *ctrl;
};

// One step.  kNE: entries gathered per working vector, kNS: its slots, kTM: touch records of one kind.
// kGlobal: the class for everything the LDS classes cannot hold (complete factorisations, rows reached by hundreds of steps):
// the same code with its working arrays in a per-wave slice of global memory and run-time capacities.
// (kLNS is a power of two: the slot hash masks with 4 kLNS - 1)
template <int kLNE, int kLNS, int kLTM, bool kGlobal>
__global__ void __launch_bounds__(64)
k_iluc_df(IlucArgs A)
{
    __shared__ int s_erow[kLNE], s_eslot[kLNE];
    __shared__ double s_eval[kLNE];
    __shared__ int s_srow[kLNS], s_scnt[kLNS], s_srank[kLNS], s_sridx[kLNS + 1], s_cand[kLNS], s_crank[kLNS], s_keptslot[kLNS + 1];
    __shared__ double s_sval[kLNS];
    __shared__ int s_hslot[4 * kLNS];
    __shared__ int s_tx[kLTM], s_tt[kLTM], s_trem[kLTM], s_tnxt[kLTM], s_tseq[kLTM];
    __shared__ double s_tv[kLTM];
    __shared__ int s_cx[kLTM], s_crem[kLTM], s_cbase[kLTM + 1], s_cnxt[kLTM];
    __shared__ double s_cv[kLTM];
    __shared__ double tie_mag[2];
    // what the z half leaves for the end of the step: its pre-drop slots (index, times announced), its kept columns and where
    // their touch records went
    __shared__ int s_zcol[kLNS], s_zcnt[kLNS], s_zkept[kLNS + 1], s_zridx[kLNS + 1], s_znx[kLTM];
    __shared__ double s_zkv[kLNS + 1];
    const int kNE = kGlobal ? A.gNE : kLNE, kNS = kGlobal ? A.gNS : kLNS, kTM = kGlobal ? A.gTM : kLTM;
    int *erow = s_erow, *eslot = s_eslot, *srow = s_srow, *scnt = s_scnt, *srank = s_srank, *sridx = s_sridx, *cand = s_cand, *crank = s_crank,
        *keptslot = s_keptslot, *hslot = s_hslot, *tx = s_tx, *tt = s_tt, *trem = s_trem, *tnxt = s_tnxt, *tseq = s_tseq, *cx = s_cx,
        *crem = s_crem, *cbase = s_cbase, *cnxt = s_cnxt, *zcol = s_zcol, *zcnt = s_zcnt, *zkept = s_zkept, *zridx = s_zridx, *znx = s_znx;
    double *eval = s_eval, *sval = s_sval, *tv = s_tv, *cv = s_cv, *zkv = s_zkv;
    if (kGlobal) {
        double *d = reinterpret_cast<double *>(A.gws + (size_t)blockIdx.x * iluc_ws_bytes(A.gNE, A.gNS, A.gTM));
        eval = d; d += kNE; sval = d; d += kNS; tv = d; d += kTM; cv = d; d += kTM; zkv = d; d += kNS + 1;
        int *q = reinterpret_cast<int *>(d);
        erow = q; q += kNE; eslot = q; q += kNE;
        srow = q; q += kNS; scnt = q; q += kNS; srank = q; q += kNS; sridx = q; q += kNS + 1; cand = q; q += kNS; crank = q; q += kNS;
        keptslot = q; q += kNS + 1; hslot = q; q += 4 * kNS;
        tx = q; q += kTM; tt = q; q += kTM; trem = q; q += kTM; tnxt = q; q += kTM; tseq = q; q += kTM;
        cx = q; q += kTM; crem = q; q += kTM; cbase = q; q += kTM + 1; cnxt = q; q += kTM;
        zcol = q; q += kNS; zcnt = q; q += kNS; zkept = q; q += kNS + 1; zridx = q; q += kNS + 1; znx = q; q += kTM;
    }

    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int m = A.n, T = A.T, nq = A.nq;
#define CU_FAIL(code) do { if (lane == 0) atomicCAS(&A.ctrl[2], 0, (code)); return; } while (0)
    // (one wave per block: __syncthreads() orders LDS; with the arrays in global memory the wave's stores must have landed too)
#define CU_SYNC() do { if (kGlobal) __builtin_amdgcn_s_waitcnt(0); __syncthreads(); } while (0)

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
                // the limit is on the time WITHOUT PROGRESS anywhere (ctrl[4] counts finished steps): a long dependency chain of
                // expensive steps (a complete factorisation in the global class) makes a late step wait for seconds
                const int done = ld_agent_i32(&A.ctrl[4]);
                if (done != seen) { seen = done; spins = 0; }
                if (spins > kCuSpinLimit) CU_FAIL(2);
            }
            __builtin_amdgcn_s_sleep(4);
        }
        k = __builtin_amdgcn_readfirstlane(k);

        int nzs = 0, nkz = 0, nzt = 0;  // z: pre-drop slots, kept entries (besides the diagonal), contributors
        double ukk = 0.0;
        bool nopivot = false;
        // ================================ the two halves: z (row k of U), then w (column k of L) ================================
        for (int half = 0; half < 2; ++half) {
            const bool Z = half == 0;
            // ---- A's part, in the reference's insertion order ----
            int na = 0;
            if (Z) {
                const int r0 = __builtin_amdgcn_readfirstlane(A.ptr[k]), r1 = __builtin_amdgcn_readfirstlane(A.ptr[k + 1]);
                int first = r1;                                        // first entry with column >= k (firstA[k], ILUC.hpp:143-145)
                for (int base = r0; base < r1; base += 64) {
                    const int q = base + lane;
                    const unsigned long long mk = __ballot(q < r1 && A.idx[q] >= k);
                    if (mk) { first = base + __ffsll((long long)mk) - 1; break; }
                }
                first = __builtin_amdgcn_readfirstlane(first);
                na = r1 - first;
                if (na > kNS) CU_FAIL(11);
                for (int e = lane; e < na; e += 64) { erow[e] = A.idx[first + e]; eval[e] = A.val[first + e]; eslot[e] = e; srow[e] = erow[e]; }
            } else {
                const int b = __builtin_amdgcn_readfirstlane(A.colptr[k]);
                na = __builtin_amdgcn_readfirstlane(A.colptr[k + 1]) - b;
                if (na > kNS) CU_FAIL(11);
                for (int e = lane; e < na; e += 64) { const int q = A.colord[b + e]; erow[e] = A.rowof[q]; eval[e] = A.val[q]; eslot[e] = e; srow[e] = erow[e]; }
            }
            // ---- the touch records of this kind ----
            const int nt = __builtin_amdgcn_readfirstlane(ld_agent_i32(Z ? &A.cntL[k] : &A.cntU[k]));
            if (nt > T || nt > kTM) CU_FAIL(12);
            const unsigned long long *recs = (Z ? A.recL : A.recU) + (size_t)k * T * 4;
            for (int q = lane; q < nt; q += 64) {
                const Rec32 rr = ld_agent_rec32(recs + (size_t)q * 4);
                const unsigned long long w0 = rr.w[0], w1 = rr.w[1], w2 = rr.w[2], w3 = rr.w[3];
                tx[q] = (int)(unsigned)(w0 >> 32);
                tt[q] = (int)(unsigned)w1; trem[q] = (int)(unsigned)(w1 >> 32);
                tv[q] = __longlong_as_double((long long)w2);
                tnxt[q] = (int)(unsigned)w3; tseq[q] = (int)(unsigned)(w3 >> 32);
            }
            CU_SYNC();
            // ---- contributors in the reference's linked-list order: (previous stored index desc, seq desc) ----
            for (int q = lane; q < nt; q += 64) {
                const int t0 = tt[q], s0 = tseq[q];
                int p = 0;
                for (int q2 = 0; q2 < nt; ++q2) { const int t2 = tt[q2], s2 = tseq[q2]; p += (t2 > t0 || (t2 == t0 && s2 > s0)) ? 1 : 0; }
                cx[p] = tx[q]; crem[p] = trem[q]; cv[p] = tv[q]; cnxt[p] = tnxt[q];
            }
            CU_SYNC();
            {
                unsigned long long *rb = Z ? A.recL : A.recU;
                for (int p = lane; p < nt; p += 64)
                    if (cnxt[p] >= 0) st_agent_i32(reinterpret_cast<int *>(rb + (size_t)cnxt[p] * 4 + 3) + 1, p + 1);
            }
            if (lane == 0) {
                int s = 0;
                for (int p = 0; p < nt; ++p) { cbase[p] = s; s += crem[p]; }
                cbase[nt] = s;
            }
            CU_SYNC();
            const int net = __builtin_amdgcn_readfirstlane(cbase[nt]);
            const int ne = na + net;
            if (ne > kNE) CU_FAIL(13);
            // ---- tails: z takes u(h, j >= k) * l(k,h) from the contributor's U row, w takes l(i > k, h) * u(h,k) from its L column ----
            {
                const int32_t *Oidx = Z ? A.Uidx : A.Lidx;
                const double *Oval = Z ? A.Uval : A.Lval;
                for (int e = lane; e < net; e += 64) {
                    int p = 0;
                    while (cbase[p + 1] <= e) ++p;
                    const int xx = cx[p] + (e - cbase[p]);
                    erow[na + e] = ld_agent_i32(&Oidx[xx]);
                    eval[na + e] = cv[p] * ld_agent_f64(&Oval[xx]);                  // L_kh * U.data[j]  /  U_hk * L.data[j]  (:154, :170)
                }
            }
            CU_SYNC();
            // ---- slots in insertion order (hash: index -> slot + 1) ----
            const unsigned hmask = (unsigned)(4 * kNS - 1);
            for (int h = lane; h < 4 * kNS; h += 64) hslot[h] = 0;
            CU_SYNC();
            for (int e = lane; e < na; e += 64) {
                unsigned h = ((unsigned)erow[e] * 0x9E3779B1u) >> 7;
                for (;;) { h &= hmask; if (atomicCAS(&hslot[h], 0, e + 1) == 0) break; ++h; }
            }
            CU_SYNC();
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
                bool need = act && sl < 0;
                unsigned long long todo = __ballot(need);
                while (todo != 0ull) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int lr = __shfl(r, leader);
                    const bool same = need && r == lr;
                    if (same) { sl = ns; need = false; }
                    if (lane == leader && ns < kNS) {
                        srow[ns] = lr;
                        unsigned h = ((unsigned)lr * 0x9E3779B1u) >> 7;
                        for (;;) { h &= hmask; if (hslot[h] == 0) { hslot[h] = ns + 1; break; } ++h; }
                    }
                    ++ns;
                    todo &= ~__ballot(same);
                    CU_SYNC();
                }
                if (act) eslot[e] = sl;
                if (ns > kNS) break;
            }
            ns = __builtin_amdgcn_readfirstlane(ns);
            if (ns > kNS) CU_FAIL(14);
            CU_SYNC();
            // ---- accumulate every slot sequentially over the entries (batches of 64 in order, inside a batch lowest lane first) ----
            for (int sl = lane; sl < ns; sl += 64) {
                sval[sl] = sl < na ? eval[sl] : 0.0;
                scnt[sl] = sl < na ? 1 : 0; srank[sl] = -1; crank[sl] = 64;
            }
            CU_SYNC();
            for (int base = na; base < ne; base += 64) {
                const int e = base + lane;
                bool rem = e < ne;
                const int sl = rem ? eslot[e] : 0;
                const double ev = rem ? eval[e] : 0.0;
                while (__ballot(rem) != 0ull) {
                    if (rem) atomicMin(&crank[sl], lane);
                    CU_SYNC();
                    const bool go = rem && crank[sl] == lane;
                    if (go) { sval[sl] = sval[sl] - ev; scnt[sl] += 1; }
                    CU_SYNC();
                    if (go) { crank[sl] = 64; rem = false; }
                    CU_SYNC();
                }
            }
            // ---- the pivot (z only): its slot is A's first entry or was created by an update; absent = the reference's error ----
            int dslot = -1;
            if (Z) {
                for (int base = 0; base < ns; base += 64) {
                    const int s = base + lane;
                    const unsigned long long mk = __ballot(s < ns && srow[s] == k);
                    if (mk) { dslot = base + __ffsll((long long)mk) - 1; break; }
                }
                dslot = __builtin_amdgcn_readfirstlane(dslot);
                if (dslot < 0) { nopivot = true; if (lane == 0) atomicMin(&A.ctrl[3], k); }
                else ukk = sval[dslot];
                // (the reference stops at the first such step; here the others go on -- only the smallest k is reported, and
                // nothing below it depends on this one -- with a stand-in pivot)
                if (nopivot) ukk = 1.0;
            }
            // ---- threshold_and_drop(v, list, p, tau, k+1, n)  (dropping.hpp:8-34): the range excludes the pivot ----
            const int budget = A.p;
            int nk = 0;
            if (budget > 0) {
                double zz = 0.0;
                for (int s = 0; s < ns; ++s) { if (s == dslot) continue; const double v = sval[s]; const double sq = v * v; zz = zz + sq; }
                const double thr = sqrt(zz) * A.tau;
                int ncand = 0;
                for (int base = 0; base < ns; base += 64) {
                    const int s = base + lane;
                    const bool is = s < ns && s != dslot && fabs(sval[s]) > thr;
                    const unsigned long long mask = __ballot(is);
                    if (is) cand[ncand + __popcll(mask & lt_mask)] = s;
                    ncand += __popcll(mask);
                }
                ncand = __builtin_amdgcn_readfirstlane(ncand);
                CU_SYNC();
                nk = ncand;
                if (ncand > budget) {
                    nk = budget;
                    for (int c = lane; c < ncand; c += 64) {
                        const double a = fabs(sval[cand[c]]);
                        int r = 0;
                        for (int c2 = 0; c2 < ncand; ++c2) { const double a2 = fabs(sval[cand[c2]]); r += (a2 > a || (a2 == a && c2 < c)) ? 1 : 0; }
                        crank[c] = r;
                        if (r == budget - 1) tie_mag[0] = a;
                        if (r == budget) tie_mag[1] = a;
                    }
                    CU_SYNC();
                    const bool tie = __builtin_amdgcn_readfirstlane((ncand > 16 && tie_mag[0] == tie_mag[1]) ? 1 : 0) != 0;
                    if (tie) {
                        // equal magnitudes across the cut: the kept set is whatever libstdc++'s introsort leaves in front
                        if (lane == 0) {
                            c_sort_slots_by_abs_desc(cand, ncand, sval);
                            for (int c = 0; c < budget; ++c) srank[cand[c]] = 0;
                        }
                    } else {
                        for (int c = lane; c < ncand; c += 64) if (crank[c] < budget) srank[cand[c]] = 0;
                    }
                } else {
                    for (int c = lane; c < ncand; c += 64) srank[cand[c]] = 0;
                }
            }
            CU_SYNC();
            // kept entries by increasing index; srank = position behind the diagonal (1-based) in the stored row / column
            for (int s = lane; s < ns; s += 64) {
                if (srank[s] < 0) continue;
                const int r0 = srow[s];
                int r = 0;
                for (int s2 = 0; s2 < ns; ++s2) r += (srank[s2] >= 0 && srow[s2] < r0) ? 1 : 0;
                crank[s] = r + 1;
                keptslot[r + 1] = s;
            }
            CU_SYNC();
            for (int s = lane; s < ns; s += 64) if (srank[s] >= 0) srank[s] = crank[s];
            CU_SYNC();
            if (Z) {
                // ---- row k of U: the pivot, then the kept entries (append_row_with_prefix, sparse_implementation.h:3189-3208) ----
                const size_t off = (size_t)k * A.cap;
                // (also for a step without a pivot, with its stand-in: the steps it reaches go on -- only the error is reported in the end -- and
                // must read a row that was written.  They used to read whatever the slab held: zeros in fresh memory, the indices of an older
                // factorisation in a block the pool handed out again -- the memory fault behind round 2's abort in the error-path tests)
                if (lane == 0) { st_agent_i32(&A.Uidx[off], k); st_agent_f64(&A.Uval[off], ukk); }
                for (int s = lane; s < ns; s += 64)
                    if (srank[s] >= 0) { st_agent_i32(&A.Uidx[off + srank[s]], srow[s]); st_agent_f64(&A.Uval[off + srank[s]], sval[s]); }
                if (lane == 0) A.Ulen[k] = nk + 1;
                nkz = nk; nzs = ns;
                for (int s = lane; s < ns; s += 64) { zcol[s] = srow[s]; zcnt[s] = s == dslot ? 0 : scnt[s]; }
                for (int r = 1 + lane; r <= nk; r += 64) { zkept[r] = srow[keptslot[r]]; zkv[r] = sval[keptslot[r]]; }
                nzt = nt;
                for (int q = lane; q < nt; q += 64) znx[q] = cnxt[q];
                CU_SYNC();
            } else {
                // ---- column k of L: 1, then the kept entries over the pivot (:189-193) ----
                const size_t off = (size_t)k * A.cap;
                for (int s = lane; s < ns; s += 64) if (srank[s] >= 0) sval[s] = sval[s] / ukk;
                CU_SYNC();
                if (lane == 0) { st_agent_i32(&A.Lidx[off], k); st_agent_f64(&A.Lval[off], 1.0); }
                for (int s = lane; s < ns; s += 64)
                    if (srank[s] >= 0) { st_agent_i32(&A.Lidx[off + srank[s]], srow[s]); st_agent_f64(&A.Lval[off + srank[s]], sval[s]); }
                if (lane == 0) A.Llen[k] = nk + 1;
                // ========================= end of the step: touch records, then the counters =========================
                const int nkw = nk, nws = ns;
                // record slots: stored rows i of L in the lists of their steps (kind L), stored columns j of U (kind U)
                bool ovf = false;
                for (int r = 1 + lane; r <= nkw; r += 64) {
                    const int i = srow[keptslot[r]];
                    const int pos = atomicAdd(&A.cntL[i], 1);
                    if (pos >= T) ovf = true;
                    sridx[r] = i * T + pos;
                }
                for (int r = 1 + lane; r <= nkz; r += 64) {
                    const int j = zkept[r];
                    const int pos = atomicAdd(&A.cntU[j], 1);
                    if (pos >= T) ovf = true;
                    zridx[r] = j * T + pos;
                }
                if (__ballot(ovf) != 0ull) CU_FAIL(15);
                CU_SYNC();
                // kind L, stored row i at position r: the step of row i will take the tail u(k, j >= i) of THIS step's U row
                for (int r = 1 + lane; r <= nkw; r += 64) {
                    const int s = keptslot[r];
                    const int i = srow[s];
                    int lb = 1;
                    while (lb <= nkz && zkept[lb] < i) ++lb;                             // first kept column >= i
                    const int tprev = r > 1 ? srow[keptslot[r - 1]] : k;
                    const int nxt = r < nkw ? sridx[r + 1] : -1;
                    unsigned long long *rp = A.recL + (size_t)sridx[r] * 4;
                    st_agent_rec32(rp, cu_pack2(k, (int)(off + lb)), cu_pack2(tprev, nkz + 1 - lb), (unsigned long long)__double_as_longlong(sval[s]),
                                   cu_pack2(nxt, 0));
                }
                // kind U, stored column j at position r: the step of column j will take the tail l(i > j, k) of THIS step's L column
                for (int r = 1 + lane; r <= nkz; r += 64) {
                    const int j = zkept[r];
                    int ub = 1;
                    while (ub <= nkw && srow[keptslot[ub]] <= j) ++ub;                   // first kept row > j
                    const int tprev = r > 1 ? zkept[r - 1] : k;
                    const int nxt = r < nkz ? zridx[r + 1] : -1;
                    unsigned long long *rp = A.recU + (size_t)zridx[r] * 4;
                    st_agent_rec32(rp, cu_pack2(k, (int)(off + ub)), cu_pack2(tprev, nkw + 1 - ub), (unsigned long long)__double_as_longlong(zkv[r]),
                                   cu_pack2(nxt, 0));
                }
                drain_stores();
                // announcements first: x in R gets the kept columns below x, x in C the kept rows below x
                for (int r = 1 + lane; r <= nkw; r += 64) {
                    const int i = srow[keptslot[r]];
                    int c = 0;
                    while (c < nkz && zkept[c + 1] < i) ++c;
                    if (c) atomicAdd(&A.pending[i], c);
                }
                for (int r = 1 + lane; r <= nkz; r += 64) {
                    const int j = zkept[r];
                    int c = 0;
                    while (c < nkw && srow[keptslot[c + 1]] < j) ++c;
                    if (c) atomicAdd(&A.pending[j], c);
                }
                // ... and every stored row (column) but the first waits for the step of the one before it in this column (row):
                // that step hands the chain on and tells the record where in ITS list the chain stood (seq) -- the order of
                // the lists is part of the arithmetic
                for (int r = 2 + lane; r <= nkw; r += 64) atomicAdd(&A.pending[srow[keptslot[r]]], 1);
                for (int r = 2 + lane; r <= nkz; r += 64) atomicAdd(&A.pending[zkept[r]], 1);
                __builtin_amdgcn_s_waitcnt(0);
                CU_SYNC();
                // the chains this step handed on (seq written above, drained): their next steps may go
                for (int half2 = 0; half2 < 2; ++half2) {
                    const int cn = half2 == 0 ? nt : nzt;
                    for (int q = lane; q < cn; q += 64) {
                        const int nx = half2 == 0 ? cnxt[q] : znx[q];
                        if (nx < 0) continue;
                        const int x = nx / T;
                        if (atomicAdd(&A.pending[x], -1) - 1 == 0) {
                            const int qq = x % nq;
                            const int pos = atomicAdd(&A.ctrl[kCuQBase + 64 * qq + 32], 1);
                            st_agent_i32(&A.rq[(size_t)qq * qcap + pos], x);
                        }
                    }
                }
                // then what this step decided: every pre-drop slot of w and of z, as many times as it was announced
                for (int half2 = 0; half2 < 2; ++half2) {
                    const int cntn = half2 == 0 ? nws : nzs;
                    for (int s = lane; s < cntn; s += 64) {
                        const int x = half2 == 0 ? srow[s] : zcol[s];
                        const int d = half2 == 0 ? scnt[s] : zcnt[s];
                        if (d == 0) continue;
                        if (atomicAdd(&A.pending[x], -d) - d == 0) {
                            const int q = x % nq;
                            const int pos = atomicAdd(&A.ctrl[kCuQBase + 64 * q + 32], 1);
                            st_agent_i32(&A.rq[(size_t)q * qcap + pos], x);
                        }
                    }
                }
                // progress mark for the waiters' time-out (not from every step: one address that a million steps update is a bottleneck of its own)
                if (lane == 0 && ((k & 255) == 0 || ne > 2048)) atomicAdd(&A.ctrl[4], 1);
                CU_SYNC();
            }
        }
    }
#undef CU_FAIL
#undef CU_SYNC
}

__global__ void k_iluc_compact(int32_t m, int32_t cap, const int32_t *__restrict__ len, const int32_t *__restrict__ optr,
                               const int32_t *__restrict__ sidx, const double *__restrict__ sval, int32_t *__restrict__ oidx,
                               double *__restrict__ oval)
{
    const int j = blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8;
    if (j >= m) return;
    const size_t src = (size_t)j * cap;
    const int dst = optr[j], n = len[j];
    for (int q = threadIdx.x % 8; q < n; q += 8) { oidx[dst + q] = sidx[src + q]; oval[dst + q] = sval[src + q]; }
}

// one attempt with one capacity class; ILUPP_OK / an error of the reference / +1 = "outside this class"
static int iluc_attempt(hipStream_t st, const DevMat &Av, int32_t max_fill_in, double threshold, DevMat *L, DevMat *U, int32_t *err_row,
                        float *kernel_ms, int cls, int *which_capacity)
{
    const int32_t m = Av.n;
    if (m < 1) return 1;
    const long reserved_l = [&] {                                     // ILUC.hpp:120 (with the caller's max_fill_in)
        const long a = (long)max_fill_in * (long)m, b = (long)(10.0 * (double)Av.nnz);
        const long r = a < b ? a : b;
        return r > 0 ? r : 0;
    }();
    int32_t fill = max_fill_in < 1 ? 1 : max_fill_in;               // :133
    if (fill > m) fill = m;
    const int cap = fill;                                             // diagonal + (fill - 1) kept entries
    if ((long)m * cap > 0x7fffffffL) return 1;
    const long avg = Av.nnz / m + 1;
    int T = 16;
    const int Tlimit = cls == 0 ? 32 : (cls == 1 ? 64 : (cls == 2 ? 128 : (cls == 3 ? 512 : 4096)));
    (void)avg;
    T = Tlimit < 4096 ? Tlimit : 16;           // as many touch records per step as the class can read (a row of L may be reached by many)
    if (cls == 4) { while (T < Tlimit && T < m) T *= 2; }                  // (a row of L can be reached by every earlier step)
    while (T > 16 && ((long)m * T > 0x7fffffffL || (size_t)m * T * 64 > ((size_t)64 << 30))) T /= 2;
    if ((long)m * T > 0x7fffffffL || (size_t)m * T * 64 > ((size_t)64 << 30)) return 1;

    // (work arrays in holders that give them back when the scope ends: an ILUPP_HIP that throws in the middle must not leak them)
    PoolBlock b_pending, b_colcnt, b_colptr, b_fillc, b_colpos, b_colord, b_rowof, b_Uidx, b_Lidx, b_Uval, b_Lval, b_Ulen, b_Llen, b_cntL,
              b_cntU, b_recL, b_recU, b_rq, b_ctrl, b_gws;
    const size_t slab = (size_t)m * cap;
    ILUPP_HIP(b_pending.alloc(sizeof(int32_t) * (size_t)m));
    ILUPP_HIP(b_colcnt.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_colptr.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_fillc.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_colpos.alloc(sizeof(int32_t) * (size_t)(Av.nnz > 0 ? Av.nnz : 1)));
    ILUPP_HIP(b_colord.alloc(sizeof(int32_t) * (size_t)(Av.nnz > 0 ? Av.nnz : 1)));
    ILUPP_HIP(b_rowof.alloc(sizeof(int32_t) * (size_t)(Av.nnz > 0 ? Av.nnz : 1)));
    ILUPP_HIP(b_Uidx.alloc(sizeof(int32_t) * slab));
    ILUPP_HIP(b_Lidx.alloc(sizeof(int32_t) * slab));
    ILUPP_HIP(b_Uval.alloc(sizeof(double) * slab));
    ILUPP_HIP(b_Lval.alloc(sizeof(double) * slab));
    ILUPP_HIP(b_Ulen.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_Llen.alloc(sizeof(int32_t) * (size_t)(m + 1)));
    ILUPP_HIP(b_cntL.alloc(sizeof(int32_t) * (size_t)m));
    ILUPP_HIP(b_cntU.alloc(sizeof(int32_t) * (size_t)m));
    ILUPP_HIP(b_recL.alloc((size_t)m * T * 32));
    ILUPP_HIP(b_recU.alloc((size_t)m * T * 32));
    const size_t rq_len = (size_t)kCuQ * (size_t)((m + kCuQ - 1) / kCuQ) + (size_t)m;
    ILUPP_HIP(b_rq.alloc(sizeof(int32_t) * rq_len));
    const size_t ctrl_bytes = sizeof(int32_t) * (size_t)(kCuQBase + 64 * kCuQ);
    ILUPP_HIP(b_ctrl.alloc(ctrl_bytes));
    int32_t *pending = b_pending.as<int32_t>(), *colcnt = b_colcnt.as<int32_t>(), *colptr = b_colptr.as<int32_t>(), *fillc = b_fillc.as<int32_t>(),
            *colpos = b_colpos.as<int32_t>(), *colord = b_colord.as<int32_t>(), *rowof = b_rowof.as<int32_t>(), *Uidx = b_Uidx.as<int32_t>(),
            *Lidx = b_Lidx.as<int32_t>(), *Ulen = b_Ulen.as<int32_t>(), *Llen = b_Llen.as<int32_t>(), *cntL = b_cntL.as<int32_t>(),
            *cntU = b_cntU.as<int32_t>(), *rq = b_rq.as<int32_t>(), *ctrl = b_ctrl.as<int32_t>();
    double *Uval = b_Uval.as<double>(), *Lval = b_Lval.as<double>();
    unsigned long long *recL = b_recL.as<unsigned long long>(), *recU = b_recU.as<unsigned long long>();
    ILUPP_HIP(hipMemsetAsync(pending, 0, sizeof(int32_t) * (size_t)m, st));
    ILUPP_HIP(hipMemsetAsync(colcnt, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(fillc, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(cntL, 0, sizeof(int32_t) * (size_t)m, st));
    ILUPP_HIP(hipMemsetAsync(cntU, 0, sizeof(int32_t) * (size_t)m, st));
    ILUPP_HIP(hipMemsetAsync(Ulen, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(Llen, 0, sizeof(int32_t) * (size_t)(m + 1), st));
    ILUPP_HIP(hipMemsetAsync(rq, 0xff, sizeof(int32_t) * rq_len, st));
    ILUPP_HIP(hipMemsetAsync(ctrl, 0, ctrl_bytes, st));
    const int32_t big = 0x7fffffff;
    ILUPP_HIP(hipMemcpyAsync(ctrl + 3, &big, sizeof(int32_t), hipMemcpyHostToDevice, st));
    const int gb = (m + 255) / 256;
    hipLaunchKernelGGL(k_iluc_prep, dim3(gb), dim3(256), 0, st, m, Av.ptr, Av.idx, pending, colcnt);
    hipLaunchKernelGGL(k_iluc_rowof, dim3(gb), dim3(256), 0, st, m, Av.ptr, rowof);
    {
        size_t tb = 0;
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, colcnt, colptr, m + 1, st));
        PoolBlock tmp;
        ILUPP_HIP(tmp.alloc(tb > 0 ? tb : 1));
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, colcnt, colptr, m + 1, st));
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    { const int rc = iluc_column_order(st, m, Av, colcnt, colptr, fillc, colpos, rowof, colord); if (rc) return rc; }
    int waves = device_cu_count() * (cls == 0 ? 16 : (cls == 1 ? 8 : (cls == 2 ? 2 : (cls == 3 ? 2 : ILUC_W4))));
    if (waves > m) waves = m;
    unsigned char *gws = nullptr;
    int gNE = 0, gNS = 0, gTM = 0;
    if (cls == 4) {
        gTM = T; gNS = 1024; while (gNS < m + 1 && gNS < 16384) gNS *= 2;       // (a power of two: the slot hash masks with 4 gNS - 1)
        gNE = 1 << 19;
        while (gNE > 4096 && (size_t)waves * iluc_ws_bytes(gNE, gNS, gTM) > ((size_t)ILUC_WSGB << 30)) gNE /= 2;
        ILUPP_HIP(b_gws.alloc((size_t)waves * iluc_ws_bytes(gNE, gNS, gTM)));
        gws = b_gws.as<unsigned char>();
    }
    const int nq = waves < kCuQ ? waves : kCuQ;
    hipLaunchKernelGGL(k_iluc_seed, dim3(gb), dim3(256), 0, st, m, nq, pending, rq, ctrl);
    IlucArgs a;
    a.gws = gws; a.gNE = gNE; a.gNS = gNS; a.gTM = gTM;
    a.n = m; a.ptr = Av.ptr; a.idx = Av.idx; a.val = Av.val; a.colptr = colptr; a.colord = colord; a.rowof = rowof;
    a.p = fill - 1; a.cap = cap; a.tau = threshold; a.T = T; a.nq = nq;
    a.Uidx = Uidx; a.Lidx = Lidx; a.Ulen = Ulen; a.Llen = Llen; a.Uval = Uval; a.Lval = Lval;
    a.cntL = cntL; a.cntU = cntU; a.recL = recL; a.recU = recU; a.pending = pending; a.rq = rq; a.ctrl = ctrl;
    struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;
    ILUPP_HIP(hipEventCreate(&ev.a));
    ILUPP_HIP(hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    ILUPP_HIP(hipEventRecord(e0, st));
    if (cls == 0) hipLaunchKernelGGL((k_iluc_df<256, 128, 32, false>), dim3(waves), dim3(64), 0, st, a);
    else if (cls == 1) hipLaunchKernelGGL((k_iluc_df<768, 256, 64, false>), dim3(waves), dim3(64), 0, st, a);
    else if (cls == 2) hipLaunchKernelGGL((k_iluc_df<1024, 512, 128, false>), dim3(waves), dim3(64), 0, st, a);
    else if (cls == 3) hipLaunchKernelGGL((k_iluc_df<960, 256, 512, false>), dim3(waves), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_iluc_df<1, 1, 1, true>), dim3(waves), dim3(64), 0, st, a);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t h[4];
    ILUPP_HIP(hipMemcpyAsync(h, ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    float ms = 0.f;
    ILUPP_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (getenv("ILUPP_DEBUG")) { fprintf(stderr, "[ilupp] iluc: class %d, T %d, waves %d: status %d, %.1f ms\n", cls, T, waves, h[2], ms); }
    int rc = ILUPP_OK;
    if (h[2] == 2) rc = ILUPP_ERR_TIMEOUT;
    else if (h[2] != 0) { rc = 1; if (which_capacity) *which_capacity = h[2]; }      // 11..15: which capacity (A's part, touch records read, entries, slots, touch records written)
    else if (h[3] != big) {
        // the reference's loop fails at the FIRST step that fails: a step k checks its pivot (ILUC.hpp:174-175), then appends the row of
        // U and the column of L (their reservation checks, sparse_implementation.h:3196-3197).  So the zero pivot of step h[3] is
        // what it reports -- unless the entries of an earlier step already exceeded the reservation
        rc = ILUPP_ERR_ZERO_PIVOT; if (err_row) *err_row = h[3];
        for (int d = 0; d < 2 && h[3] > 0; ++d) {
            PoolBlock pre, tmp;
            ILUPP_HIP(pre.alloc(sizeof(int32_t) * (size_t)(m + 1)));
            size_t tb = 0;
            ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d == 0 ? Ulen : Llen, pre.as<int32_t>(), m + 1, st));
            ILUPP_HIP(tmp.alloc(tb > 0 ? tb : 1));
            ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, d == 0 ? Ulen : Llen, pre.as<int32_t>(), m + 1, st));
            int32_t upto = 0;
            ILUPP_HIP(hipMemcpyAsync(&upto, pre.as<int32_t>() + h[3], sizeof(int32_t), hipMemcpyDeviceToHost, st));
            ILUPP_HIP(hipStreamSynchronize(st));
            if ((long)upto > reserved_l) { rc = ILUPP_ERR_MEMORY; break; }
        }
    }
    if (rc == ILUPP_OK) {
        DevMat *out[2] = {L, U};
        int32_t *lens[2] = {Llen, Ulen}, *sidx[2] = {Lidx, Uidx};
        double *svals[2] = {Lval, Uval};
        for (int d = 0; d < 2 && rc == ILUPP_OK; ++d) {
            DevMat *M = out[d];
            ILUPP_HIP(pool_malloc(&M->ptr, sizeof(int32_t) * (size_t)(m + 1)));
            size_t tb = 0;
            ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, lens[d], M->ptr, m + 1, st));
            PoolBlock tmp;
            ILUPP_HIP(tmp.alloc(tb > 0 ? tb : 1));
            ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, lens[d], M->ptr, m + 1, st));
            int32_t nnz = 0;
            ILUPP_HIP(hipMemcpyAsync(&nnz, M->ptr + m, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            ILUPP_HIP(hipStreamSynchronize(st));
            if ((long)nnz > reserved_l) { rc = ILUPP_ERR_MEMORY; break; }     // append_row_with_prefix's check, :3196-3197
            M->n = m; M->nnz = nnz; M->is_csr = true; M->owns = true;
            ILUPP_HIP(pool_malloc(&M->idx, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
            ILUPP_HIP(pool_malloc(&M->val, sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
            hipLaunchKernelGGL(k_iluc_compact, dim3((m + 31) / 32), dim3(256), 0, st, m, cap, lens[d], M->ptr, sidx[d], svals[d], M->idx, M->val);
        }
        ILUPP_HIP(hipStreamSynchronize(st));
        if (rc != ILUPP_OK) { L->release(); U->release(); }
    }
    // (the holders give the work arrays back)
    return rc;
}

// Av: the major-order view of the input (CSR arrays of A, or of A^T for COLUMN input).  L: the unit lower factor by columns
// (arrays = CSR of L^T: the 1 first, rows ascending), U: the upper factor by rows (pivot first).
int iluc_factor(hipStream_t st, const DevMat &Av, int32_t max_fill_in, double threshold, DevMat *L, DevMat *U, int32_t *err_row,
                float *kernel_ms)
{
    int rc = 1;
    // where to start: a working row gathers about (entries of A's row + fill) tails of up to fill entries; an attempt in a class
    // that turns out too small costs the time until the first step that does not fit
    const long fill = max_fill_in < 1 ? 1 : max_fill_in;
    const long est = (Av.nnz / (Av.n > 0 ? Av.n : 1) / 2 + 1 + fill) * fill;
#ifdef ILUC_FIRST
    const int first = ILUC_FIRST;
#else
    const int first = est <= 192 ? 0 : (est <= 640 ? 1 : 2);
#endif
    for (int cls = first; cls < 5 && rc == 1; ++cls) {
        int which = 0;
        rc = iluc_attempt(st, Av, max_fill_in, threshold, L, U, err_row, kernel_ms, cls, &which);
        // a step reached by more stored entries than the class has touch records: the next class has only twice as many, the
        // one after it 512 (a matrix with a few heavily reached rows would fail there again, after a longer run)
        if (rc == 1 && (which == 12 || which == 15) && cls < 2) cls = 2;
    }
    if (rc == 1) { set_error("ILUC: a working row does not fit the largest capacity class"); rc = ILUPP_ERR_UNSUPPORTED; }
    return rc;
}

}  // namespace ilupp
