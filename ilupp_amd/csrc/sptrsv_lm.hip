// ilupp_amd/csrc/sptrsv_lm.hip -- level-major packed triangular sweeps for short-row factors (gfx950).
//
// Same arithmetic as sptrsv.hip (reference matrix_sparse::triangular_solve, sparse_implementation.h:4040-4087:
// sequential accumulation in stored order, separate multiply and subtract, division by the diagonal found
// by position), different data layout.  sptrsv.hip streams CSR: every lane follows its own rows, i.e.
// five private address streams per lane; on MI355X that access pattern tops out near 1.5-2 TB/s because the
// 128-byte lines of 65536 x 5 concurrent streams do not survive in the 4 MB L2 of an XCD between two touches.
// Here the factor is re-packed ONCE per factorisation into the order in which the sweep consumes it:
//
//   * a workgroup's lanes keep their rows (Schedule: lane = chain of consecutive rows), but every lane t gets
//     a skew s(t) such that row k of lane t is due at step tau = k + s(t) (its dependency level inside the
//     workgroup: tau(dep) <= tau(row) is verified for every entry when the records are built);
//   * for every wave and every step there is one 3 KB chunk  [64 x {d0,d1,d2,flags}][64 x {v0,v1}][64 x {v2,vdiag}]
//     holding, lane-interleaved, the row each of the 64 lanes solves at that step: dependencies already in
//     accumulation order (so one kernel serves the three sweep kinds), diagonal value last.
//
// A loader wave fetches chunk after chunk with three fully coalesced 1 KB loads (plus the lane's right-hand
// side), consumer waves run the steps in lock-step out of an LDS ring, results are handed over through the
// same {tag,x} LDS rings / ghost rings / write-through stores as in sptrsv.hip.  Unknowns are still written to
// the solution vector in natural order.  Liveness: the records are accepted only if every in-workgroup
// dependency has tau(dep) <= tau(row) and every foreign producer belongs to a workgroup with a smaller
// ticket, which orders all (workgroup, step) pairs lexicographically; otherwise the CSR kernels are used.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

#ifndef IMPORT_SLEEP
#define IMPORT_SLEEP 1
#endif

namespace ilupp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

static constexpr int kNoDep = -1;          // record slot without a dependency
static constexpr int kOwnPrev = -3;        // dependency on the lane's own previous row (stays in a register)
static constexpr int kMaxSkew = 30000;
static constexpr unsigned kLmSpinLimit = 1u << 22;

void PackedSweep::release()
{
    if (skew) (void)pool_free(skew);
    if (wtab) (void)pool_free(wtab);
    if (flags) (void)pool_free(flags);
    if (pk) (void)pool_free(pk);
    if (uslot) (void)pool_free(uslot);
    if (ysrc) (void)pool_free(ysrc);
    if (ybuf) (void)pool_free(ybuf);
    if (ltab) (void)pool_free(ltab);
    if (dump) (void)pool_free(dump);
    if (xlm) (void)pool_free(xlm);
    if (xe) (void)pool_free(xe);
    if (xw) (void)pool_free(xw);
    if (xch) (void)pool_free(xch);
    if (pkT) (void)pool_free(pkT);
    pkT = nullptr; pair = false; desc = false;
    dump = nullptr; xlm = nullptr; y_chunks = 0; xe = xw = nullptr; xch = nullptr; xch_len = 0;
    ysrc = nullptr; ybuf = nullptr; ltab = nullptr; stat = false; wx = false; fmt = 0;
    skew = wtab = flags = uslot = nullptr; pk = nullptr; nchunks = 0; valid = false; built = false; linked = false;
}

// ---------------------------------------------------------------------------------------------
// skews: longest-path fixpoint over the constraints sampled from each lane's first, middle and last row
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_lm_skew(const int32_t *__restrict__ ptr, const int32_t *__restrict__ desc, const int32_t *__restrict__ sfirst,
          const int32_t *__restrict__ scount, int dr, int32_t *__restrict__ skew, int32_t *__restrict__ wtab,
          int32_t *__restrict__ flags)
{
    __shared__ int s[kThreads];
    const int wg = blockIdx.x, t = threadIdx.x;
    const int slot = wg * kThreads + t;
    const int cnt = scount[slot], first = sfirst[slot];
    int cb[12], cd[12], nc = 0;
    for (int sample = 0; sample < 3 && cnt > 0; ++sample) {
        const int k = sample == 0 ? 0 : (sample == 1 ? cnt / 2 : cnt - 1);
        if ((sample == 1 && k == 0) || (sample == 2 && (k == 0 || k == cnt / 2))) continue;
        const int r = first + dr * k;
        const int q0 = ptr[r], q1 = ptr[r + 1];
        for (int q = q0; q < q1 && q < q0 + 5; ++q) {
            const int d = desc[q];
            if (d == -1) continue;
            const unsigned os = (unsigned)d >> 15;
            const int kl = d & 0x7fff;
            if (os >= (unsigned)kGhostBase || (int)(os >> 8) != wg || (int)(os & 255u) == t) continue;
            if (nc < 12) { cb[nc] = (int)(os & 255u); cd[nc] = kl + 1 - k; ++nc; }
        }
    }
    s[t] = 0;
    __syncthreads();
    for (int it = 0; it < 600; ++it) {
        int v = s[t];
        for (int j = 0; j < nc; ++j) { const int c = s[cb[j]] + cd[j]; v = c > v ? c : v; }
        v = v > kMaxSkew ? kMaxSkew : v;
        const int changed = v != s[t];
        __syncthreads();
        s[t] = v;
        if (!__syncthreads_or(changed)) break;
    }
    const int sk = s[t];
    skew[slot] = sk;
    if (sk >= kMaxSkew) atomicOr(&flags[0], 1);
    // chunk range of each wave
    int lo = cnt > 0 ? sk : 0x7fffffff, hi = cnt > 0 ? sk + cnt : -0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
    if ((t & 63) == 0) {
        int32_t *w = wtab + (size_t)(wg * 4 + (t >> 6)) * 4;
        const int nch = hi > lo ? hi - lo : 0;
        w[0] = 0; w[1] = nch > 0 ? lo : 0; w[2] = nch; w[3] = 0;
    }
}

// exclusive scan of the waves' chunk counts (one block); flags[1] = total, flags[2] = longest wave
__global__ void __launch_bounds__(kThreads)
k_lm_scan(int32_t nwaves, int32_t *__restrict__ wtab, int32_t *__restrict__ flags)
{
    __shared__ int part[kThreads];
    __shared__ int pmax[kThreads];
    const int t = threadIdx.x;
    const int per = (nwaves + kThreads - 1) / kThreads;
    int sum = 0, mx = 0;
    for (int i = t * per; i < (t + 1) * per && i < nwaves; ++i) { const int c = wtab[(size_t)i * 4 + 2]; sum += c; mx = c > mx ? c : mx; }
    part[t] = sum; pmax[t] = mx;
    __syncthreads();
    if (t == 0) {
        int run = 0, m = 0;
        for (int i = 0; i < kThreads; ++i) { const int c = part[i]; part[i] = run; run += c; m = pmax[i] > m ? pmax[i] : m; }
        flags[1] = run; flags[2] = m;
    }
    __syncthreads();
    int run = part[t];
    for (int i = t * per; i < (t + 1) * per && i < nwaves; ++i) { wtab[(size_t)i * 4] = run; run += wtab[(size_t)i * 4 + 2]; }
}

// ---------------------------------------------------------------------------------------------
// records
// ---------------------------------------------------------------------------------------------
// WHAT: bit 0 = the pattern half of every record ({d0,d1,d2,flags}, with the liveness checks), bit 1 = the values
template <int KIND, int WHAT>
__global__ void __launch_bounds__(512)
k_lm_pack(const int32_t *__restrict__ ptr, const int32_t *__restrict__ desc, const double *__restrict__ val,
          const int32_t *__restrict__ wtab, const int32_t *__restrict__ skew, const int32_t *__restrict__ sfirst,
          const int32_t *__restrict__ scount, const int32_t *__restrict__ gtab, v4i *__restrict__ pk, int32_t *__restrict__ flags)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr bool DESC = (KIND == SWEEP_BWD_FIRST_DESC);
    constexpr int DR = FWD ? 1 : -1;
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    if (c >= nch) return;
    const int wg = w >> 2;
    const int slot = wg * kThreads + (w & 3) * 64 + L;
    const int tau = tmin + c;
    const int k = tau - skew[slot];
    const bool valid = k >= 0 && k < scount[slot];
    int d[3] = {kNoDep, kNoDep, kNoDep};
    double v[4] = {0.0, 0.0, 0.0, 1.0};
    int bad = 0;
    if (valid) {
        const int r = sfirst[slot] + DR * k;
        const int q0 = ptr[r], q1 = ptr[r + 1];
        const int nd = q1 - q0 - 1;                    // dependencies (the host admits rows of at most 4 entries)
        if (WHAT & 2) v[3] = val[FWD ? q1 - 1 : q0];
        for (int j = 0; j < 3; ++j) {
            if (j < nd) {
                const int q = FWD ? q0 + j : (DESC ? q1 - 1 - j : q0 + 1 + j);
                if (WHAT & 2) v[j] = val[q];
                if (!(WHAT & 1)) continue;
                int dd = desc[q];
                const unsigned os = (unsigned)dd >> 15;
                const int kl = dd & 0x7fff;
                if ((int)os == slot && kl == k - 1) {
                    dd = kOwnPrev;
                } else if (os >= (unsigned)kGhostBase) {
                    const int prod = gtab[(size_t)wg * kGhosts + (os & (kGhosts - 1))];
                    if (prod < 0 || (prod >> 8) >= wg) bad = 1;
                } else if ((int)(os >> 8) == wg) {
                    if (kl + skew[os] > tau) bad = 1;           // would wait for a later step of this workgroup
                } else if ((int)(os >> 8) >= wg) {
                    bad = 1;                                    // producer not ahead of us in ticket order
                }
                d[j] = dd;
            }
        }
        if (nd > 3 || nd < 0) bad = 1;
    }
    if (bad) atomicOr(&flags[0], 2);
    v4i *p = pk + ((size_t)base + c) * 192 + L;
    if (WHAT & 1) {
        v4i rec; rec.x = d[0]; rec.y = d[1]; rec.z = d[2]; rec.w = valid ? 1 : 0;
        p[0] = rec;
    }
    if (WHAT & 2) {
        v2d a; a.x = v[0]; a.y = v[1];
        v2d b; b.x = v[2]; b.y = v[3];
        reinterpret_cast<v2d *>(p)[64] = a;
        reinterpret_cast<v2d *>(p)[128] = b;
    }
}

// slot of the backward schedule that owns the same rows as each slot of the forward schedule; flags[3] is set
// when some forward chain is not also a backward chain (then the factor kernel cannot address the U records)
__global__ void k_lm_uslot(int32_t nslots, const int32_t *__restrict__ slot2blkA, const int32_t *__restrict__ startA,
                           const int32_t *__restrict__ startU, int32_t nbU, const int32_t *__restrict__ blk2slotU,
                           int32_t BU, int32_t *__restrict__ uslot, int32_t *__restrict__ flags)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int b = slot2blkA[s];
    int us = -1;
    if (b >= 0) {
        const int lo = startA[b], hi = startA[b + 1];
        if (hi > lo) {
            const int bu = block_of(lo, BU, nbU, startU);
            if (startU[bu] == lo && startU[bu + 1] == hi) us = blk2slotU[bu]; else atomicOr(&flags[3], 1);
        }
    }
    uslot[s] = us;
}

// where the backward sweep finds, in the forward sweep's level-major vector, the right-hand side of each of its
// slots' FIRST row (its later rows sit 64 doubles further down each): the forward slot f of the same rows has its
// last row at chunk  cnt-1 + skew(f) - tmin(wave of f)
__global__ void k_lm_ysrc(int32_t nslots, const int32_t *__restrict__ uslot, const int32_t *__restrict__ scount,
                          const int32_t *__restrict__ wtabL, const int32_t *__restrict__ skewL, int32_t *__restrict__ ysrc)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nslots) return;
    const int su = uslot[f];
    if (su < 0) return;
    const int w = f >> 6;
    ysrc[su] = (wtabL[(size_t)w * 4] + (scount[f] - 1 + skewL[f] - wtabL[(size_t)w * 4 + 1])) * 64 + (f & 63);
}

// ---------------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------------
#ifndef KLB
#define KLB 3
#endif
#ifndef LM_CD
#define LM_CD 8
#endif
#ifndef LM_XD
#define LM_XD 8
#endif
#ifndef LM_BACK
#define LM_BACK 4
#endif
static constexpr int kCD = LM_CD;              // chunk ring depth
static constexpr int kLB = KLB;              // chunks a loader fetches per round
static constexpr int kXD8 = LM_XD;             // {tag,x} hand-off ring depth
static constexpr int kGD8 = 8;             // ghost ring depth
static constexpr int kIB4 = 4;             // unknowns an importer lane polls per trip
static constexpr int kBack = LM_BACK;            // a wave runs at most this many steps ahead of the slowest wave of its workgroup
static constexpr int kPatience = 256;
#ifndef LM_LOADER_SETS
#define LM_LOADER_SETS 1
#endif
static constexpr int kLS = LM_LOADER_SETS;   // loader waves per consumer wave (measured: 2 changes nothing, the sweeps are not loader-bound): set s fetches the chunks c = s (mod kLS), so that
                                              // kLS fetch rounds are in flight per consumer wave even when each round tops up one chunk
static constexpr int kLmThreads = 2 * kThreads + 64 + (kLS - 1) * kThreads;
static constexpr size_t kChunkLds = (size_t)kThreads * (16 + 16 + 16 + 8);           // one ring slot: desc, v01, v23, rhs
static constexpr size_t kLmLds = kCD * kChunkLds + (size_t)kXD8 * kThreads * 16 + (size_t)kGD8 * kGhosts * 16 + (16 + 2 * kGhosts + 8) * 4;
static constexpr int kDone = 0x7fffffff;

template <int DR>
__global__ void __launch_bounds__(kLmThreads)
k_sptrsv_lm(const v4i *__restrict__ pk, const int32_t *__restrict__ wtab, const int32_t *__restrict__ skew, int32_t n,
            const double *__restrict__ rhs, double *out, int32_t nslots_used, const int32_t *__restrict__ sfirst,
            const int32_t *__restrict__ scount, const int32_t *__restrict__ exported, const int32_t *__restrict__ gtab,
            double *__restrict__ ypk_out, const double *__restrict__ ypk_in, const int32_t *__restrict__ ysrc,
            int32_t *ticket, int32_t *err)
{
    // ypk_out: the unknowns are (also) stored level-major, next to their record -- one coalesced 512-byte store per wave
    // and step -- and only exported lanes write the natural-order vector; ypk_in/ysrc: the right-hand side is read from
    // such a vector written by the sweep of the opposite direction (lm_link_y), instead of the natural-order one
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ring slot q: [desc 256 x 16][v01 256 x 16][v23 256 x 16][rhs 256 x 8]
    v4i *xr = reinterpret_cast<v4i *>(smem + kCD * kChunkLds);                 // [kXD8][256] {tag,-,x.lo,x.hi}
    v4i *gr = xr + kXD8 * kThreads;                                            // [kGD8][kGhosts]
    int *avail = reinterpret_cast<int *>(gr + kGD8 * kGhosts);                 // [2][4] per loader set and wave: first chunk of the set not yet loaded
    int *cons = avail + 8;                                                     // [4] chunks taken by the consumer wave
    int *wdone = cons + 4;                                                     // [4] last completed step (absolute), kDone when finished
    int *gfirst = wdone + 4;                                                   // [kGhosts]
    int *gack = gfirst + kGhosts;                                              // [kGhosts]
    int *hasg = gack + kGhosts;
    unsigned *wg_ticket = reinterpret_cast<unsigned *>(hasg + 1);
    if (threadIdx.x == 0) *wg_ticket = (unsigned)atomicAdd(ticket, 1);
    __syncthreads();
    const unsigned wg = *wg_ticket;
    const int tid = threadIdx.x & (kThreads - 1);
    const bool is_consumer = threadIdx.x < kThreads;
    const bool is_loader = !is_consumer && (threadIdx.x < 2 * kThreads || threadIdx.x >= 2 * kThreads + 64);
    const int lset = threadIdx.x >= 2 * kThreads + 64 ? 1 + (int)(threadIdx.x - (2 * kThreads + 64)) / kThreads : 0;   // loader set
    const int wv = tid >> 6;                                    // consumer wave / the wave a loader serves
    const unsigned myslot = wg * kThreads + tid;

    int cnt = 0, r0 = 0, sk = 0;
    bool exports = true;
    if (!(!is_consumer && !is_loader) && (int)myslot < nslots_used) {
        cnt = scount[myslot]; r0 = sfirst[myslot]; sk = skew[myslot]; exports = exported[myslot] != 0;
    }
    const int32_t *wt = wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = wt[0], tmin = wt[1], nch = wt[2];
    int g_first = 0, g_cnt = 0;
    if (!is_consumer && !is_loader) {
        const int g = threadIdx.x - 2 * kThreads;
        const int os = gtab ? gtab[(size_t)wg * kGhosts + g] : -1;
        if (os >= 0) { g_first = sfirst[os]; g_cnt = scount[os]; }
        const bool any = __any(os >= 0);
        if (g == 0) *hasg = any ? 1 : 0;
        for (int s = 0; s < kGD8; ++s) { v4i e; e.x = -1; e.y = 0; e.z = 0; e.w = 0; gr[s * kGhosts + g] = e; }
        gfirst[g] = g_first; gack[g] = 0;
    }
    if (is_consumer) {
        for (int s = 0; s < kXD8; ++s) { v4i e; e.x = -1; e.y = 0; e.z = 0; e.w = 0; xr[s * kThreads + tid] = e; }
        if ((tid & 63) == 0) { avail[wv] = 0; avail[4 + wv] = 1; cons[wv] = 0; wdone[wv] = nch > 0 ? tmin - 1 : kDone; }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    if (is_loader) {
        // ------------------------------------------------------------------ loader
        const int L = tid & 63;
        const v4i *p = pk + (size_t)base * 192 + L;
        const long ysrc0 = (ypk_in && cnt > 0) ? (long)ysrc[myslot] : 0;
        int c_next = lset;
        unsigned idle = 0;
        while (c_next < nch) {
            asm volatile("" ::: "memory");
            int lim = cons[wv] + kCD;
            lim = lim < nch ? lim : nch;
            int room = (lim - c_next + kLS - 1) / kLS;              // chunks of this set in [c_next, lim)
            room = __builtin_amdgcn_readfirstlane(room);
            if (room <= 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++idle > kLmSpinLimit) break;
                continue;
            }
            idle = 0;
            v4i d[kLB], a[kLB], b[kLB];
            double rr[kLB];
#pragma unroll
            for (int u = 0; u < kLB; ++u) {
                if (u < room) {
                    const int cu = c_next + kLS * u;
                    const size_t o = (size_t)cu * 192;
                    // read once: streaming loads
                    d[u] = __builtin_nontemporal_load(p + o); a[u] = __builtin_nontemporal_load(p + o + 64); b[u] = __builtin_nontemporal_load(p + o + 128);
                    const int k = tmin + cu - sk;
                    int r = r0 + DR * k;
                    r = (k >= 0 && k < cnt) ? r : r0;
                    r = r < 0 ? 0 : (r >= n ? n - 1 : r);
                    rr[u] = ypk_in ? ypk_in[(k >= 0 && k < cnt) ? (size_t)(ysrc0 - 64 * (long)k) : 0] : rhs[r];
                }
            }
#pragma unroll
            for (int u = 0; u < kLB; ++u) {
                if (u < room) {
                    unsigned char *q = smem + (size_t)((c_next + kLS * u) & (kCD - 1)) * kChunkLds;
                    reinterpret_cast<v4i *>(q)[tid] = d[u];
                    reinterpret_cast<v4i *>(q + kThreads * 16)[tid] = a[u];
                    reinterpret_cast<v4i *>(q + kThreads * 32)[tid] = b[u];
                    reinterpret_cast<double *>(q + kThreads * 48)[tid] = rr[u];
                }
            }
            c_next += kLS * (room < kLB ? room : kLB);
            asm volatile("" ::: "memory");
            if (L == 0) avail[4 * lset + wv] = c_next;
        }
        return;
    }

    const unsigned long long *outb = reinterpret_cast<const unsigned long long *>(out);
    if (!is_consumer) {
        // ---------------------------------------------------------------- importer (as in sptrsv.hip)
        if (!*hasg) return;
        const int g = threadIdx.x - 2 * kThreads;
        int next = 0;
        unsigned idle = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            if (wdone[0] == kDone && wdone[1] == kDone && wdone[2] == kDone && wdone[3] == kDone) break;
            bool did = false;
            if (next < g_cnt) {
                const int ack = gack[g];
                if (ack > next) next = ack;
                int nb = g_cnt - next;
                nb = nb < kIB4 ? nb : kIB4;
                nb = nb < ack + kGD8 - next ? nb : ack + kGD8 - next;
                if (nb > 0) {
                    unsigned long long b[kIB4];
#pragma unroll
                    for (int q = 0; q < kIB4; ++q) {
                        const int cq = g_first + DR * (next + q);
                        b[q] = ld_agent_u64(outb + (cq < 0 ? 0 : (cq >= n ? n - 1 : cq)));
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    int got = 0;
#pragma unroll
                    for (int q = 0; q < kIB4; ++q) got += (got == q && q < nb && b[q] != kSentinel) ? 1 : 0;
#pragma unroll
                    for (int q = 0; q < kIB4; ++q) {
                        if (q < got) {
                            v4i e;
                            e.x = next + q; e.y = 0; e.z = (int)(unsigned)b[q]; e.w = (int)(unsigned)(b[q] >> 32);
                            gr[((next + q) & (kGD8 - 1)) * kGhosts + g] = e;
                        }
                    }
                    next += got;
                    did = got > 0;
                }
            }
            if (__any(did)) {
                idle = 0;
            } else {
                __builtin_amdgcn_s_sleep(IMPORT_SLEEP);
                if (++idle > kLmSpinLimit) break;
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    // Hot path per step: ONE LDS round trip for the hand-shake words plus the record (read speculatively,
    // validated afterwards), then rounds of three {tag,x} ring reads until every dependency has shown up --
    // no branches besides "row complete".  Everything rare (recycled ring entries, producers without a ghost,
    // a silent importer) sits behind one wave-uniform branch.
    double prev_val = 0.0;
    int gpat = kPatience;
    unsigned ec_oslot = 0xffffffffu; int ec_first = 0;
    bool dead = false;
    auto poll_one = [&](int c, double &xc) -> bool {
        const unsigned long long b = ld_agent_u64(outb + c);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        xc = __longlong_as_double((long long)b);
        return b != kSentinel;
    };
    auto poll_slot = [&](unsigned oslot, int kl, double &xc) -> bool {
        if (oslot != ec_oslot) { ec_first = sfirst[oslot]; ec_oslot = oslot; __builtin_amdgcn_s_waitcnt(0x0F70); }
        return poll_one(ec_first + DR * kl, xc);
    };
    const v4i *wdone4 = reinterpret_cast<const v4i *>(wdone);

    for (int c = 0; c < nch && !dead; ++c) {
        const int tau = tmin + c;
        const unsigned char *q = smem + (size_t)(c & (kCD - 1)) * kChunkLds;
        v4i rec; v2d va, vb; double rr;
        unsigned spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            const int av = avail[4 * (kLS > 1 ? (c % kLS) : 0) + wv];
            const v4i wd = *wdone4;
            asm volatile("" ::: "memory");
            rec = reinterpret_cast<const v4i *>(q)[tid];
            va = reinterpret_cast<const v2d *>(q + kThreads * 16)[tid];
            vb = reinterpret_cast<const v2d *>(q + kThreads * 32)[tid];
            rr = reinterpret_cast<const double *>(q + kThreads * 48)[tid];
            asm volatile("" ::: "memory");
            // the chunk must be in LDS, and no wave of the workgroup further behind than the hand-off ring tolerates
            if (av > c && min(min(wd.x, wd.y), min(wd.z, wd.w)) >= tau - 1 - kBack) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kLmSpinLimit) { dead = true; break; }
        }
        if (dead) break;
        if ((tid & 63) == 0) cons[wv] = c + 1;
        const int k = tau - sk;
        const bool valid = rec.w != 0;
        // decode once per step.  kl = -2 never matches a tag, so non-ring dependencies stay out of the hot path
        const int dd[3] = {rec.x, rec.y, rec.z};
        int at[3], kl[3];
        double xs[3] = {0.0, 0.0, 0.0};
        int need = 0, ringm = 0, ghostm = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned d = (unsigned)dd[j];
            const unsigned os = d >> 15;
            const int kk = (int)(d & 0x7fffu);
            const bool real = valid && dd[j] != kNoDep && dd[j] != kOwnPrev;
            const bool isg = os >= (unsigned)kGhostBase;
            const bool ring = real && (isg || (os >> 8) == wg);
            at[j] = isg ? kXD8 * kThreads + (kk & (kGD8 - 1)) * kGhosts + (int)(os & (kGhosts - 1))
                        : (kk & (kXD8 - 1)) * kThreads + (int)(os & 255u);
            kl[j] = ring ? kk : -2;
            if (real) need |= 1 << j;
            if (ring) ringm |= 1 << j;
            if (ring && isg) { ghostm |= 1 << j; gack[os & (kGhosts - 1)] = kk; }     // older entries may be recycled; the importer skips to this one
            if (valid && dd[j] == kOwnPrev) xs[j] = prev_val;
        }
        bool done = !valid;
        int stall = 0;
        spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            const v4i e0 = xr[at[0]], e1 = xr[at[1]], e2 = xr[at[2]];
            asm volatile("" ::: "memory");
            const bool h0 = e0.x == kl[0], h1 = e1.x == kl[1], h2 = e2.x == kl[2];
            xs[0] = h0 ? __hiloint2double(e0.w, e0.z) : xs[0];
            xs[1] = h1 ? __hiloint2double(e1.w, e1.z) : xs[1];
            xs[2] = h2 ? __hiloint2double(e2.w, e2.z) : xs[2];
            need &= ~((h0 ? 1 : 0) | (h1 ? 2 : 0) | (h2 ? 4 : 0));
            if (!done && need == 0) {
                // sequential accumulation in the record's (= the reference's) order
                double acc = rr;
                if (dd[0] != kNoDep) { const double prod = va.x * xs[0]; acc = acc - prod; }
                if (dd[1] != kNoDep) { const double prod = va.y * xs[1]; acc = acc - prod; }
                if (dd[2] != kNoDep) { const double prod = vb.x * xs[2]; acc = acc - prod; }
                double x = acc / vb.y;
                if (x != x) x = __longlong_as_double((long long)kCanonNaN);
                v4i e;
                e.x = k; e.y = 0; e.z = __double2loint(x); e.w = __double2hiint(x);
                xr[(k & (kXD8 - 1)) * kThreads + tid] = e;
                const int r = r0 + DR * k;
                if (ypk_out) {
                    __builtin_nontemporal_store(x, ypk_out + ((size_t)base + c) * 64 + (tid & 63));
                    if (exports) st_agent_f64(out + r, x);
                } else {
                    if (exports) st_agent_f64(out + r, x); else out[r] = x;
                }
                prev_val = x;
                done = true;
            }
            if (__all(done)) break;
            // ---- rare: a dependency that will not (or no longer) show up in a ring
            int slow = 0;
            if (!done) {
                slow = need & ~ringm;                                               // producer without a ghost lane
                if ((need & 1) && (ringm & 1) && e0.x > kl[0]) slow |= 1;           // ring entry already recycled
                if ((need & 2) && (ringm & 2) && e1.x > kl[1]) slow |= 2;
                if ((need & 4) && (ringm & 4) && e2.x > kl[2]) slow |= 4;
                if (stall > gpat) slow |= need & ghostm;                            // nothing arrived for a long time
            }
            if (__any(slow != 0)) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (slow & (1 << j)) {
                        const unsigned d = (unsigned)dd[j];
                        const unsigned os = d >> 15;
                        const int kk = (int)(d & 0x7fffu);
                        double xc = 0.0;
                        bool have;
                        if (os >= (unsigned)kGhostBase) {
                            const int tag = j == 0 ? e0.x : (j == 1 ? e1.x : e2.x);
                            have = poll_one(gfirst[os & (kGhosts - 1)] + DR * kk, xc);
                            if (tag < kk) { if (have) gpat = 0; else stall = 0; }       // the ring failed us once: stop relying on it
                        } else {
                            have = poll_slot(os, kk, xc);
                        }
                        if (have) { xs[j] = xc; need &= ~(1 << j); }
                    }
                }
            }
            ++stall;
            __builtin_amdgcn_s_sleep(0);
            if (++spins > kLmSpinLimit) { dead = true; break; }
        }
        asm volatile("" ::: "memory");
        if ((tid & 63) == 0) wdone[wv] = tau;
    }
    if (dead && (tid & 63) == 0) atomicExch(err, 1);
    if ((tid & 63) == 0) wdone[wv] = kDone;
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
bool lm_prepare(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, const int32_t *desc,
                int32_t max_row_len, PackedSweep *ps)
{
    ps->release();
    static const bool off = getenv("ILUPP_NO_PACKED") != nullptr;
    (void)max_row_len;
    // rows of at most 4 entries (checked per row when the records are built; nnz <= 4n is the cheap necessary condition)
    if (off || !desc || M.nnz > 4 * (int64_t)M.n || sch.nslots < kThreads || M.nnz < 16 || !sch.gtab) return false;
    const int nwg = sch.nslots / kThreads;
    ps->nwg = nwg;
    ILUPP_HIP(pool_malloc(&ps->skew, sizeof(int32_t) * (size_t)sch.nslots));
    ILUPP_HIP(pool_malloc(&ps->wtab, sizeof(int32_t) * 16 * (size_t)nwg));
    ILUPP_HIP(pool_malloc(&ps->flags, 64));
    ILUPP_HIP(hipMemsetAsync(ps->flags, 0, 64, st));
    const int dr = kind == SWEEP_FWD_LAST_ASC ? 1 : -1;
    hipLaunchKernelGGL(k_lm_skew, dim3((unsigned)nwg), dim3(kThreads), 0, st, M.ptr, desc, sch.sfirst, sch.scount, dr,
                       ps->skew, ps->wtab, ps->flags);
    hipLaunchKernelGGL(k_lm_scan, dim3(1), dim3(kThreads), 0, st, nwg * 4, ps->wtab, ps->flags);
    int32_t h[4] = {0, 0, 0, 0};
    ILUPP_HIP(d2h_async(st, h, ps->flags, sizeof(h)));
    ILUPP_HIP(stream_sync(st));
    // give up on irregular structures: skews ran away, or the padding would outweigh the gain
    const int64_t rows_padded = (int64_t)h[1] * 64;
    if (h[0] != 0 || h[1] <= 0 || rows_padded > 2 * (int64_t)M.n + 64 * 4 * (int64_t)nwg) { ps->release(); return false; }
    ps->nchunks = h[1];
    ps->max_chunks = h[2];
    ILUPP_HIP(pool_malloc(&ps->pk, (size_t)ps->nchunks * 3072));
    ps->built = true;
    return true;
}

void lm_pack(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, const int32_t *desc, PackedSweep *ps, int what)
{
    if (!ps->built) return;
    const dim3 grid((unsigned)(ps->nwg * 4), (unsigned)((ps->max_chunks + 7) / 8));
#define PACK(K, W) hipLaunchKernelGGL((k_lm_pack<K, W>), grid, dim3(512), 0, st, M.ptr, desc, M.val, ps->wtab, ps->skew, sch.sfirst, \
                                      sch.scount, sch.gtab, reinterpret_cast<v4i *>(ps->pk), ps->flags)
#define PACKW(K) do { if (what == 1) PACK(K, 1); else if (what == 2) PACK(K, 2); else PACK(K, 3); } while (0)
    switch (kind) {
    case SWEEP_FWD_LAST_ASC: PACKW(SWEEP_FWD_LAST_ASC); break;
    case SWEEP_BWD_FIRST_ASC: PACKW(SWEEP_BWD_FIRST_ASC); break;
    default: PACKW(SWEEP_BWD_FIRST_DESC); break;
    }
#undef PACKW
#undef PACK
    ILUPP_HIP(hipGetLastError());
    ps->kind = (int)kind;
}

// can the factor kernel (forward schedule fwd) write the records of the backward sweep ps_u (schedule bwd) itself?
// Builds the slot map it needs; the answer is in flags[3] of ps_u, read by lm_finish.
void lm_link_factor(hipStream_t st, const Schedule &fwd, const Schedule &bwd, PackedSweep *ps_u)
{
    if (!ps_u->built) return;
    ILUPP_HIP(pool_malloc(&ps_u->uslot, sizeof(int32_t) * (size_t)fwd.nslots));
    hipLaunchKernelGGL(k_lm_uslot, dim3((unsigned)((fwd.nslots + 255) / 256)), dim3(256), 0, st, fwd.nslots, fwd.slot2blk, fwd.start,
                       bwd.start, bwd.nb, bwd.blk2slot, bwd.B, ps_u->uslot, ps_u->flags);
}

// after the stream has been synchronised: did the records pass the liveness checks?
bool lm_finish(hipStream_t st, PackedSweep *ps)
{
    if (!ps->built) return false;
    int32_t h[4] = {0, 0, 0, 0};
    ILUPP_HIP(d2h_async(st, h, ps->flags, sizeof(h)));
    ILUPP_HIP(stream_sync(st));
    static const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[ilupp] packed sweep kind %d: flags %d/%d, %lld chunks\n", ps->kind, h[0], h[3], (long long)ps->nchunks);
    if (h[0] != 0) { ps->release(); return false; }
    ps->valid = true;
    ps->linked = ps->uslot != nullptr && h[3] == 0;
    return true;
}

// the forward sweep pl may hand its result to the backward sweep pu level-major (pu linked to pl's schedule)
void lm_link_y(hipStream_t st, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu)
{
    if (!pl->built || !pu->built || !pu->uslot) return;
    ILUPP_HIP(pool_malloc(&pl->ybuf, sizeof(double) * 64 * (size_t)pl->nchunks));
    ILUPP_HIP(pool_malloc(&pu->ysrc, sizeof(int32_t) * (size_t)fwd.nslots));
    ILUPP_HIP(hipMemsetAsync(pu->ysrc, 0, sizeof(int32_t) * (size_t)fwd.nslots, st));
    hipLaunchKernelGGL(k_lm_ysrc, dim3((unsigned)((fwd.nslots + 255) / 256)), dim3(256), 0, st, fwd.nslots, pu->uslot, fwd.scount,
                       pl->wtab, pl->skew, pu->ysrc);
}

int sptrsv_lm(hipStream_t st, const PackedSweep &ps, const Schedule &sch, int32_t n, const double *rhs, double *out,
              int32_t *d_ticket, int32_t *d_err, double *ypk_out, const double *ypk_in, const int32_t *ysrc)
{
    const unsigned grid = (unsigned)ps.nwg;
    {
        static std::once_flag once[64];      // once per device
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_lm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLmLds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_lm<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLmLds));
        });
    }
    if (ps.kind == (int)SWEEP_FWD_LAST_ASC) {
        hipLaunchKernelGGL((k_sptrsv_lm<1>), dim3(grid), dim3(kLmThreads), kLmLds, st, reinterpret_cast<const v4i *>(ps.pk), ps.wtab,
                           ps.skew, n, rhs, out, sch.nslots, sch.sfirst, sch.scount, sch.exported, sch.gtab, ypk_out, ypk_in, ysrc, d_ticket, d_err);
    } else {
        hipLaunchKernelGGL((k_sptrsv_lm<-1>), dim3(grid), dim3(kLmThreads), kLmLds, st, reinterpret_cast<const v4i *>(ps.pk), ps.wtab,
                           ps.skew, n, rhs, out, sch.nslots, sch.sfirst, sch.scount, sch.exported, sch.gtab, ypk_out, ypk_in, ysrc, d_ticket, d_err);
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

}  // namespace ilupp
