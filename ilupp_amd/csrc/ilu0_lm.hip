// ilupp_amd/csrc/ilu0_lm.hip -- level-major ILU(0) numeric factorisation for short-row matrices (gfx950).
//
// Same arithmetic as k_ilu0_numeric_lc (ilu0.hip; reference ILU0.hpp:69-106: row-wise IKJ, eliminations in
// ascending k, the matches of each in ascending column, separate multiply and subtract), driven by the same
// fixed update program (schedule.hip "F3").  What changes is the traffic: the rows a wave eliminates at one
// dependency step arrive as one lane-interleaved chunk (the chunks of the forward sweep, sptrsv_lm.hip --
// factor rows and L-solve rows have the same dependencies, hence the same steps), and the results leave as
// the value halves of the two sweeps' records.  Nothing is written to the CSR value arrays; they are filled
// on demand (lm_unpack) when a caller asks for the factors.
//
//   input per (chunk, lane)    80 bytes: {a0..a6 | w0,w1}: the row of A, diagonal-aligned (diagonal at a3), and the
//                              two header words of its F3 program; {dep0,dep1,dep2,hdr}: who produces the U rows the
//                              row eliminates with, already decoded into hand-off ring addresses and tags
//   output per (chunk, lane)   L record values {l0,l1},{l2,1.0} -- one coalesced 2 KB store per wave and step --
//                              and U record values {u1,u2},{u3,u0} at the row's place in the backward sweep
//   hand-off inside a workgroup   48-byte LDS ring entries {tag,-,u0},{u1,u2},{u3,-}
//   hand-off between workgroups   exported lanes also store their U rows (write-through, sentinel = not yet)
//                              in a dense exchange buffer; an importer wave feeds them into ghost ring entries
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include <hipcub/hipcub.hpp>

#include "common.h"

#ifndef IMPORT_SLEEP
#define IMPORT_SLEEP 1
#endif

namespace ilupp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

static constexpr unsigned kFlmSpinLimit = 1u << 22;
// hand-off ring: [4][3][256] own entries, then [4][3][kGhosts] ghost entries (addresses precomputed by records_lm.hip)

void FactorLM::release()
{
    if (pkA) (void)pool_free(pkA);
    if (xbase) (void)pool_free(xbase);
    if (xch) (void)pool_free(xch);
    if (xcount) (void)pool_free(xcount);
    pkA = nullptr; xbase = nullptr; xch = nullptr; xcount = nullptr; built = false; values_packed = false; stat = false; direct = false; wxf = false; spec = false;
}

// ---------------------------------------------------------------------------------------------
// per-factorisation helpers
// ---------------------------------------------------------------------------------------------
__global__ void k_flm_fill(unsigned long long *__restrict__ p, const long long *__restrict__ count, unsigned long long v)
{
    const long long n = *count;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// the rows of A, diagonal-aligned, into the factor records (whose header words the analysis wrote: records_lm.hip)
struct __attribute__((aligned(8))) D2a { double v[2]; };
__global__ void __launch_bounds__(512)
k_flm_pack_a(const int32_t *__restrict__ Aptr, const double *__restrict__ Aval, int64_t nnz,
             const int32_t *__restrict__ wtab, const int32_t *__restrict__ skew, const int32_t *__restrict__ sfirst,
             const int32_t *__restrict__ scount, v4i *__restrict__ pkA)
{
    // a block owns the 64 lanes of one wave x 8 consecutive rows of each (8 lanes x 8 rows per wave: the loads of a
    // wave touch 8 contiguous pieces of A, its stores diagonals of 8 neighbouring places)
    const int w = blockIdx.x;
    const int L = (threadIdx.x >> 6) * 8 + (threadIdx.x & 7);
    const int k = blockIdx.y * 8 + ((threadIdx.x >> 3) & 7);
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const bool valid = k < scount[slot];
    if (!valid) return;
    const int base = wtab[(size_t)w * 4], c = k + skew[slot] - wtab[(size_t)w * 4 + 1];
    v4i *p = pkA + ((size_t)base + c) * 320 + L;
    double a[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (valid) {
        const int r = sfirst[slot] + k;
        const int w0 = reinterpret_cast<const int *>(p + 192)[2];
        const int len = w0 & 15, cl = (w0 >> 4) & 3;
        const int a0 = Aptr[r];
        // the row's (at most 7) values with four 16-byte loads, then shifted into diagonal-aligned position
        double v[8];
        if ((int64_t)a0 + 8 <= nnz) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const D2a t = *reinterpret_cast<const D2a *>(Aval + a0 + 2 * i); v[2 * i] = t.v[0]; v[2 * i + 1] = t.v[1]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (int64_t)a0 + i < nnz ? Aval[a0 + i] : 0.0;
        }
        const int sh = 3 - cl;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int e = j - sh;
            const double x = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : e == 3 ? v[3] : e == 4 ? v[4] : e == 5 ? v[5] : v[6];
            a[j] = (e >= 0 && e < len) ? x : 0.0;
        }
    }
    v2d x; x.x = a[0]; x.y = a[1]; reinterpret_cast<v2d *>(p)[0] = x;
    x.x = a[2]; x.y = a[3]; reinterpret_cast<v2d *>(p)[64] = x;
    x.x = a[4]; x.y = a[5]; reinterpret_cast<v2d *>(p)[128] = x;
    reinterpret_cast<double *>(p + 192)[0] = a[6];
}

// ---------------------------------------------------------------------------------------------
// the factor kernel
// ---------------------------------------------------------------------------------------------
#ifndef KLF
#define KLF 3
#endif
static constexpr int kCF = 4;               // chunk ring depth (not a power of two: slots are taken modulo)
static constexpr int kLF = KLF;               // chunks a loader fetches per round
static constexpr int kUF = kFlmUF;          // own U-row hand-off ring depth (rows per lane)
static constexpr int kGF = kFlmGF;          // ghost U-row ring depth
static constexpr int kBackF = kFlmUF - 2;   // a wave runs at most this many steps ahead of the slowest wave of its workgroup
static constexpr int kIF = 2;               // rows an importer lane polls per trip
static constexpr int kPatienceF = 256;
static constexpr int kFlmThreads = 2 * kThreads;   // 4 consumer waves + 4 loader waves (16 lanes of each loader double as importers)
static constexpr size_t kChunkF = (size_t)kThreads * 80;                                  // desc + 4 x 16 B of A / header
static constexpr size_t kFlmLds = kCF * kChunkF + (size_t)kUF * 3 * kThreads * 16 + (size_t)kGF * 3 * kGhosts * 16 + (12 + kGhosts + 8) * 4;
static constexpr int kDoneF = 0x7fffffff;

// the record streams are read once and written once: keep them out of the way of the data that is re-used
#define FLM_LD(p) __builtin_nontemporal_load(p)
#define FLM_ST(p, v) __builtin_nontemporal_store(v, p)

#ifdef ILUPP_TIMELINE
__device__ unsigned long long *g_flm_timeline = nullptr;     // diagnostics build only: 8 words per workgroup
#endif

struct FlmArgs {
    const v4i *pkL_in; v2d *pkL_out;          // forward sweep records: pattern half read, value halves written
    const v4i *pkA;
    v2d *pkU_out;
    const int32_t *wtabL, *skewL, *wtabU, *skewU, *uslot;
    const int32_t *sfirst, *scount, *exported, *gtab, *xbase;
    double *xch;
    int32_t nslots_used;
    int32_t *ctrl;
};

__global__ void __launch_bounds__(kFlmThreads)
k_ilu0_lm(FlmArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i *ur = reinterpret_cast<v4i *>(smem + kCF * kChunkF);            // [kUF][3][256]: {tag,-,u0} {u1,u2} {u3,-}
    v4i *gur = ur + kUF * 3 * kThreads;                                  // [kGF][3][kGhosts]
    int *avail = reinterpret_cast<int *>(gur + kGF * 3 * kGhosts);       // [4]
    int *cons = avail + 4;                                               // [4]
    int *wdone = cons + 4;                                               // [4]
    int *gack = wdone + 4;                                               // [kGhosts]
    int *hasg = gack + kGhosts;
    unsigned *wg_ticket = reinterpret_cast<unsigned *>(hasg + 1);
    if (threadIdx.x == 0) *wg_ticket = (unsigned)atomicAdd(&A.ctrl[0], 1);
    __syncthreads();
    const unsigned wg = *wg_ticket;
    const int tid = threadIdx.x & (kThreads - 1);
    const bool is_consumer = threadIdx.x < kThreads;
    const int wv = tid >> 6;
    const unsigned myslot = wg * kThreads + tid;

    int cnt = 0, sk = 0;
    bool exports = false;
    int xb = -1;
    if ((int)myslot < A.nslots_used) {
        cnt = A.scount[myslot]; sk = A.skewL[myslot];
        if (is_consumer) { xb = A.xbase[myslot]; exports = xb >= 0; }
    }
    const int32_t *wt = A.wtabL + (size_t)(wg * 4 + wv) * 4;
    const int base = wt[0], tmin = wt[1], nch = wt[2];
    // importer duty: lane L < 16 of loader wave j follows ghost 16 j + L
    const bool is_importer = !is_consumer && (tid & 63) < kGhosts / 4;
    const int gid = wv * (kGhosts / 4) + (tid & 63);
    int g_xb = 0, g_cnt = 0;
    if (is_importer) {
        const int os = A.gtab[(size_t)wg * kGhosts + gid];
        if (os >= 0) { g_xb = A.xbase[os]; g_cnt = g_xb >= 0 ? A.scount[os] : 0; }
        for (int s = 0; s < kGF * 3; ++s) { v4i e; e.x = -1; e.y = 0; e.z = 0; e.w = 0; gur[s * kGhosts + gid] = e; }
        gack[gid] = 0;
    }
    if (is_consumer) {
        for (int s = 0; s < kUF * 3; ++s) { v4i e; e.x = -1; e.y = 0; e.z = 0; e.w = 0; ur[s * kThreads + tid] = e; }
        if ((tid & 63) == 0) { avail[wv] = 0; cons[wv] = 0; wdone[wv] = nch > 0 ? tmin - 1 : kDoneF; }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    const unsigned long long *xchb = reinterpret_cast<const unsigned long long *>(A.xch);
    if (!is_consumer) {
        // ------------------------------------------------------------------ loader (+ importer lanes)
        // every round: the chunks the ring has room for (five coalesced 1 KB loads each) and, on the importer lanes,
        // the next exchange rows of their ghosts; one wait; everything into LDS
        const int L = tid & 63;
        const v4i *pa = A.pkA + (size_t)base * 320 + L;
        int c_next = 0, next = 0;
        unsigned idle = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            const v4i wd = *reinterpret_cast<const v4i *>(wdone);
            if (wd.x == kDoneF && wd.y == kDoneF && wd.z == kDoneF && wd.w == kDoneF) break;
            int room = cons[wv] + kCF - c_next;
            room = room < nch - c_next ? room : nch - c_next;
            room = __builtin_amdgcn_readfirstlane(room);
            int nb = 0;
            if (is_importer && next < g_cnt) {
                const int ack = gack[gid];
                if (ack > next) next = ack;
                nb = g_cnt - next;
                nb = nb < kIF ? nb : kIF;
                nb = nb < ack + kGF - next ? nb : ack + kGF - next;
            }
            if (room <= 0 && !__any(nb > 0)) {
                __builtin_amdgcn_s_sleep(2);
                if (++idle > kFlmSpinLimit) break;
                continue;
            }
            v4i d[kLF], a0[kLF], a1[kLF], a2[kLF], a3[kLF];
            unsigned long long b[kIF][4];
#pragma unroll
            for (int u = 0; u < kLF; ++u) {
                if (u < room) {
                    const size_t o = (size_t)(c_next + u);
                    d[u] = FLM_LD(pa + o * 320 + 256);
                    a0[u] = FLM_LD(pa + o * 320); a1[u] = FLM_LD(pa + o * 320 + 64); a2[u] = FLM_LD(pa + o * 320 + 128); a3[u] = FLM_LD(pa + o * 320 + 192);
                }
            }
            if (nb > 0) {
#pragma unroll
                for (int q = 0; q < kIF; ++q) {
                    const int kq = next + q < g_cnt ? next + q : g_cnt - 1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[q][e] = ld_agent_u64(xchb + ((size_t)g_xb + kq) * 4 + e);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (int u = 0; u < kLF; ++u) {
                if (u < room) {
                    v4i *q = reinterpret_cast<v4i *>(smem + (size_t)((c_next + u) % kCF) * kChunkF);
                    q[tid] = d[u]; q[kThreads + tid] = a0[u]; q[2 * kThreads + tid] = a1[u];
                    q[3 * kThreads + tid] = a2[u]; q[4 * kThreads + tid] = a3[u];
                }
            }
            bool did = room > 0;
            if (room > 0) c_next += room < kLF ? room : kLF;
            if (nb > 0) {
                int got = 0;
#pragma unroll
                for (int q = 0; q < kIF; ++q)
                    got += (got == q && q < nb && b[q][0] != kSentinel && b[q][1] != kSentinel && b[q][2] != kSentinel && b[q][3] != kSentinel) ? 1 : 0;
#pragma unroll
                for (int q = 0; q < kIF; ++q) {
                    if (q < got) {
                        v4i *e = gur + (size_t)((next + q) & (kGF - 1)) * 3 * kGhosts + gid;
                        v4i w1, w2, w0;
                        w1.x = (int)(unsigned)b[q][1]; w1.y = (int)(unsigned)(b[q][1] >> 32); w1.z = (int)(unsigned)b[q][2]; w1.w = (int)(unsigned)(b[q][2] >> 32);
                        w2.x = (int)(unsigned)b[q][3]; w2.y = (int)(unsigned)(b[q][3] >> 32); w2.z = 0; w2.w = 0;
                        w0.x = next + q; w0.y = 0; w0.z = (int)(unsigned)b[q][0]; w0.w = (int)(unsigned)(b[q][0] >> 32);
                        e[kGhosts] = w1; e[2 * kGhosts] = w2;
                        asm volatile("" ::: "memory");
                        e[0] = w0;                                 // the tag word last
                    }
                }
                next += got;
                did = did || got > 0;
            }
            asm volatile("" ::: "memory");
            if (L == 0) avail[wv] = c_next;
            if (__any(did)) {
                idle = 0;
            } else {
                __builtin_amdgcn_s_sleep(IMPORT_SLEEP);
                if (++idle > kFlmSpinLimit) break;
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    double pu0 = 0.0, pu1 = 0.0, pu2 = 0.0, pu3 = 0.0;          // U row of the lane's previous row
    int gpat = kPatienceF;
    bool dead = false;
    const v4i *wdone4 = reinterpret_cast<const v4i *>(wdone);
    // this lane's rows in the backward sweep's records: 192 sixteen-byte units apart, descending
    long up0 = 0;
    if (cnt > 0) {
        const int su = A.uslot[myslot];
        const int wu = su >> 6;
        up0 = ((long)A.wtabU[wu * 4] + (cnt - 1 + A.skewU[su] - A.wtabU[wu * 4 + 1])) * 192 + 64 + (su & 63);
    }
    v2d *lout = A.pkL_out + (size_t)base * 192 + 64 + (tid & 63);

#ifdef ILUPP_TIMELINE
    unsigned long long *const tl = g_flm_timeline;
    if (tl && tid == 0) tl[(size_t)wg * 8] = wall_clock64();
#endif
    for (int c = 0; c < nch && !dead; ++c) {
#ifdef ILUPP_TIMELINE
        if (tl && (tid & 63) == 0 && (c == 0 || c == nch / 2 || c == nch - 1))
            tl[(size_t)wg * 8 + 1 + (wv == 0 ? 0 : 3) + (c == 0 ? 0 : (c == nch - 1 ? 2 : 1))] = wall_clock64();
#endif
        const int tau = tmin + c;
        const v4i *q = reinterpret_cast<const v4i *>(smem + (size_t)(c % kCF) * kChunkF);
        v4i rec, r0, r1, r2, r3;
        unsigned spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            const int av = avail[wv];
            const v4i wd = *wdone4;
            asm volatile("" ::: "memory");
            rec = q[tid]; r0 = q[kThreads + tid]; r1 = q[2 * kThreads + tid]; r2 = q[3 * kThreads + tid]; r3 = q[4 * kThreads + tid];
            asm volatile("" ::: "memory");
            if (av > c && min(min(wd.x, wd.y), min(wd.z, wd.w)) >= tau - 1 - kBackF) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kFlmSpinLimit) { dead = true; break; }
        }
        if (dead) break;
        if ((tid & 63) == 0) cons[wv] = c + 1;
        const int k = tau - sk;
        const int hdr = rec.w;
        const bool valid = (hdr & 1) != 0;
        double w0 = __hiloint2double(r0.y, r0.x), w1 = __hiloint2double(r0.w, r0.z), w2 = __hiloint2double(r1.y, r1.x),
               w3 = __hiloint2double(r1.w, r1.z), w4 = __hiloint2double(r2.y, r2.x), w5 = __hiloint2double(r2.w, r2.z),
               w6 = __hiloint2double(r3.y, r3.x);
        const int pw0 = r3.z, pw1 = r3.w;
        const int cl = (hdr >> 7) & 3, ulen = (hdr >> 9) & 7, nmt = (hdr >> 13) & 7;
        const int s0 = 3 - cl;
        // dependency slots (right-aligned, as in the program), decoded when the records were built:
        // kind 0 none, 1 the lane's own previous row, 2 hand-off ring (address | tag << 12 | off << 27 | ghost << 29), 3 exchange buffer
        const int dw[3] = {rec.x, rec.y, rec.z};
        int at[3], kl[3], off[3];
        int need = 0, ringm = 0, ghostm = 0, ownm = 0;
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int kind = (hdr >> (1 + 2 * sl)) & 3;
            const bool ring = kind == 2;
            at[sl] = ring ? (dw[sl] & 0xfff) : 0;
            kl[sl] = ring ? ((dw[sl] >> 12) & 0x7fff) : -2;
            off[sl] = (dw[sl] >> 27) & 3;
            if (kind >= 2) need |= 1 << sl;
            if (ring) ringm |= 1 << sl;
            if (ring && ((dw[sl] >> 29) & 1)) { ghostm |= 1 << sl; gack[dw[sl] & (kGhosts - 1)] = kl[sl]; }
            if (kind == 1) ownm |= 1 << sl;
        }
        bool done = !valid;
        int stall = 0;
        spins = 0;

#ifdef EXP_FLM_NOL
#define FLM_STORE_L() do { } while (0)
#else
#define FLM_STORE_L() do { FLM_ST(lout + (size_t)c * 192, la_); FLM_ST(lout + (size_t)c * 192 + 64, lb_); } while (0)
#endif
#ifdef EXP_FLM_NOU
#define FLM_STORE_U() do { (void)uo_; } while (0)
#else
#define FLM_STORE_U() do { FLM_ST(uo_, ua_); FLM_ST(uo_ + 64, ub_); } while (0)
#endif
        // results of a finished row: hand-off entry (value words first, the tag word last), exchange row, the two
        // sweeps' records (eliminations in stored order + unit diagonal; strictly-upper entries + pivot)
#define FLM_PUBLISH()                                                                                                  \
        do {                                                                                                           \
            if ((unsigned long long)__double_as_longlong(w3) == kSentinel) w3 = __longlong_as_double((long long)kCanonNaN); \
            if ((unsigned long long)__double_as_longlong(w4) == kSentinel) w4 = __longlong_as_double((long long)kCanonNaN); \
            if ((unsigned long long)__double_as_longlong(w5) == kSentinel) w5 = __longlong_as_double((long long)kCanonNaN); \
            if ((unsigned long long)__double_as_longlong(w6) == kSentinel) w6 = __longlong_as_double((long long)kCanonNaN); \
            if (ulen < 2) w4 = 0.0;                                                                                    \
            if (ulen < 3) w5 = 0.0;                                                                                    \
            if (ulen < 4) w6 = 0.0;                                                                                    \
            v4i *e_ = ur + (size_t)(k % kUF) * 3 * kThreads + tid;                                                     \
            v4i e1_, e2_, e0_;                                                                                         \
            e1_.x = __double2loint(w4); e1_.y = __double2hiint(w4); e1_.z = __double2loint(w5); e1_.w = __double2hiint(w5); \
            e2_.x = __double2loint(w6); e2_.y = __double2hiint(w6); e2_.z = 0; e2_.w = 0;                              \
            e0_.x = k; e0_.y = 0; e0_.z = __double2loint(w3); e0_.w = __double2hiint(w3);                              \
            e_[kThreads] = e1_; e_[2 * kThreads] = e2_;                                                                \
            asm volatile("" ::: "memory");                                                                             \
            e_[0] = e0_;                                                                                               \
            if (exports) {                                                                                             \
                double *xr_ = A.xch + ((size_t)xb + k) * 4;                                                            \
                st_agent_f64(xr_, w3); st_agent_f64(xr_ + 1, w4); st_agent_f64(xr_ + 2, w5); st_agent_f64(xr_ + 3, w6); \
            }                                                                                                          \
            v2d la_, lb_, ua_, ub_;                                                                                    \
            la_.x = s0 == 0 ? w0 : (s0 == 1 ? w1 : w2); la_.y = s0 == 0 ? w1 : w2;                                     \
            lb_.x = w2; lb_.y = 1.0;                                                                                   \
            ua_.x = w4; ua_.y = w5; ub_.x = w6; ub_.y = w3;                                                            \
            FLM_STORE_L();                                                                                             \
            v2d *uo_ = A.pkU_out + (up0 - 192 * (long)k);                                                              \
            FLM_STORE_U();                                                                                             \
            pu0 = w3; pu1 = w4; pu2 = w5; pu3 = w6;                                                                    \
            done = true;                                                                                               \
        } while (0)

        // ---- fast path: every row of the wave is "simple" (each elimination has one match, on the diagonal --
        // all rows of a 5-/7-point stencil) and reads rings only.  Per round: 3 tag/pivot words + 3 match words.
        const bool simple = ((hdr >> 12) & 1) != 0;
        bool generic = !__all(!valid || (simple && (need & ~ringm) == 0));
        if (!generic) {
            // match m belongs to slot s0 + m; the word that holds U entry `off` of that slot's row
            int mat[3];
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                const bool g = ((ghostm >> sl) & 1) != 0;
                mat[sl] = ((ringm >> sl) & 1) ? at[sl] + (off[sl] == 3 ? 2 : 1) * (g ? kGhosts : kThreads) : 0;
            }
            for (;;) {
                asm volatile("" ::: "memory");
                const v4i p0 = ur[at[0]], p1 = ur[at[1]], p2 = ur[at[2]];
                asm volatile("" ::: "memory");                           // tag words before the value words they guard
#ifdef EXP_FLM_CONST_U
                // experiment only (valid for matrices whose off-diagonals are all -1, e.g. the 7-point Poisson matrix): how much of
                // the step is the second half of the hand-off (the matched U entries)?
                v4i m0, m1, m2; m0.x = 0; m0.y = (int)0xBFF00000; m0.z = 0; m0.w = (int)0xBFF00000; m1 = m0; m2 = m0;
#else
                const v4i m0 = ur[mat[0]], m1 = ur[mat[1]], m2 = ur[mat[2]];
#endif
                asm volatile("" ::: "memory");
                const int hit = (p0.x == kl[0] ? 1 : 0) | (p1.x == kl[1] ? 2 : 0) | (p2.x == kl[2] ? 4 : 0);
                if (!done && (need & ~hit) == 0) {
                    const double piv0 = (ownm & 1) ? pu0 : __hiloint2double(p0.w, p0.z);
                    const double piv1 = (ownm & 2) ? pu0 : __hiloint2double(p1.w, p1.z);
                    const double piv2 = (ownm & 4) ? pu0 : __hiloint2double(p2.w, p2.z);
                    const double own0 = off[0] == 1 ? pu1 : (off[0] == 2 ? pu2 : pu3);
                    const double own1 = off[1] == 1 ? pu1 : (off[1] == 2 ? pu2 : pu3);
                    const double own2 = off[2] == 1 ? pu1 : (off[2] == 2 ? pu2 : pu3);
                    const double u0v = (ownm & 1) ? own0 : (off[0] == 2 ? __hiloint2double(m0.w, m0.z) : __hiloint2double(m0.y, m0.x));
                    const double u1v = (ownm & 2) ? own1 : (off[1] == 2 ? __hiloint2double(m1.w, m1.z) : __hiloint2double(m1.y, m1.x));
                    const double u2v = (ownm & 4) ? own2 : (off[2] == 2 ? __hiloint2double(m2.w, m2.z) : __hiloint2double(m2.y, m2.x));
                    if (0 >= s0) { const double l = w0 / piv0; const double pr = l * u0v; w3 = w3 - pr; w0 = l; }
                    if (1 >= s0) { const double l = w1 / piv1; const double pr = l * u1v; w3 = w3 - pr; w1 = l; }
                    if (2 >= s0) { const double l = w2 / piv2; const double pr = l * u2v; w3 = w3 - pr; w2 = l; }
                    FLM_PUBLISH();
                }
                if (__all(done)) break;
                // a ghost entry that was recycled or never came: the general loop knows how to fetch it
                bool trouble = false;
                if (!done) {
                    const int open = need & ~hit;
                    trouble = ((open & 1) && (ghostm & 1) && p0.x > kl[0]) || ((open & 2) && (ghostm & 2) && p1.x > kl[1]) ||
                              ((open & 4) && (ghostm & 4) && p2.x > kl[2]) || (stall > gpat && (open & ghostm) != 0);
                }
                if (__any(trouble)) { generic = true; break; }
                ++stall;
                __builtin_amdgcn_s_sleep(0);
                if (++spins > kFlmSpinLimit) { dead = true; break; }
            }
        }
        if (generic && !dead) {
            // ---- general path: any program the F3 format can express, dependencies from rings or the exchange buffer
            int me_s[5], me_off[5], me_pp[5], mat[5];
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const unsigned mw = m < 3 ? ((unsigned)pw0 >> (9 + 7 * m)) & 127u : ((unsigned)pw1 >> (7 * (m - 3))) & 127u;
                me_s[m] = (int)(mw & 3u); me_off[m] = (int)((mw >> 2) & 3u); me_pp[m] = (int)((mw >> 4) & 7u);
                const int a = me_s[m] == 0 ? at[0] : (me_s[m] == 1 ? at[1] : at[2]);
                const bool g = ((ghostm >> me_s[m]) & 1) != 0;
                mat[m] = ((ringm >> me_s[m]) & 1) ? a + (me_off[m] == 3 ? 2 : 1) * (g ? kGhosts : kThreads) : 0;
            }
            // values that had to be fetched from the exchange buffer (rare)
            double xp[3] = {0.0, 0.0, 0.0}, xu[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
            int memm = 0;
            for (;;) {
                asm volatile("" ::: "memory");
                const v4i p0 = ur[at[0]], p1 = ur[at[1]], p2 = ur[at[2]];
                asm volatile("" ::: "memory");
                const v4i m0 = ur[mat[0]], m1 = ur[mat[1]], m2 = ur[mat[2]], m3 = ur[mat[3]], m4 = ur[mat[4]];
                asm volatile("" ::: "memory");
                const int hit = (p0.x == kl[0] ? 1 : 0) | (p1.x == kl[1] ? 2 : 0) | (p2.x == kl[2] ? 4 : 0);
                if (!done && (need & ~(hit | memm)) == 0) {
                    double piv[3], um[5];
                    piv[0] = (ownm & 1) ? pu0 : ((memm & 1) ? xp[0] : __hiloint2double(p0.w, p0.z));
                    piv[1] = (ownm & 2) ? pu0 : ((memm & 2) ? xp[1] : __hiloint2double(p1.w, p1.z));
                    piv[2] = (ownm & 4) ? pu0 : ((memm & 4) ? xp[2] : __hiloint2double(p2.w, p2.z));
#pragma unroll
                    for (int m = 0; m < 5; ++m) {
                        const v4i e = m == 0 ? m0 : (m == 1 ? m1 : (m == 2 ? m2 : (m == 3 ? m3 : m4)));
                        const double ring_v = me_off[m] == 2 ? __hiloint2double(e.w, e.z) : __hiloint2double(e.y, e.x);
                        const double own_v = me_off[m] == 1 ? pu1 : (me_off[m] == 2 ? pu2 : pu3);
                        const int sl = me_s[m];
                        const double mem_v = sl == 0 ? (me_off[m] == 1 ? xu[0][0] : (me_off[m] == 2 ? xu[0][1] : xu[0][2]))
                                           : sl == 1 ? (me_off[m] == 1 ? xu[1][0] : (me_off[m] == 2 ? xu[1][1] : xu[1][2]))
                                                     : (me_off[m] == 1 ? xu[2][0] : (me_off[m] == 2 ? xu[2][1] : xu[2][2]));
                        um[m] = ((ownm >> sl) & 1) ? own_v : (((memm >> sl) & 1) ? mem_v : ring_v);
                    }
                    // eliminations in ascending k (= ascending slot), matches of each in ascending column
                    // (reference merge order, ILU0.hpp:8-23, :47-62); every position is a fixed register
                    if (simple) {
                        const double u0v = s0 == 0 ? um[0] : 0.0;
                        const double u1v = s0 == 0 ? um[1] : (s0 == 1 ? um[0] : 0.0);
                        const double u2v = s0 == 0 ? um[2] : (s0 == 1 ? um[1] : (s0 == 2 ? um[0] : 0.0));
                        if (0 >= s0) { const double l = w0 / piv[0]; const double pr = l * u0v; w3 = w3 - pr; w0 = l; }
                        if (1 >= s0) { const double l = w1 / piv[1]; const double pr = l * u1v; w3 = w3 - pr; w1 = l; }
                        if (2 >= s0) { const double l = w2 / piv[2]; const double pr = l * u2v; w3 = w3 - pr; w2 = l; }
                    } else {
#define SEL7(i) ((i) == 0 ? w0 : (i) == 1 ? w1 : (i) == 2 ? w2 : (i) == 3 ? w3 : (i) == 4 ? w4 : (i) == 5 ? w5 : w6)
#define PUT7(i, nv)                                                                                         \
    do {                                                                                                    \
        const int i_ = (i); const double nv_ = (nv);                                                        \
        w0 = i_ == 0 ? nv_ : w0; w1 = i_ == 1 ? nv_ : w1; w2 = i_ == 2 ? nv_ : w2; w3 = i_ == 3 ? nv_ : w3; \
        w4 = i_ == 4 ? nv_ : w4; w5 = i_ == 5 ? nv_ : w5; w6 = i_ == 6 ? nv_ : w6;                          \
    } while (0)
#define ELIM(SL, WS)                                                                                       \
                        if (SL >= s0) {                                                                     \
                            const double l_ik = WS / piv[SL];                                               \
                            _Pragma("unroll") for (int m = 0; m < 5; ++m) {                                 \
                                if (m < nmt && me_s[m] == SL) {                                             \
                                    const double prod = l_ik * um[m];                                       \
                                    if (me_pp[m] == 3) { w3 = w3 - prod; }                                  \
                                    else { const double nv = SEL7(me_pp[m]) - prod; PUT7(me_pp[m], nv); }   \
                                }                                                                           \
                            }                                                                               \
                            WS = l_ik;                                                                      \
                        }
                        ELIM(0, w0)
                        ELIM(1, w1)
                        ELIM(2, w2)
#undef ELIM
#undef SEL7
#undef PUT7
                    }
                    FLM_PUBLISH();
                }
                if (__all(done)) break;
                // ---- rare: a U row that will not (or no longer) show up in a ring: take it from the exchange buffer
                int slow = 0;
                if (!done) {
                    const int open = need & ~(hit | memm);
                    slow = open & ~ringm;
                    if ((open & 1) && (ghostm & 1) && p0.x > kl[0]) slow |= 1;
                    if ((open & 2) && (ghostm & 2) && p1.x > kl[1]) slow |= 2;
                    if ((open & 4) && (ghostm & 4) && p2.x > kl[2]) slow |= 4;
                    if (stall > gpat) slow |= open & ghostm;
                }
                if (__any(slow != 0)) {
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl) {
                        if (slow & (1 << sl)) {
                            const bool isg = ((ghostm >> sl) & 1) != 0;
                            const int kk = isg ? kl[sl] : (dw[sl] & 0x7fff);
                            const int prod = isg ? A.gtab[(size_t)wg * kGhosts + (dw[sl] & (kGhosts - 1))] : (int)((unsigned)dw[sl] >> 15);
                            const int pxb = A.xbase[prod];
                            __builtin_amdgcn_s_waitcnt(0x0F70);
                            unsigned long long b[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) b[e] = ld_agent_u64(xchb + ((size_t)(pxb < 0 ? 0 : pxb) + kk) * 4 + e);
                            __builtin_amdgcn_s_waitcnt(0x0F70);
                            const bool have = pxb >= 0 && b[0] != kSentinel && b[1] != kSentinel && b[2] != kSentinel && b[3] != kSentinel;
                            if (isg) {
                                const int tag = sl == 0 ? p0.x : (sl == 1 ? p1.x : p2.x);
                                if (tag < kk) { if (have) gpat = 0; else stall = 0; }
                            }
                            if (have) {
                                memm |= 1 << sl;
                                xp[sl] = __longlong_as_double((long long)b[0]);
                                xu[sl][0] = __longlong_as_double((long long)b[1]);
                                xu[sl][1] = __longlong_as_double((long long)b[2]);
                                xu[sl][2] = __longlong_as_double((long long)b[3]);
                            }
                        }
                    }
                }
                ++stall;
                __builtin_amdgcn_s_sleep(0);
                if (++spins > kFlmSpinLimit) { dead = true; break; }
            }
        }
#undef FLM_PUBLISH
#undef FLM_STORE_L
#undef FLM_STORE_U
        asm volatile("" ::: "memory");
        if ((tid & 63) == 0) wdone[wv] = tau;
    }
    if (dead && (tid & 63) == 0) atomicExch(&A.ctrl[1], 1);
    if ((tid & 63) == 0) wdone[wv] = kDoneF;
}

// ---------------------------------------------------------------------------------------------
// CSR values on demand
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(512)
k_lm_unpack(const int32_t *__restrict__ ptr, double *__restrict__ val, const int32_t *__restrict__ wtab,
            const int32_t *__restrict__ skew, const int32_t *__restrict__ sfirst, const int32_t *__restrict__ scount,
            const v4i *__restrict__ pk)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr bool DESC = (KIND == SWEEP_BWD_FIRST_DESC);
    constexpr int DR = FWD ? 1 : -1;
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    if (c >= nch) return;
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const int k = tmin + c - skew[slot];
    if (k < 0 || k >= scount[slot]) return;
    const int r = sfirst[slot] + DR * k;
    const int q0 = ptr[r], q1 = ptr[r + 1];
    const int nd = q1 - q0 - 1;
    const v4i *p = pk + ((size_t)base + c) * 192 + L;
    const v2d a = reinterpret_cast<const v2d *>(p)[64], b = reinterpret_cast<const v2d *>(p)[128];
    const double v[3] = {a.x, a.y, b.x};
    val[FWD ? q1 - 1 : q0] = b.y;
    for (int j = 0; j < 3; ++j)
        if (j < nd) val[FWD ? q0 + j : (DESC ? q1 - 1 - j : q0 + 1 + j)] = v[j];
}

void lm_unpack(hipStream_t st, const DevMat &M, const Schedule &sch, const PackedSweep &ps)
{
    const dim3 grid((unsigned)(ps.nwg * 4), (unsigned)((ps.max_chunks + 7) / 8));
#define UNPACK(K) hipLaunchKernelGGL((k_lm_unpack<K>), grid, dim3(512), 0, st, M.ptr, M.val, ps.wtab, ps.skew, sch.sfirst, sch.scount, \
                                     reinterpret_cast<const v4i *>(ps.pk))
    switch ((SweepKind)ps.kind) {
    case SWEEP_FWD_LAST_ASC: UNPACK(SWEEP_FWD_LAST_ASC); break;
    case SWEEP_BWD_FIRST_ASC: UNPACK(SWEEP_BWD_FIRST_ASC); break;
    default: UNPACK(SWEEP_BWD_FIRST_DESC); break;
    }
#undef UNPACK
    ILUPP_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
int ilu0_numeric_lm(hipStream_t st, const DevMat &A, const Schedule &fwd, PackedSweep *pl,
                    PackedSweep *pu, FactorLM *f, int32_t *d_ctrl, float *kernel_ms, hipEvent_t e0, hipEvent_t e1)
{
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    hipLaunchKernelGGL(k_flm_fill, dim3(512), dim3(256), 0, st, reinterpret_cast<unsigned long long *>(f->xch),
                       reinterpret_cast<const long long *>(f->xcount), kSentinel);
    const dim3 gridr((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));   // every lane has at most max_chunks rows
    if (!f->values_packed)
        hipLaunchKernelGGL(k_flm_pack_a, gridr, dim3(512), 0, st, A.ptr, A.val, (int64_t)A.nnz, pl->wtab, pl->skew, fwd.sfirst, fwd.scount,
                           reinterpret_cast<v4i *>(f->pkA));
    f->values_packed = false;
    FlmArgs a;
    a.pkL_in = reinterpret_cast<const v4i *>(pl->pk); a.pkL_out = reinterpret_cast<v2d *>(pl->pk);
    a.pkA = reinterpret_cast<const v4i *>(f->pkA);
    a.pkU_out = reinterpret_cast<v2d *>(pu->pk);
    a.wtabL = pl->wtab; a.skewL = pl->skew; a.wtabU = pu->wtab; a.skewU = pu->skew; a.uslot = pu->uslot;
    a.sfirst = fwd.sfirst; a.scount = fwd.scount; a.exported = fwd.exported; a.gtab = fwd.gtab; a.xbase = f->xbase;
    a.xch = f->xch; a.nslots_used = fwd.nslots; a.ctrl = d_ctrl;
#ifdef ILUPP_TIMELINE
    static unsigned long long *d_tl = nullptr;
    if (!d_tl) {
        ILUPP_HIP(hipMalloc(&d_tl, 8 * 8 * 4096));
        ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_flm_timeline), &d_tl, sizeof(d_tl)));
    }
    ILUPP_HIP(hipMemsetAsync(d_tl, 0, 8 * 8 * 4096, st));
#endif
    ILUPP_HIP(hipEventRecord(e0, st));
    {
        // once per device (the attribute is a property of the function on a device)
        static std::once_flag once[64];
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_lm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFlmLds));
        });
    }
    hipLaunchKernelGGL(k_ilu0_lm, dim3((unsigned)pl->nwg), dim3(kFlmThreads), kFlmLds, st, a);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4];
    ILUPP_HIP(d2h_async(st, ctrl, d_ctrl, 16));
    ILUPP_HIP(stream_sync(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
#ifdef ILUPP_TIMELINE
    if (pl->nwg <= 4096) {
        static unsigned long long h_tl[8 * 4096];
        ILUPP_HIP(hipMemcpy(h_tl, d_tl, 8 * 8 * (size_t)pl->nwg, hipMemcpyDeviceToHost));
        if (FILE *ftl = fopen("/tmp/timeline_factor.bin", "wb")) { fwrite(h_tl, 8, 8 * (size_t)pl->nwg, ftl); fclose(ftl); }
    }
#endif
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

}  // namespace ilupp
