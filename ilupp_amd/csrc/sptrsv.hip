// ilupp_amd/csrc/sptrsv.hip -- sparse triangular solves of apply() for gfx950.
//
// Replaces matrix_sparse::triangular_solve (reference sparse_implementation.h:4040-4087).  The four
// loops of the reference reduce to three gather sweeps over row-major storage (the two scatter loops
// T2/T4 perform, for every unknown, the same subtractions in the same order as a gather over the
// transposed storage; see DESIGN.md "Solve variants"):
//
//   SWEEP_FWD_LAST_ASC    rows ascending,  x_r = (x_r - sum_{j<last} v_j x[c_j]) / v_last     (T1, T2')
//   SWEEP_BWD_FIRST_ASC   rows descending, x_r = (x_r - sum_{j>first} v_j x[c_j]) / v_first   (T3)
//   SWEEP_BWD_FIRST_DESC  the same with the off-diagonal entries taken last-to-first          (T4')
//
// The accumulation is sequential in stored order with separate multiply and subtract, and the
// diagonal is found by POSITION and always divided by (unit diagonals too), exactly as the
// reference does, so solve vectors are bit-identical to the CPU.
//
// One persistent launch per sweep; hand-off between lanes is "the data is the flag": the output
// vector starts as all-sentinel (a NaN payload no arithmetic produces), a finished x_r is stored
// with one 8-byte write-through (sc1) store, consumers poll x[c] with sc1 loads until it is not the
// sentinel.  No flag array, no fences, no grid barrier.  The right-hand side is read once per row
// and immediately overwritten with the sentinel, which makes that buffer the ready-made output of
// the next sweep: L-solve  x -> y,  U-solve  y -> x  leaves the result in place in x and y reset.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace ilupp {

static constexpr unsigned kSolveSpinLimit = 1u << 22;

template <int KIND>
__global__ void __launch_bounds__(kThreads)
k_sptrsv(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
         double *rhs, double *out, int32_t nb, const int32_t *__restrict__ bstart, int32_t *ticket, int32_t *err)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr bool DESC = (KIND == SWEEP_BWD_FIRST_DESC);
    __shared__ unsigned wg_ticket;
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(ticket, 1);
    __syncthreads();
    const int tid = threadIdx.x;
    const int64_t slot = (int64_t)wg_ticket * kThreads + tid;

    // rows of this lane in sweep order: r, r+dir, ..., until r == rstop
    int r = 0, rstop = 0;
    if (slot < nb) {
        const int b = FWD ? (int)slot : (int)(nb - 1 - slot);
        const int lo = bstart[b], hi = bstart[b + 1];
        if (FWD) { r = lo; rstop = hi; } else { r = hi - 1; rstop = lo - 1; }
    }
    constexpr int dir = FWD ? 1 : -1;
    bool active = (r != rstop);
    bool need_init = true;
    int j = 0, jend = 0, dpos = 0;
    double acc = 0.0, prev_val = 0.0;
    int prev_row = -1;
    unsigned spins = 0;
    const unsigned long long *outb = reinterpret_cast<const unsigned long long *>(out);

    for (;;) {
        if (!__any(active)) break;
        bool progressed = false;
        if (active) {
            if (need_init) {
                const int lo = ptr[r], hi = ptr[r + 1];
                acc = rhs[r];
                reinterpret_cast<unsigned long long *>(rhs)[r] = kSentinel;   // this buffer is the next sweep's output
                if (FWD)       { j = lo;     jend = hi - 1; dpos = hi - 1; }
                else if (!DESC){ j = lo + 1; jend = hi;     dpos = lo; }
                else           { j = hi - 1; jend = lo;     dpos = lo; }
                need_init = false;
                progressed = true;
            }
            while (j != jend) {
                const int c = idx[j];
                double xc;
                if (c == prev_row) {
                    xc = prev_val;                      // own previous row: never leaves the lane
                } else {
                    const unsigned long long bits = ld_agent_u64(outb + c);
                    if (bits == kSentinel) break;       // x[c] not there yet: retry next round
                    xc = __longlong_as_double((long long)bits);
                }
                const double prod = val[j] * xc;
                acc = acc - prod;                       // x[k] -= data[j]*x[indices[j]]  (:4049, :4070)
                j += DESC ? -1 : 1;
                progressed = true;
            }
            if (j == jend) {
                acc = acc / val[dpos];                  // x[k] /= diagonal (by position)  (:4051, :4072)
                if (acc != acc) acc = __longlong_as_double((long long)kCanonNaN);   // never store the sentinel
                st_agent_f64(out + r, acc);
                prev_row = r;
                prev_val = acc;
                r += dir;
                need_init = true;
                active = (r != rstop);
                progressed = true;
            }
        }
        if (__any(progressed)) {
            spins = 0;
        } else {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSolveSpinLimit) {
                if ((tid & 63) == 0) atomicExch(err, 1);
                break;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// one lane per ROW (factors whose rows are too long for the level-major records and have no chains worth a lane:
// ILUT factors, ICholT with fill).  A workgroup holds kRowsBlock consecutive rows of the sweep; a dependency on a row of
// the same workgroup is read from an LDS copy of the workgroup's unknowns (sentinel = not yet) instead of through
// memory: on mesh-like factors consecutive rows ARE a chain and the next grid line is a few hundred rows away; a link
// through the write-through store / cache-bypassing poll costs ~3 us, a link through LDS ~0.1 us.  Everything else as in
// k_sptrsv: sequential accumulation in stored order, diagonal by position, data-is-flag on the output vector, block
// ids from an atomic ticket (a row only polls rows of earlier tickets or earlier lanes of its own workgroup).
// ---------------------------------------------------------------------------------------------
#ifndef ROWS_W
#define ROWS_W 8
#endif
#ifndef ROWS_BLOCK
#define ROWS_BLOCK 1024
#endif
static constexpr int kRowsBlock = ROWS_BLOCK;

template <int KIND>
__global__ void __launch_bounds__(kRowsBlock)
k_sptrsv_rows(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
              double *rhs, double *out, int32_t *ticket, int32_t *err)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr bool DESC = (KIND == SWEEP_BWD_FIRST_DESC);
    constexpr int dj = DESC ? -1 : 1;
    constexpr int W = ROWS_W;                                        // dependencies fetched per round trip
    __shared__ unsigned wg_ticket;
    __shared__ unsigned long long xs[kRowsBlock];                    // this workgroup's unknowns, sentinel = not yet
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(ticket, 1);
    xs[threadIdx.x] = kSentinel;
    __syncthreads();
    const int64_t tb = (int64_t)wg_ticket * kRowsBlock;              // position of the workgroup's first row in sweep order
    const int64_t t = tb + threadIdx.x;
    bool active = t < n;
    const int r = active ? (FWD ? (int)t : (int)(n - 1 - t)) : 0;
    const long row0 = FWD ? (long)tb : (long)n - 1 - (long)tb;       // row at position tb
    int j = 0, jend = 0;
    double acc = 0.0, dv = 1.0;
    if (active) {
        const int lo = ptr[r], hi = ptr[r + 1];
        acc = rhs[r];
        reinterpret_cast<unsigned long long *>(rhs)[r] = kSentinel;   // this buffer is the next sweep's output
        int dpos;
        if (FWD)        { j = lo;     jend = hi - 1; dpos = hi - 1; }
        else if (!DESC) { j = lo + 1; jend = hi;     dpos = lo; }
        else            { j = hi - 1; jend = lo;     dpos = lo; }
        if (hi > lo) dv = val[dpos];
        else { j = jend = lo; dv = __longlong_as_double((long long)kCanonNaN); }   // a row without entries (factors with NaN columns):
                                                                                    // the reference divides by whatever sits next to it
    }
    const unsigned long long *outb = reinterpret_cast<const unsigned long long *>(out);
    // window of the next W entries: columns, values, and what the output vector held when the window was fetched.  The values
    // are applied strictly in stored order; only the FETCHES run ahead (a backward sweep in ascending stored order meets its
    // nearest -- latest -- dependency first and would otherwise pay one cache-bypassing round trip per remaining entry
    // after it has arrived)
    int wc[W];
    double wv[W];
    unsigned long long wb[W];
    int wn = 0, cur = 0;
#pragma unroll
    for (int u = 0; u < W; ++u) { wc[u] = -1; wv[u] = 0.0; wb[u] = kSentinel; }
    unsigned spins = 0;
    for (;;) {
        if (__ballot(active) == 0ull) break;
        bool progressed = false;
        if (active && cur == wn && j != jend) {
            const int left = DESC ? j - jend : jend - j;
            wn = left < W ? left : W;
            cur = 0;
#pragma unroll
            for (int u = 0; u < W; ++u) if (u < wn) { wc[u] = idx[j + u * dj]; wv[u] = val[j + u * dj]; }
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const long lb_u = FWD ? (long)wc[u] - row0 : row0 - (long)wc[u];
                wb[u] = (u < wn && (unsigned long)lb_u >= (unsigned long)kRowsBlock) ? ld_agent_u64(outb + wc[u]) : kSentinel;
            }
            progressed = true;
        }
        if (active) {
            if (cur < wn) {
                // everything of the window that is there, in stored order, in this round: after a blocker has arrived (in a
                // backward sweep in ascending order that is the FIRST entry: the chain neighbour) the rest must not cost one
                // round each -- the rounds of a chain add up
                // (sweeps that meet their nearest dependency LAST -- forward, or backward in descending order -- take one entry
                // per round: the heavier round costs them more than it saves)
                constexpr bool MULTI = (KIND == SWEEP_BWD_FIRST_ASC);
                asm volatile("" ::: "memory");
                if constexpr (MULTI) {
                    bool stop = false;
#pragma unroll
                    for (int u = 0; u < W; ++u) {
                        if (!stop && u >= cur && u < wn) {
                            const int c = wc[u];
                            const long lb = FWD ? (long)c - row0 : row0 - (long)c;   // place of row c in this workgroup, if it is one of ours
                            unsigned long long b = wb[u];
                            if ((unsigned long)lb < (unsigned long)kRowsBlock) b = xs[lb];
                            else if (b == kSentinel) b = ld_agent_u64(outb + c);       // was not there when the window was fetched: poll this one only
                            if (b != kSentinel) {
                                const double prod = wv[u] * __longlong_as_double((long long)b);
                                acc = acc - prod;               // x[k] -= data[j]*x[indices[j]]  (:4049, :4070)
                                j += dj;
                                ++cur;
                                progressed = true;
                            } else {
                                stop = true;
                            }
                        }
                    }
                } else {
                    int c = -1; double v = 0.0; unsigned long long b = kSentinel;
#pragma unroll
                    for (int u = 0; u < W; ++u) if (cur == u) { c = wc[u]; v = wv[u]; b = wb[u]; }
                    const long lb = FWD ? (long)c - row0 : row0 - (long)c;
                    if ((unsigned long)lb < (unsigned long)kRowsBlock) b = xs[lb];
                    else if (b == kSentinel) b = ld_agent_u64(outb + c);
                    if (b != kSentinel) {
                        const double prod = v * __longlong_as_double((long long)b);
                        acc = acc - prod;                       // x[k] -= data[j]*x[indices[j]]  (:4049, :4070)
                        j += dj;
                        ++cur;
                        progressed = true;
                    }
                }
            }
            if (j == jend) {
                double x = acc / dv;                        // x[k] /= diagonal (by position)  (:4051, :4072)
                if (x != x) x = __longlong_as_double((long long)kCanonNaN);   // never store the sentinel
                st_agent_f64(out + r, x);
                xs[threadIdx.x] = (unsigned long long)__double_as_longlong(x);
                asm volatile("" ::: "memory");
                active = false;
                progressed = true;
            }
        }
        if (__any(progressed)) {
            spins = 0;
        } else {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSolveSpinLimit) {
                if ((threadIdx.x & 63) == 0) atomicExch(err, 1);
                break;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// descriptor-driven sweep: the production kernel
// ---------------------------------------------------------------------------------------------
// Same arithmetic; the column index of every stored entry is replaced by its solve descriptor
// (schedule.hip), so a lane knows without any lookup whether the unknown it needs is produced by
// its own workgroup.  If so it comes through an LDS ring (slot = kloc mod D, seqlock tag), ~100
// cycles; otherwise through the data-is-flag poll on the output vector in HBM.
static constexpr int kXRing = 8;

template <int KIND>
__global__ void __launch_bounds__(kThreads)
k_sptrsv_desc(const int32_t *__restrict__ ptr, const int32_t *__restrict__ desc, const double *__restrict__ val,
              double *rhs, double *out, int32_t nslots_used, const int32_t *__restrict__ sfirst,
              const int32_t *__restrict__ scount, const int32_t *__restrict__ gtab, int32_t *ticket, int32_t *err)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr bool DESC = (KIND == SWEEP_BWD_FIRST_DESC);
    constexpr int D = kXRing;
    __shared__ __attribute__((aligned(16))) double xring_s[D * kThreads];
    __shared__ int tag_s[D * kThreads];
    __shared__ unsigned wg_ticket;
    volatile double *xring = xring_s;
    volatile int *tag = tag_s;
    const int tid = threadIdx.x;
    if (tid == 0) wg_ticket = (unsigned)atomicAdd(ticket, 1);
    for (int s = 0; s < D; ++s) tag[s * kThreads + tid] = -1;
    __syncthreads();
    const unsigned wg = wg_ticket;
    const unsigned myslot = wg * kThreads + tid;

    int cnt = 0, r = 0;
    if ((int)myslot < nslots_used) { cnt = scount[myslot]; r = sfirst[myslot]; }
    constexpr int dir = FWD ? 1 : -1;
    bool active = cnt > 0;
    bool need_init = true;
    int rloc = 0;
    int j = 0, jend = 0, dpos = 0;
    double acc = 0.0, prev_val = 0.0;
    unsigned spins = 0;
    const unsigned long long *outb = reinterpret_cast<const unsigned long long *>(out);

    for (;;) {
        if (!__any(active)) break;
        bool progressed = false;
        if (active) {
            if (need_init) {
                const int lo = ptr[r], hi = ptr[r + 1];
                acc = rhs[r];
                reinterpret_cast<unsigned long long *>(rhs)[r] = kSentinel;
                if (FWD)       { j = lo;     jend = hi - 1; dpos = hi - 1; }
                else if (!DESC){ j = lo + 1; jend = hi;     dpos = lo; }
                else           { j = hi - 1; jend = lo;     dpos = lo; }
                need_init = false;
                progressed = true;
            }
            while (j != jend) {
                const unsigned d = (unsigned)desc[j];
                const unsigned oslot = d >> 15;
                const int kl = (int)(d & 0x7fffu);
                double xc;
                bool have = false;
                if (oslot == myslot && kl == rloc - 1) {
                    xc = prev_val;                                   // own previous row
                    have = true;
                } else if ((oslot >> 8) == wg) {
                    const int lane = (int)(oslot & 255u);
                    const int s = kl & (D - 1);
                    const int t1 = tag[s * kThreads + lane];
                    if (t1 == kl) {
                        xc = xring[s * kThreads + lane];
                        have = (tag[s * kThreads + lane] == kl);     // slot not recycled while we read
                    } else if (t1 < kl) {
                        break;                                       // producer not there yet
                    }
                }
                if (!have) {
                    // a ghost owner names an entry of this workgroup's import table (schedule.hip)
                    const unsigned os = oslot >= (unsigned)kGhostBase ? (unsigned)gtab[wg * kGhosts + (oslot & (kGhosts - 1))] : oslot;
                    const int c = sfirst[os] + dir * kl;
                    const unsigned long long bits = ld_agent_u64(outb + c);
                    if (bits == kSentinel) break;
                    xc = __longlong_as_double((long long)bits);
                }
                const double prod = val[j] * xc;
                acc = acc - prod;
                j += DESC ? -1 : 1;
                progressed = true;
            }
            if (j == jend) {
                acc = acc / val[dpos];
                if (acc != acc) acc = __longlong_as_double((long long)kCanonNaN);
                const int s = rloc & (D - 1);
                tag[s * kThreads + tid] = -1;
                xring[s * kThreads + tid] = acc;
                tag[s * kThreads + tid] = rloc;
                st_agent_f64(out + r, acc);
                prev_val = acc;
                r += dir;
                ++rloc;
                need_init = true;
                active = rloc < cnt;
                progressed = true;
            }
        }
        if (__any(progressed)) {
            spins = 0;
        } else {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSolveSpinLimit) {
                if ((tid & 63) == 0) atomicExch(err, 1);
                break;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// loader/consumer sweep: the production solve kernel for short-row factors
// ---------------------------------------------------------------------------------------------
// The sweeps are latency-bound: ~22 k independent rows per dependency level over 256 CUs leave each
// CU a few waves, and gfx950 returns a wave's vector-memory operations IN ORDER (one vmcnt queue), so
// a wave that prefetches cannot also poll cheaply.  The workgroup therefore splits roles:
//   * waves 4-7 are LOADERS: loader lane t streams the four input streams of consumer lane t (row
//     pointers, right-hand side, descriptors, values; all contiguous because the lane owns consecutive
//     rows) a few rows ahead with independent 16-byte loads and drops them into per-lane LDS rings
//     ([slot][lane] layout: the bank depends on the lane only, so any per-lane offset is conflict-free);
//     only loaders ever wait on HBM latency;
//   * waves 0-3 are CONSUMERS: they read LDS only.  Unknowns produced in the same workgroup arrive
//     through a 16-byte {tag,x} LDS ring entry (one ds_read_b128); results leave by write-through stores;
//   * wave 8 is the IMPORTER: lane g follows one foreign producer lane (a "ghost", schedule.hip gtab): it
//     polls that lane's unknowns in HBM (data-is-flag) a few rows ahead of their consumer and drops them
//     into a ghost {tag,x} ring, so a consumer reads a foreign unknown exactly like a local one and never
//     waits on a memory round trip (a poll by one lane stalls all 64 lanes of its wave).  The ghost ring is
//     only a fast path: an entry that is late or already recycled is fetched by the consumer itself.
struct __attribute__((aligned(4))) I4u { int v[4]; };
struct __attribute__((aligned(8))) D2u { double v[2]; };
typedef int v4i __attribute__((ext_vector_type(4)));

// Branch-free window loads: a 16-byte load whose base is clamped into the array; the elements are
// dropped at their TRUE absolute ring positions (clamp_base(...) + k), so array ends need neither
// padding nor a scalar fallback (a fallback branch would make hipcc wait for every load in turn).
__device__ __forceinline__ long clamp_base(long i, long n, int w) { return i < 0 ? 0 : (i > n - w ? n - w : i); }

static constexpr int kEW = 32, kEQ = 8, kENQ = 3;   // entry ring / refill quantum / quanta per loader round
static constexpr int kRW = 16, kRQ = 4, kRNQ = 3;   // right-hand-side ring / quantum / quanta per round
static constexpr int kXD = 4;              // depth of the {tag,x} hand-off ring
static constexpr int kEC = 4;              // external unknowns fetched per trip
static constexpr int kGD = 8;              // depth of a ghost ring (overlays the external cache: kGD*kGhosts*16 == kEC*256*8)
static constexpr int kIB = 4;              // unknowns an importer lane polls per trip
#ifndef GP
#define GP 256
#endif
static constexpr int kGhostPatience = GP;  // rounds a consumer waits for a ghost entry before polling itself
static constexpr int kLcThreads = 2 * kThreads + 64;
static constexpr size_t kLcLds = (size_t)kThreads * (kEW * 4 + kEW * 8 + kRW * 8 + kXD * 16 + kEC * 8 + 5 * 4) + 2 * kGhosts * 4 + 32;
static_assert(kGD * kGhosts * 16 == kEC * kThreads * 8, "ghost rings overlay the external cache");
static constexpr int kDiag = -1;           // descriptor of a diagonal entry (schedule.hip); delimits rows in the stream

#ifdef ILUPP_TIMELINE
__device__ unsigned long long *g_timeline = nullptr;     // diagnostics build only: 8 words per workgroup
#define TL(i) do { if (tl) tl[(size_t)wg * 8 + (i)] = wall_clock64(); } while (0)
#else
#define TL(i) do { } while (0)
#endif

template <int KIND>
__global__ void __launch_bounds__(kLcThreads)
k_sptrsv_lc(const int32_t *__restrict__ ptr, const int32_t *__restrict__ desc, const double *__restrict__ val,
            int32_t n, long nnz, double *rhs, double *out, int32_t nslots_used,
            const int32_t *__restrict__ sfirst, const int32_t *__restrict__ scount,
            const int32_t *__restrict__ exported, const int32_t *__restrict__ gtab, int32_t *ticket, int32_t *err)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr bool DESC = (KIND == SWEEP_BWD_FIRST_DESC);
    constexpr int DR = FWD ? 1 : -1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x & (kThreads - 1);        // consumer lane / client lane
    const bool is_consumer = threadIdx.x < kThreads;
    const bool is_loader = !is_consumer && threadIdx.x < 2 * kThreads;
    const bool is_importer = threadIdx.x >= 2 * kThreads;
    // carve (every offset a multiple of 16)
    double *sval = reinterpret_cast<double *>(smem);                         // [kEW][256]
    double *srhs = sval + kEW * kThreads;                                    // [kRW][256]
    double *secv = srhs + kRW * kThreads;                                    // [kEC][256]
    v4i *xr = reinterpret_cast<v4i *>(secv + kEC * kThreads);                // [kXD][256] {tag,-,x.lo,x.hi}
    int *sdesc = reinterpret_cast<int *>(xr + kXD * kThreads);               // [kEW][256]
    // hand-shake words (plain LDS accesses; ordering comes from the in-order LDS queue of each wave plus
    // compiler barriers -- volatile would demote them to flat, system-scope accesses)
    int *e_avail = sdesc + kEW * kThreads;                                   // [256] each
    int *r_avail = e_avail + kThreads;
    int *e_cons = r_avail + kThreads;
    int *r_cons = e_cons + kThreads;
    int *fin = r_cons + kThreads;
    int *gfirst = fin + kThreads;                                            // [kGhosts] first row of the ghost's producer
    int *gack = gfirst + kGhosts;                                            // [kGhosts] oldest entry still wanted by the consumer
    int *hasg = gack + kGhosts;                                              // workgroup imports through ghosts
    unsigned *wg_ticket = reinterpret_cast<unsigned *>(hasg + 4);
    v4i *gr = reinterpret_cast<v4i *>(secv);                                 // [kGD][kGhosts] {tag,-,x.lo,x.hi}, ghost mode only
    if (threadIdx.x == 0) *wg_ticket = (unsigned)atomicAdd(ticket, 1);
    __syncthreads();
    const unsigned wg = *wg_ticket;
    const unsigned myslot = wg * kThreads + tid;
#ifdef ILUPP_TIMELINE
    unsigned long long *const tl = g_timeline;
#endif
    if (threadIdx.x == 0) TL(0);

#define RD(i) sdesc[((i) & (kEW - 1)) * kThreads + tid]
#define RV(i) sval[((i) & (kEW - 1)) * kThreads + tid]
#define RR(i) srhs[((i) & (kRW - 1)) * kThreads + tid]

    int cnt = 0, r0 = 0;
    bool exports = true;          // results read by another workgroup must leave the XCD (write-through); the rest may stay in L2
    if ((int)myslot < nslots_used) { cnt = scount[myslot]; r0 = sfirst[myslot]; exports = exported[myslot] != 0; }
    int bound0 = 0;          // FWD: start of the first row; BWD: end of the first row
    if (cnt > 0) bound0 = FWD ? ptr[r0] : ptr[r0 + 1];
    int g_first = 0, g_cnt = 0;
    if (is_importer) {
        const int g = threadIdx.x - 2 * kThreads;
        const int os = gtab ? gtab[(size_t)wg * kGhosts + g] : -1;
        if (os >= 0) { g_first = sfirst[os]; g_cnt = scount[os]; }
        const bool any = __any(os >= 0);
        if (g == 0) *hasg = any ? 1 : 0;
        if (any) for (int s = 0; s < kGD; ++s) { v4i e; e.x = -1; e.y = 0; e.z = 0; e.w = 0; gr[s * kGhosts + g] = e; }
        gfirst[g] = g_first; gack[g] = 0;
    }
    if (is_consumer) {
        for (int s = 0; s < kXD; ++s) { v4i e; e.x = -1; e.y = 0; e.z = 0; e.w = 0; xr[s * kThreads + tid] = e; }
        e_avail[tid] = bound0; r_avail[tid] = FWD ? r0 : r0 + 1;
        e_cons[tid] = bound0;  r_cons[tid] = r0;
        fin[tid] = cnt > 0 ? 0 : 1;
    }
    // retire the set-up loads with a wait hipcc can see: a load still pending at the loop header would
    // make it re-wait (vmcnt(0), i.e. also for our own write-through stores) on every round
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    if (is_loader) {
        // ------------------------------------------------------------------ loader
        // every round: issue ALL quanta the rings have room for (independent 16-byte loads), one wait,
        // drop them into LDS, publish.  Only this wave ever waits on HBM latency.
        int e_next = bound0;                 // FWD: next entry to load; BWD: one past the next quantum
        int r_next = FWD ? r0 : r0 + 1;      // FWD: next row to load;   BWD: one past the next quantum
        bool live = cnt > 0;
        unsigned idle = 0;
        for (;;) {
            if (!__any(live)) break;
            asm volatile("" ::: "memory");
            bool did = false;
            if (live) {
                if (fin[tid]) {
                    live = false;
                } else {
                    const int ec = e_cons[tid], rc = r_cons[tid];
                    bool te[kENQ], tr[kRNQ];
                    long bd[kENQ][kEQ / 4], bv[kENQ][kEQ / 2], br[kRNQ][kRQ / 2];
                    I4u td[kENQ][kEQ / 4];
                    D2u tv[kENQ][kEQ / 2], tw[kRNQ][kRQ / 2];
#pragma unroll
                    for (int u = 0; u < kENQ; ++u) {
                        const int en = e_next + (FWD ? u * kEQ : -u * kEQ);
                        te[u] = FWD ? (en + kEQ <= ec + kEW) : (en - kEQ >= ec - kEW);
                        const int base = FWD ? en : en - kEQ;
                        if (te[u]) {
#pragma unroll
                            for (int q = 0; q < kEQ / 4; ++q) { bd[u][q] = clamp_base((long)base + 4 * q, nnz, 4); td[u][q] = *reinterpret_cast<const I4u *>(desc + bd[u][q]); }
#pragma unroll
                            for (int q = 0; q < kEQ / 2; ++q) { bv[u][q] = clamp_base((long)base + 2 * q, nnz, 2); tv[u][q] = *reinterpret_cast<const D2u *>(val + bv[u][q]); }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kRNQ; ++u) {
                        const int rn = r_next + (FWD ? u * kRQ : -u * kRQ);
                        tr[u] = FWD ? (rn + kRQ <= rc + kRW) : (rn - kRQ >= rc + 1 - kRW);
                        const int base = FWD ? rn : rn - kRQ;
                        if (tr[u]) {
#pragma unroll
                            for (int q = 0; q < kRQ / 2; ++q) { br[u][q] = clamp_base((long)base + 2 * q, n, 2); tw[u][q] = *reinterpret_cast<const D2u *>(rhs + br[u][q]); }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kENQ; ++u) {
                        if (te[u]) {
#pragma unroll
                            for (int q = 0; q < kEQ / 4; ++q)
#pragma unroll
                                for (int k = 0; k < 4; ++k) RD((int)bd[u][q] + k) = td[u][q].v[k];
#pragma unroll
                            for (int q = 0; q < kEQ / 2; ++q)
#pragma unroll
                                for (int k = 0; k < 2; ++k) RV((int)bv[u][q] + k) = tv[u][q].v[k];
                            e_next += FWD ? kEQ : -kEQ;
                            did = true;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kRNQ; ++u) {
                        if (tr[u]) {
#pragma unroll
                            for (int q = 0; q < kRQ / 2; ++q)
#pragma unroll
                                for (int k = 0; k < 2; ++k) RR((int)br[u][q] + k) = tw[u][q].v[k];
                            r_next += FWD ? kRQ : -kRQ;
                            did = true;
                        }
                    }
                    asm volatile("" ::: "memory");
                    e_avail[tid] = e_next;
                    r_avail[tid] = r_next;
                }
            }
            if (__any(did)) {
                idle = 0;
            } else {
                __builtin_amdgcn_s_sleep(2);
                if (++idle > kSolveSpinLimit) break;     // consumers report the time-out
            }
        }
        return;
    }

    const unsigned long long *outb = reinterpret_cast<const unsigned long long *>(out);
    if (is_importer) {
        // ---------------------------------------------------------------- importer
        if (!*hasg) return;
        const int g = threadIdx.x - 2 * kThreads;
        int next = 0;
        unsigned idle = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            const v4i f = reinterpret_cast<const v4i *>(fin)[g];
            if (__all((f.x & f.y & f.z & f.w) != 0)) break;             // every consumer lane is done
            bool did = false;
            if (next < g_cnt) {
                const int ack = gack[g];
                if (ack > next) next = ack;                              // the consumer fetched those itself
                int nb = g_cnt - next;
                nb = nb < kIB ? nb : kIB;
                nb = nb < ack + kGD - next ? nb : ack + kGD - next;      // entry k recycles the slot of k-kGD
                if (nb > 0) {
                    unsigned long long b[kIB];
#pragma unroll
                    for (int q = 0; q < kIB; ++q) {
                        const int cq = g_first + DR * (next + q);
                        b[q] = ld_agent_u64(outb + (cq < 0 ? 0 : (cq >= n ? n - 1 : cq)));
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    int got = 0;
#pragma unroll
                    for (int q = 0; q < kIB; ++q) got += (got == q && q < nb && b[q] != kSentinel) ? 1 : 0;
#pragma unroll
                    for (int q = 0; q < kIB; ++q) {
                        if (q < got) {
                            v4i e;
                            e.x = next + q; e.y = 0; e.z = (int)(unsigned)b[q]; e.w = (int)(unsigned)(b[q] >> 32);
                            gr[((next + q) & (kGD - 1)) * kGhosts + g] = e;
                        }
                    }
                    next += got;
                    did = got > 0;
                }
            }
            if (__any(did)) {
                idle = 0;
            } else {
                __builtin_amdgcn_s_sleep(1);
                if (++idle > kSolveSpinLimit) break;
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    // A row costs TWO dependent LDS round trips when it has at most 4 stored entries (the 5-/7-point
    // factors): round 1 reads the hand-shake words, the right-hand side and -- from the already known row
    // start -- four descriptor/value pairs speculatively (the diagonal's marker descriptor delimits the
    // row, so no row-pointer stream exists); round 2 reads the {tag,x} ring entries of all its
    // in-workgroup dependencies at once.  Longer rows take the entry loop.
    int r = r0;
    bool active = cnt > 0;
    int rloc = 0;
    int bound = bound0;
    int phase = 0;            // 0: fetch the row; 1: short row, waiting for dependencies; 2: long row, entry loop
    int j = 0, jend = 0, dpos = 0, lo = 0, hi = 0, len = 0;
    int ed[4] = {0, 0, 0, 0};
    double ev[4] = {0.0, 0.0, 0.0, 0.0};
    double acc = 0.0, prev_val = 0.0;
    unsigned ec_oslot = 0xffffffffu; int ec_kl0 = 0, ec_cnt = 0, ec_first = 0;
    unsigned spins = 0;
    int stall = 0;                         // rounds the current row has been waiting for its dependencies
    int gpat = kGhostPatience;
    const bool ghost_mode = *hasg != 0;    // the external cache's LDS then belongs to the ghost rings

    // external unknown (other workgroup): cached batch of kEC consecutive ones, else one poll trip
    auto poll_one = [&](int c, double &xc) -> bool {
        const unsigned long long b = ld_agent_u64(outb + c);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        xc = __longlong_as_double((long long)b);
        return b != kSentinel;
    };
    auto external = [&](unsigned oslot, int kl, double &xc) -> bool {
        if (ghost_mode) {
            if (oslot != ec_oslot) { ec_first = sfirst[oslot]; ec_oslot = oslot; __builtin_amdgcn_s_waitcnt(0x0F70); }
            return poll_one(ec_first + DR * kl, xc);
        }
        if (oslot == ec_oslot && kl >= ec_kl0 && kl < ec_kl0 + ec_cnt) {
            xc = secv[(kl - ec_kl0) * kThreads + tid];
            return true;
        }
        if (oslot != ec_oslot) { ec_first = sfirst[oslot]; ec_oslot = oslot; ec_cnt = 0; __builtin_amdgcn_s_waitcnt(0x0F70); }
        const int c = ec_first + DR * kl;
        unsigned long long b[kEC];
#pragma unroll
        for (int q = 0; q < kEC; ++q) {
            const int cq = c + DR * q;
            b[q] = ld_agent_u64(outb + (cq < 0 ? 0 : (cq >= n ? n - 1 : cq)));
        }
        // retire the polls HERE: otherwise hipcc parks a vmcnt(0) at the join, where it would also wait
        // for the previous row's write-through store on every row
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (b[0] == kSentinel) { ec_cnt = 0; return false; }
        int got = 1;
#pragma unroll
        for (int q = 1; q < kEC; ++q) got += (got == q && b[q] != kSentinel && c + DR * q >= 0 && c + DR * q < n) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < kEC; ++q) secv[q * kThreads + tid] = __longlong_as_double((long long)b[q]);
        ec_kl0 = kl; ec_cnt = got;
        xc = __longlong_as_double((long long)b[0]);
        return true;
    };
    // unknown kl of ghost lane g; e is its ring entry read this round
    auto ghost = [&](int g, int kl, const v4i &e, double &xc) -> bool {
        bool have = false;
        if (e.x == kl) {
            xc = __hiloint2double(e.w, e.z);
            have = true;
        } else if (e.x > kl || stall > gpat) {
            // recycled before we came, or nothing arrived for a long time: fetch it here
            have = poll_one(gfirst[g] + DR * kl, xc);
            if (e.x < kl) { if (have) gpat = 0; else stall = 0; }     // the ring failed us once: stop relying on it
        }
        gack[g] = kl;              // entries older than the one we want may be recycled; the importer skips to it
        return have;
    };
    auto publish = [&](double x) {
        if (x != x) x = __longlong_as_double((long long)kCanonNaN);
        v4i e;
        e.x = rloc; e.y = 0; e.z = __double2loint(x); e.w = __double2hiint(x);
        xr[(rloc & (kXD - 1)) * kThreads + tid] = e;                  // one ds_write_b128
        asm volatile("" ::: "memory");
#ifdef EXP_NOSTORE
        if (exports) st_agent_f64(out + r, x);
#else
        if (exports) st_agent_f64(out + r, x); else out[r] = x;
#endif
#ifdef ILUPP_TIMELINE
        if (rloc == 0 || rloc == cnt - 1 || rloc == cnt / 2) {
            if (tid == 0) { if (rloc == 0) TL(1); else if (rloc == cnt - 1) TL(2); else TL(7); }
            if (tid == 255) { if (rloc == 0) TL(3); if (rloc == cnt - 1) TL(4); }
            if (tid == 15 && rloc == 0) TL(5);
            if (tid == 240 && rloc == 0) TL(6);
        }
#endif
        prev_val = x;
        bound = FWD ? hi : lo;
        r += DR;
        ++rloc;
        phase = 0;
        stall = 0;
        active = rloc < cnt;
        if (!active) fin[tid] = 1;
    };

    for (;;) {
        if (!__any(active)) break;
        asm volatile("" ::: "memory");     // LDS written by other waves is re-read every round
        bool progressed = false;
        if (active) {
            if (phase == 0) {
                // round 1: everything addressable from registers, issued together
                const int ebase = FWD ? bound : bound - 4;
                const int ra = r_avail[tid];
                const int ea = e_avail[tid];
                asm volatile("" ::: "memory");      // hand-shake words are read BEFORE the data they guard
                const double rr = RR(r);
#pragma unroll
                for (int k = 0; k < 4; ++k) { ed[k] = RD(ebase + k); ev[k] = RV(ebase + k); }
                asm volatile("" ::: "memory");      // ... and the data BEFORE the words that release its slots
                if (FWD ? (r < ra) : (r >= ra)) {
                    // locate the diagonal marker among the entries that are really loaded
                    int dk = -1;
                    if (FWD) {
#pragma unroll
                        for (int k = 3; k >= 0; --k) if (ed[k] == kDiag && ebase + k < ea) dk = k;      // first marker
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (ed[k] == kDiag && ebase + k >= ea) dk = k;     // nearest marker below hi
                    }
                    const bool all4 = FWD ? (ebase + 4 <= ea) : (ebase >= ea);
                    if (dk >= 0 || all4) {
                        r_cons[tid] = r;
                        e_cons[tid] = bound;
                        acc = rr;
#ifndef EXP_NORESET
                        reinterpret_cast<unsigned long long *>(rhs)[r] = kSentinel;
#endif
                        if (dk >= 0) {
                            if (FWD) { lo = bound; len = dk + 1; hi = lo + len; }
                            else     { hi = bound; len = 4 - dk; lo = hi - len; }
                            phase = 1;
                        } else {
                            // long row: entry loop.  FWD walks up to the marker; BWD first finds the marker below
                            if (FWD) { lo = bound; j = lo; }
                            else     { hi = bound; j = hi - 1; lo = -1; }
                            phase = 2;
                        }
                        progressed = true;
                    }
                }
            }
            if (phase == 1) {
                // register slots of the row: FWD [0,len) with the diagonal last, BWD [4-len,4) with the diagonal first
                const int first = FWD ? 0 : 4 - len;
                const int dslot = FWD ? len - 1 : first;
                // round 2: ring entries of all in-workgroup dependencies at once
                v4i re[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned d = (unsigned)ed[k];
                    const int kl = (int)(d & 0x7fffu);
                    const unsigned os = d >> 15;
                    // one ds_read_b128 either way: the ghost rings sit kEC*kThreads*8 bytes below xr
                    const int at = os >= (unsigned)kGhostBase
                        ? -(kEC * kThreads * 8 / 16) + (kl & (kGD - 1)) * kGhosts + (int)(os & (kGhosts - 1))
                        : (kl & (kXD - 1)) * kThreads + (int)(os & 255u);
                    re[k] = xr[at];
                }
                bool ready = true;
                double xs[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool isdep = FWD ? (k < len - 1) : (k > first);
                    if (isdep && ready) {
                        const unsigned d = (unsigned)ed[k];
                        const unsigned oslot = d >> 15;
                        const int kl = (int)(d & 0x7fffu);
                        if (oslot == myslot && kl == rloc - 1) {
                            xs[k] = prev_val;
                        } else if (oslot >= (unsigned)kGhostBase) {
                            double xc = 0.0;
                            if (ghost((int)(oslot & (kGhosts - 1)), kl, re[k], xc)) xs[k] = xc; else ready = false;
                        } else if ((oslot >> 8) == wg && re[k].x <= kl) {
                            if (re[k].x == kl) xs[k] = __hiloint2double(re[k].w, re[k].z);
                            else ready = false;                                  // producer not there yet
                        } else {
                            double xc = 0.0;
                            if (external(oslot, kl, xc)) xs[k] = xc; else ready = false;
                        }
                    }
                }
                if (ready) {
                    // sequential accumulation in stored order (reference loop order), diagonal by position
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = DESC ? 3 - kk : kk;
                        const bool isdep = FWD ? (k < len - 1) : (k > first);
                        if (isdep) { const double prod = ev[k] * xs[k]; acc = acc - prod; }
                    }
                    double dv = ev[0];
#pragma unroll
                    for (int k = 1; k < 4; ++k) dv = (k == dslot) ? ev[k] : dv;
                    publish(acc / dv);
                    progressed = true;
                } else {
                    ++stall;
                }
            } else if (phase == 2) {
                const int ea = e_avail[tid];
                asm volatile("" ::: "memory");
                if (!FWD && lo < 0) {
                    // BWD: find the row start (the marker) before any arithmetic when entries go ascending
                    while (j >= ea && RD(j) != kDiag) --j;
                    if (j >= ea) {
                        lo = j;
                        if (DESC) { j = hi - 1; jend = lo; } else { j = lo + 1; jend = hi; }
                        dpos = lo;
                        progressed = true;
                    } else {
                        j = hi - 1;      // marker not loaded yet: rescan next round
                    }
                }
                if (FWD || lo >= 0) {
                    for (;;) {
                        if (FWD) {
                            if (j >= ea) break;                               // not loaded yet
                            if (RD(j) == kDiag) { dpos = j; hi = j + 1; jend = j; break; }
                        } else if (j == jend) break;
                        const unsigned d = (unsigned)RD(j);
                        const unsigned oslot = d >> 15;
                        const int kl = (int)(d & 0x7fffu);
                        double xc = 0.0;
                        bool have = false;
                        if (oslot == myslot && kl == rloc - 1) {
                            xc = prev_val;
                            have = true;
                        } else if (oslot >= (unsigned)kGhostBase) {
                            const int g = (int)(oslot & (kGhosts - 1));
                            const v4i e = gr[(kl & (kGD - 1)) * kGhosts + g];
                            if (!ghost(g, kl, e, xc)) { ++stall; break; }
                            have = true;
                        } else if ((oslot >> 8) == wg) {
                            const int lane = (int)(oslot & 255u);
                            const v4i e = xr[(kl & (kXD - 1)) * kThreads + lane];
                            if (e.x == kl) {
                                xc = __hiloint2double(e.w, e.z);
                                have = true;
                            } else if (e.x < kl) {
                                break;
                            }
                        }
                        if (!have && !external(oslot, kl, xc)) break;
                        const double prod = RV(j) * xc;
                        acc = acc - prod;
                        j += DESC ? -1 : 1;
                        progressed = true;
                    }
                    const bool at_end = FWD ? (j < ea && RD(j) == kDiag) : (j == jend);
                    if (at_end) {
                        publish(acc / RV(dpos));
                        progressed = true;
                    }
                }
            }
        }
        if (__any(progressed)) {
            spins = 0;
        } else {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSolveSpinLimit) {
                if ((tid & 63) == 0) atomicExch(err, 1);
                fin[tid] = 1;
                break;
            }
        }
    }
#undef RD
#undef RV
#undef RR
}

int sptrsv(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, const int32_t *desc,
           int32_t max_row_len, double *rhs_and_reset, double *out, int32_t *d_ticket, int32_t *d_err)
{
    // *d_ticket must be zero on entry (the caller zeroes the whole control block once per apply)
    if (desc) {
        const unsigned grid = (unsigned)(sch.nslots / kThreads);
        // rows must fit the entry ring with room for the refill quantum; tiny systems keep the simple kernel
        if (max_row_len > 0 && max_row_len <= kEW - 2 * kEQ && M.nnz >= 16 && M.n >= 8) {
#define LAUNCHS(K)                                                                                           \
            do {                                                                                             \
                ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_lc<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLcLds)); \
                hipLaunchKernelGGL((k_sptrsv_lc<K>), dim3(grid), dim3(kLcThreads), kLcLds, st, M.ptr, desc, M.val, M.n, (long)M.nnz, \
                                   rhs_and_reset, out, sch.nslots, sch.sfirst, sch.scount, sch.exported, sch.gtab, d_ticket, d_err);  \
            } while (0)
#ifdef ILUPP_TIMELINE
            static unsigned long long *d_tl = nullptr;
            if (!d_tl) {
                ILUPP_HIP(hipMalloc(&d_tl, 8 * 8 * 4096));
                ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &d_tl, sizeof(d_tl)));
            }
            ILUPP_HIP(hipMemsetAsync(d_tl, 0, 8 * 8 * 4096, st));
#endif
            switch (kind) {
            case SWEEP_FWD_LAST_ASC: LAUNCHS(SWEEP_FWD_LAST_ASC); break;
            case SWEEP_BWD_FIRST_ASC: LAUNCHS(SWEEP_BWD_FIRST_ASC); break;
            default: LAUNCHS(SWEEP_BWD_FIRST_DESC); break;
            }
#undef LAUNCHS
#ifdef ILUPP_TIMELINE
            if (grid <= 4096) {
                static unsigned long long h_tl[8 * 4096];
                ILUPP_HIP(hipStreamSynchronize(st));
                ILUPP_HIP(hipMemcpy(h_tl, d_tl, 8 * 8 * (size_t)grid, hipMemcpyDeviceToHost));
                char name[64];
                snprintf(name, sizeof name, "/tmp/timeline_%d.bin", (int)kind);
                if (FILE *f = fopen(name, "wb")) { fwrite(h_tl, 8, 8 * (size_t)grid, f); fclose(f); }
            }
#endif
            ILUPP_HIP(hipGetLastError());
            return ILUPP_OK;
        }
#define LAUNCHD(K)                                                                                           \
        hipLaunchKernelGGL((k_sptrsv_desc<K>), dim3(grid), dim3(kThreads), 0, st, M.ptr, desc, M.val, rhs_and_reset, out, \
                           sch.nslots, sch.sfirst, sch.scount, sch.gtab, d_ticket, d_err)
        switch (kind) {
        case SWEEP_FWD_LAST_ASC: LAUNCHD(SWEEP_FWD_LAST_ASC); break;
        case SWEEP_BWD_FIRST_ASC: LAUNCHD(SWEEP_BWD_FIRST_ASC); break;
        default: LAUNCHD(SWEEP_BWD_FIRST_DESC); break;
        }
#undef LAUNCHD
        ILUPP_HIP(hipGetLastError());
        return ILUPP_OK;
    }
    // generic kernel (no descriptors: block size or grid beyond the compact encoding)
    const unsigned grid = (unsigned)((sch.nb + kThreads - 1) / kThreads);
    switch (kind) {
    case SWEEP_FWD_LAST_ASC:
        hipLaunchKernelGGL((k_sptrsv<SWEEP_FWD_LAST_ASC>), dim3(grid), dim3(kThreads), 0, st,
                           M.ptr, M.idx, M.val, rhs_and_reset, out, sch.nb, sch.start, d_ticket, d_err);
        break;
    case SWEEP_BWD_FIRST_ASC:
        hipLaunchKernelGGL((k_sptrsv<SWEEP_BWD_FIRST_ASC>), dim3(grid), dim3(kThreads), 0, st,
                           M.ptr, M.idx, M.val, rhs_and_reset, out, sch.nb, sch.start, d_ticket, d_err);
        break;
    default:
        hipLaunchKernelGGL((k_sptrsv<SWEEP_BWD_FIRST_DESC>), dim3(grid), dim3(kThreads), 0, st,
                           M.ptr, M.idx, M.val, rhs_and_reset, out, sch.nb, sch.start, d_ticket, d_err);
        break;
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

int sptrsv_rows(hipStream_t st, SweepKind kind, const DevMat &M, double *rhs_and_reset, double *out, int32_t *d_ticket, int32_t *d_err)
{
    const unsigned grid = (unsigned)((M.n + kRowsBlock - 1) / kRowsBlock);
    switch (kind) {
    case SWEEP_FWD_LAST_ASC:
        hipLaunchKernelGGL((k_sptrsv_rows<SWEEP_FWD_LAST_ASC>), dim3(grid), dim3(kRowsBlock), 0, st, M.n, M.ptr, M.idx, M.val, rhs_and_reset, out, d_ticket, d_err);
        break;
    case SWEEP_BWD_FIRST_ASC:
        hipLaunchKernelGGL((k_sptrsv_rows<SWEEP_BWD_FIRST_ASC>), dim3(grid), dim3(kRowsBlock), 0, st, M.n, M.ptr, M.idx, M.val, rhs_and_reset, out, d_ticket, d_err);
        break;
    default:
        hipLaunchKernelGGL((k_sptrsv_rows<SWEEP_BWD_FIRST_DESC>), dim3(grid), dim3(kRowsBlock), 0, st, M.n, M.ptr, M.idx, M.val, rhs_and_reset, out, d_ticket, d_err);
        break;
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

}  // namespace ilupp
