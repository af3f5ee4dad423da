// ilupp_amd/csrc/ilut_wp.hip -- ILUT(p, tau) rows computed by a whole wave (gfx950).
//
// Reference ILUT_heap, ILUT.hpp:199-278; threshold_and_drop, dropping.hpp:8-34.  Every wave takes the next row from an atomic
// counter, so a row only waits on rows claimed earlier.  The working row is kept as three insertion-ordered pieces, which is all the reference's
// results depend on (its 2-norms and candidate lists run over the slots of one index range in insertion order,
// sparse_implementation.h:1087-1093, dropping.hpp:14-21):
//   * POOL: entries left of the diagonal that have not been eliminated yet {column, value, seq}; seq numbers the
//     left-part insertions (A's entries in CSR order, then fill in creation order);
//   * KEPT: the multipliers {column, w_k / U_kk, seq} of the eliminated columns (append-only until the row ends: kept in
//     global memory); entries removed by the stage-1 drop (ILUT.hpp:244-245) or found exactly zero (:239-240) are simply
//     forgotten -- the reference leaves a zero slot behind, which adds 0.0 to the norm and can never be a candidate (strict >);
//   * U slots: the entries right of the diagonal in insertion order (they are never removed) with a hash table
//     column -> slot, the stand-in for the reference's n-long occupancy array; the diagonal is a scalar.
// What the reference does with a binary heap -- "next column in ascending order" -- is a wave-wide minimum over the
// pool (DPP reduction) of the LIVE entries: the columns the reference pops and forgets (zero, or below the stage-1 threshold) go
// with the next elimination.  One elimination (round 5: about four dependent trips to memory instead of ten):
//   * top: the wave-wide minimum of what the last pass found; ONE batch of loads -- row k of U, the popped entry, the pool's tail
//     entries that may have to move; the places freed by the last elimination are filled from the tail (nothing depends on an
//     entry's place in the pool: its insertion order is seq);
//   * subtracting the U row: its (<= 16) columns go into scalars and every pool entry is compared against those LEFT of the
//     diagonal (a binary search per entry is a chain of dependent LDS reads that the whole wave pays for as soon as one lane
//     needs it); that same pass finds the next column to eliminate and the places that fall free (WP_SCAN); the entries right of
//     the diagonal look their column up in the hash, each on its own lane, the first probe asked for before the pass; the misses
//     are appended in row order (= the reference's insertion order) by ballot/prefix-sum and enter the hash at the empty cells
//     their walks ended at.
// A finished U row is fetched in ONE memory round trip and, for a budget of 10, one 128-byte line: rows are records (length,
// columns, values: common.h, UrowLayout) initialised to sentinels (index -1, value kSentinel, length 0); the writer stores every
// datum write-through, the reader validates every datum it needs and retries otherwise (write-once data: a set of individually
// fresh values is consistent) -- by the length word alone while the row is not there.
// Dropping: norm in insertion order, candidates by strict >, the p-1 largest by repeated wave-wide maximum with
// (magnitude desc, position asc) order -- equal to std::sort's result unless the cut falls between equal magnitudes
// among more than 16 candidates; then one lane runs libstdc++'s algorithm (stdsort.h) on the candidate list.  Pieces of up to
// 1 024 entries are read once into registers for all of that; the U row is selected and published before the L row is touched.
// The pieces live in LDS (tier 1); a row that outgrows them is started over with its U part in the wave's global-memory arrays
// (64 K entries, private to the wave) and the whole LDS block as its pool (tier 2); a pool that outgrows that MOVES to the global
// arrays between two eliminations and the row goes on (WpResume); a row that outgrows those, or a fill budget beyond the LDS
// selection queue, makes the host run the whole factorisation in the largest capacity class (k_ilut_rows_wp_big: pieces as long
// as the matrix is wide, on fewer waves).
#include <stdio.h>
#include <algorithm>
#include <vector>
#include <stdlib.h>

#include "common.h"
#include "stdsort.h"

namespace ilupp {

// LDS pieces of a wave (pool, U slots, hash cells) in two sizes.  What decides the kernel's time is how many waves a CU holds: it is
// bound by the chains of dependent memory operations of each row, and the time goes like 1 / waves up to 16 waves per CU -- what the
// kernel's 128 VGPRs (amdgpu_waves_per_eu below) and its LDS block allow; DESIGN.md section 4b has the history of that number.
// With pieces of 1 536 entries (49 KB, 3 waves per CU) C3 took 1.12 s and 1 % of its rows started over in global memory; with 128
// entries (6 KB, 20 waves per CU) 74 % of the rows start over (early: the pieces fill within the first eliminations) and it takes
// 0.63 s.  Rows of a factorisation with a large fill budget (p > 32; 48^3 mesh, ILUT(100, 1e-3): ~110 entries per row) get
// 256 entries and 14 waves per CU (65 ms against 230 ms with 128 and 155 ms with 1 536).
static constexpr int kWpSel = 256;      // (the KEPT list is append-only until the end: it lives in global memory)
static constexpr int kWpGCapU = 65534, kWpGCapL = 1 << 15, kWpGCapK = 1 << 15;      // (U slot ids + 1 fit the table's 16-bit cells)
static constexpr int kWpHashG = 1 << 17;
#ifndef ILUT_HASH_SMALL
#define ILUT_HASH_SMALL 4096
#endif
static constexpr int kWpHashSmall = ILUT_HASH_SMALL;    // cells a row's U-slot hash starts on in a global table (wp_row: hm)
#ifndef ILUT_SPIN
#define ILUT_SPIN (1u << 24)
#endif

#ifdef ILUT_PROFILE
// Profile build (profiles/tools/ilut_profile.sh): cycles per phase summed over the waves (ctrl + 8: [0] pops and stage-1 drops, [1] waiting
// for / fetching a U row, [2] the update with it ([6] of that: appending fill), [3] dropping and storing the row, [7] scatter of A's row;
// [4] fetch retries, [5] fetches), and per row: its dependency level, the row it last had to WAIT for, its start and finish times --
// the host walks the chain of last-awaited rows back from the row that finished last: the realised critical path.
#define WP_T(var) const long long var = wall_clock64()
// (per wave in registers, added to ctrl + 8 once when the wave has no more rows: an atomic per phase and fetch on one address from
// 5 120 waves made the kernel eight times slower)
#define WP_ACC(slot, t0, t1) do { prof[slot] += (unsigned long long)((t1) - (t0)); } while (0)
#ifdef ILUT_PROFILE_SUB
// (experiments: slots 4 and 5 -- normally the fetch counters -- take the clock of two sub-phases of the update instead)
#define WP_SUB(which, slot, t0, t1) do { if (ILUT_PROFILE_SUB == (which)) prof[slot] += (unsigned long long)((t1) - (t0)); } while (0)
#else
#define WP_SUB(which, slot, t0, t1)
#endif
__device__ int *g_wp_lvl, *g_wp_parent, *g_wp_lparent, *g_wp_size;      // size: max pool | U slots << 12 | eliminations << 22 | global pieces << 31
__device__ long long *g_wp_tfin, *g_wp_tstart, *g_wp_wait;
#else
#define WP_T(var)
#define WP_ACC(slot, t0, t1)
#define WP_SUB(which, slot, t0, t1)
#endif

// IdT: the type of a U-slot id and of a left-part sequence number: 16 bits for the LDS pieces and the 64 K global pieces, 32 bits
// for the largest capacity class (pieces as long as the matrix is wide)
template <typename IdT> struct WpArraysT {
    IdT *uh; int hmask;                      // column -> U slot + 1 (0 = empty), open addressing; all cells 0 between rows
    int *ucol; double *uval; int capU;
    int *lcol; double *lval; IdT *lseq; int capL;      // (seq counts the left-part insertions of one row)
    int *kcol; double *kval; IdT *kseq; int capK;
};
typedef WpArraysT<unsigned short> WpArrays;
template <typename IdT> struct WpIdMax { static constexpr int value = 65535; };
template <> struct WpIdMax<unsigned int> { static constexpr int value = 0x7ffffff0; };

// accessors of the working-row pieces: LDS, or the wave's PRIVATE global arrays.  Those are only ever touched by this
// wave (one CU, one L1), so plain accesses are coherent once the stores have been acknowledged (s_waitcnt in sync()).
template <bool G> struct WpAcc {
    static __device__ __forceinline__ int ldi(const int *p) { return *p; }
    static __device__ __forceinline__ double ldd(const double *p) { return *p; }
    static __device__ __forceinline__ void sti(int *p, int v) { *p = v; }
    static __device__ __forceinline__ void std_(double *p, double v) { *p = v; }
    static __device__ __forceinline__ int ldi(const unsigned short *p) { return (int)*p; }
    static __device__ __forceinline__ void sti(unsigned short *p, int v) { *p = (unsigned short)v; }
    static __device__ __forceinline__ int ldi(const unsigned int *p) { return (int)*p; }
    static __device__ __forceinline__ void sti(unsigned int *p, int v) { *p = (unsigned int)v; }
    // cross-lane hand-over inside the wave: LDS is in order per wave; global stores must have landed
#ifdef ILUT_NOACK
    // (experiment: a wave's own loads and stores are executed in order, so the acknowledgement is not needed for the hand-over between
    // lanes of ONE wave -- same bits on C3 and 243 -> 238 ms; the product keeps the wait)
    static __device__ __forceinline__ void sync() { __builtin_amdgcn_wave_barrier(); }
#else
    static __device__ __forceinline__ void sync() { if constexpr (G) __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier(); }
#endif
};

template <int CTRL, int RM, int BM> __device__ __forceinline__ unsigned wp_dpp(unsigned identity, unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, RM, BM, false);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, wp_dpp<0x111, 0xf, 0xf>(~0u, v));    // row_shr:1
    v = min(v, wp_dpp<0x112, 0xf, 0xf>(~0u, v));    // row_shr:2
    v = min(v, wp_dpp<0x114, 0xf, 0xe>(~0u, v));    // row_shr:4
    v = min(v, wp_dpp<0x118, 0xf, 0xc>(~0u, v));    // row_shr:8
    v = min(v, wp_dpp<0x142, 0xa, 0xf>(~0u, v));    // row_bcast:15
    v = min(v, wp_dpp<0x143, 0xc, 0xf>(~0u, v));    // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, wp_dpp<0x111, 0xf, 0xf>(0u, v));
    v = max(v, wp_dpp<0x112, 0xf, 0xf>(0u, v));
    v = max(v, wp_dpp<0x114, 0xf, 0xe>(0u, v));
    v = max(v, wp_dpp<0x118, 0xf, 0xc>(0u, v));
    v = max(v, wp_dpp<0x142, 0xa, 0xf>(0u, v));
    v = max(v, wp_dpp<0x143, 0xc, 0xf>(0u, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
    const unsigned h = (unsigned)(v >> 32);
    const unsigned gh = wave_max_u32(h);
    const unsigned gl = wave_max_u32(h == gh ? (unsigned)v : 0u);
    return ((unsigned long long)gh << 32) | gl;
}
__device__ __forceinline__ double wave_bcast_f64(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// threshold_and_drop over one piece (dropping.hpp:8-34): cols/vals[0..cnt) in insertion order; writes the kept entries
// by increasing column to out_idx/out_val (STORE_AGENT: write-through, sentinel-safe) and returns their number
// (SELG: the selection queue lives in global memory too -- budgets beyond kWpSel: hand-overs between lanes wait for the stores)
template <bool G, bool STORE_AGENT, bool SELG = false>
__device__ __forceinline__ int wp_select(const int lane, const int *cols, const double *vals, const int cnt, const int nkeep,
                                         const double tau, int *selq, int *gscratch, int *out_idx, double *out_val)
{
    using A = WpAcc<G>;
    if (nkeep <= 0) return 0;                                                 // dropping.hpp:11-12
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nsel;
    constexpr int kRC = 16;
    if (cnt <= 64 * kRC) {
        // Pieces of up to 1 024 entries (all but a handful of rows) are read ONCE, lane t holding entries t, t + 64, ...: the norm's
        // ordered sum takes the squares out of the registers lane by lane (same operands, same order as one lane adding them up),
        // the candidate test and every selection round work on the registers.  Reading the piece again per round -- a trip per 64
        // entries, a dozen rounds, and the norm a trip per four entries -- was 160 us per row, all of it between a row's last
        // elimination and its publication.
        double vr[kRC];
#pragma unroll
        for (int u = 0; u < kRC; ++u) { const int q = 64 * u + lane; vr[u] = q < cnt ? A::ldd(&vals[q]) : 0.0; }
        double z = 0.0;
#pragma unroll
        for (int u = 0; u < kRC; ++u) {
            if (64 * u < cnt) {
                const double sq = vr[u] * vr[u];
                const long long sb = __double_as_longlong(sq);
                const int rem = cnt - 64 * u < 64 ? cnt - 64 * u : 64;
#define WP_ZADD(l_) do { const int lo = __builtin_amdgcn_readlane((int)sb, (l_)), hi = __builtin_amdgcn_readlane((int)(sb >> 32), (l_)); \
                         z = z + __longlong_as_double(((long long)hi << 32) | (unsigned)lo); } while (0)
                int l = 0;
                for (; l + 4 <= rem; l += 4) { WP_ZADD(l); WP_ZADD(l + 1); WP_ZADD(l + 2); WP_ZADD(l + 3); }
                for (; l < rem; ++l) WP_ZADD(l);
#undef WP_ZADD
            }
        }
        const double thr = sqrt(z) * tau;
        unsigned long long mbr[kRC];
        int ncand = 0;
#pragma unroll
        for (int u = 0; u < kRC; ++u) {
            const double a = fabs(vr[u]);
            const bool is = 64 * u + lane < cnt && a > thr;
            mbr[u] = is ? (unsigned long long)__double_as_longlong(a) : 0ull;
            if (64 * u < cnt) ncand += __popcll(__ballot(is));
        }
        if (ncand <= nkeep) {
            nsel = 0;
#pragma unroll
            for (int u = 0; u < kRC; ++u) {
                if (64 * u < cnt) {
                    const bool is = mbr[u] != 0ull;
                    const unsigned long long m = __ballot(is);
                    if (is) selq[nsel + __popcll(m & lt)] = 64 * u + lane;
                    nsel += __popcll(m);
                }
            }
        } else {
            unsigned long long pm = ~0ull;
            int pp = -1;
            bool tie = false;
            for (int t = 0; t <= nkeep; ++t) {          // the extra round finds the first entry NOT kept (tie test)
                unsigned long long bm = 0ull;
                unsigned bq = 0x7fffffffu;
#pragma unroll
                for (int u = 0; u < kRC; ++u) {
                    const unsigned long long mb = mbr[u];
                    const int q = 64 * u + lane;
                    const bool after = mb < pm || (mb == pm && q > pp);
                    if (mb != 0ull && after && mb > bm) { bm = mb; bq = (unsigned)q; }
                }
                const unsigned long long gm = wave_max_u64(bm);
                const unsigned gq = wave_min_u32(bm == gm ? bq : 0x7fffffffu);
                if (t < nkeep) { if (lane == 0) selq[t] = (int)gq; }
                else tie = gm == pm;
                pm = gm; pp = (int)gq;
            }
            nsel = nkeep;
            if (tie && ncand > 16) {
                // equal magnitudes across the cut: the kept set is what libstdc++'s introsort leaves in front (stdsort.h)
                int c = 0;
                for (int base = 0; base < cnt; base += 64) {
                    const int q = base + lane;
                    const bool is = q < cnt && fabs(A::ldd(&vals[q])) > thr;
                    const unsigned long long m = __ballot(is);
                    if (is) gscratch[c + __popcll(m & lt)] = q;
                    c += __popcll(m);
                }
                __builtin_amdgcn_s_waitcnt(0);
                if (lane == 0) {
                    c_sort_slots_by_abs_desc(gscratch, ncand, vals);
                    for (int t = 0; t < nkeep; ++t) selq[t] = gscratch[t];
                }
                __builtin_amdgcn_s_waitcnt(0);
            }
        }
    } else {
        double z = 0.0;
    #pragma unroll 4
        for (int q = 0; q < cnt; ++q) { const double v = A::ldd(&vals[q]); const double sq = v * v; z = z + sq; }
        const double thr = sqrt(z) * tau;
        int ncand = 0;
        for (int base = 0; base < cnt; base += 64) {
            const int q = base + lane;
            const bool is = q < cnt && fabs(A::ldd(&vals[q])) > thr;
            ncand += __popcll(__ballot(is));
        }
        if (ncand <= nkeep) {
            nsel = 0;
            for (int base = 0; base < cnt; base += 64) {
                const int q = base + lane;
                const bool is = q < cnt && fabs(A::ldd(&vals[q])) > thr;
                const unsigned long long m = __ballot(is);
                if (is) selq[nsel + __popcll(m & lt)] = q;
                nsel += __popcll(m);
            }
        } else {
            unsigned long long pm = ~0ull;
            int pp = -1;
            bool tie = false;
            for (int t = 0; t <= nkeep; ++t) {          // the extra round finds the first entry NOT kept (tie test)
                unsigned long long bm = 0ull;
                unsigned bq = 0x7fffffffu;
                for (int q = lane; q < cnt; q += 64) {
                    const double a = fabs(A::ldd(&vals[q]));
                    if (!(a > thr)) continue;
                    const unsigned long long mb = (unsigned long long)__double_as_longlong(a);
                    const bool after = mb < pm || (mb == pm && q > pp);
                    if (after && mb > bm) { bm = mb; bq = (unsigned)q; }
                }
                const unsigned long long gm = wave_max_u64(bm);
                const unsigned gq = wave_min_u32(bm == gm ? bq : 0x7fffffffu);
                if (t < nkeep) { if (lane == 0) selq[t] = (int)gq; }
                else tie = gm == pm;
                pm = gm; pp = (int)gq;
            }
            nsel = nkeep;
            if (tie && ncand > 16) {
                // equal magnitudes across the cut: the kept set is what libstdc++'s introsort leaves in front (stdsort.h)
                int c = 0;
                for (int base = 0; base < cnt; base += 64) {
                    const int q = base + lane;
                    const bool is = q < cnt && fabs(A::ldd(&vals[q])) > thr;
                    const unsigned long long m = __ballot(is);
                    if (is) gscratch[c + __popcll(m & lt)] = q;
                    c += __popcll(m);
                }
                __builtin_amdgcn_s_waitcnt(0);
                if (lane == 0) {
                    c_sort_slots_by_abs_desc(gscratch, ncand, vals);
                    for (int t = 0; t < nkeep; ++t) selq[t] = gscratch[t];
                }
                __builtin_amdgcn_s_waitcnt(0);
            }
        }
    }
    if constexpr (SELG) __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // by increasing column (dropping.hpp:32-33; unique keys)
    for (int t = lane; t - lane < nsel; t += 64) {
        const int q = t < nsel ? selq[t] : 0;
        const int c = t < nsel ? A::ldi(&cols[q]) : 0x7fffffff;
        int r = 0;
        if (nsel <= 64) {
            // (the kept columns are in the lanes: their ranks without another trip per column)
            for (int t2 = 0; t2 < nsel; ++t2) r += (__builtin_amdgcn_readlane(c, t2) < c) ? 1 : 0;
        } else {
            for (int t2 = 0; t2 < nsel; ++t2) r += (A::ldi(&cols[selq[t2]]) < c) ? 1 : 0;
        }
        if (t >= nsel) continue;
        double v = A::ldd(&vals[q]);
        if constexpr (STORE_AGENT) {
            if ((unsigned long long)__double_as_longlong(v) == kSentinel) v = __longlong_as_double((long long)kCanonNaN);
            st_agent_f64(&out_val[r], v);
            st_agent_i32(&out_idx[r], c);
        } else {
            out_val[r] = v;
            out_idx[r] = c;
        }
    }
    return nsel;
}

// U-slot hash: what the reference does with its n-long occupancy array (sparse_implementation.h:987-993)
__device__ __forceinline__ unsigned wp_hash(int c, int hmask) { return (((unsigned)c * 0x9E3779B1u) >> 12) & (unsigned)hmask; }
// (the walk for a column -- to its slot or to the empty cell where it goes -- is written out in wp_row: its first probe is asked for before
// the pass over the pool, and the slot's value comes with its key)
// all lanes with `mine` insert their (distinct) columns at once: everybody walks to an empty cell; where two lanes stand at the same
// cell the lower lane takes it (found by comparing the cells inside the wave -- reading the cell back was two more trips to the
// table) and the other walks on once the winners' writes have landed
template <bool G, typename IdT>
__device__ __forceinline__ void wp_uh_insert_all(const WpArraysT<IdT> &w, const int hmask, bool mine, int c, int slot, unsigned at = ~0u)
{
    // (at: a cell the caller's own walk for column c has just found empty -- no second walk to it)
    bool known = at != ~0u;
    unsigned h = known ? at : wp_hash(c, hmask);
    bool pending = mine;
    unsigned long long pm;
    while ((pm = __ballot(pending)) != 0ull) {
        if (pending && !known) { while (w.uh[h] != 0u) h = (h + 1) & (unsigned)hmask; }
        known = false;
        bool lose = false;
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        for (unsigned long long m = pm; m != 0ull; m &= m - 1ull) {
            const int l = __ffsll((long long)m) - 1;
            const unsigned hh = (unsigned)__builtin_amdgcn_readlane((int)h, l);
            lose = lose || (l < lane && hh == h);
        }
        if (pending && !lose) { w.uh[h] = (IdT)(slot + 1); pending = false; }
        if (__ballot(pending) == 0ull) break;                       // (the caller's hand-over covers the writes)
        WpAcc<G>::sync();
        if (pending) h = (h + 1) & (unsigned)hmask;
    }
}

#ifndef ILUT_POLL_NAP
#define ILUT_POLL_NAP 8
#endif

// a row between two eliminations (wp_row's alt_*: returned with 3, taken up again by the next call)
struct WpResume {
    int active, nL, nU, nK, seq, klast, hm;
    double wdiag, thr1;
#ifdef ILUT_PROFILE
    int lvl, parent, lparent, maxl;
    long long wait, t0;
#endif
};

// one row; returns 0 = done, 1 = a piece outgrew its capacity (nothing was published), 2 = timeout, 3 = the pool moved (see alt_*)
// this row's cells of the wave's table in global memory: found first (nothing is removed while anybody still walks), then emptied
template <bool G, typename IdT>
__device__ __forceinline__ void wp_uh_clear(const WpArraysT<IdT> &w, const int hmask, const int lane, const int nU, int *gscratch)
{
    if (!G) return;
    WpAcc<G>::sync();
    for (int q = lane; q < nU; q += 64) {
        unsigned h = wp_hash(WpAcc<G>::ldi(&w.ucol[q]), hmask);
        while (w.uh[h] != (IdT)(q + 1)) h = (h + 1) & (unsigned)hmask;
        gscratch[q] = (int)h;
    }
    WpAcc<G>::sync();
    for (int q = lane; q < nU; q += 64) w.uh[gscratch[q]] = (IdT)0;
    WpAcc<G>::sync();
}

template <bool G, typename IdT = unsigned short, bool SELG = false>
__device__ __forceinline__ int wp_row(const int lane, const int i, const int n, const int p, const double tau,
                                      const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval,
                                      int32_t *Lrow_idx, double *Lrow_val, int32_t *Llen,
                                      int32_t *Urow_idx, double *Urow_val, int32_t *Ulen, const UrowLayout ul_,
                                      const WpArraysT<IdT> w, int *bcol, double *bpr, int *bfound, int *dlist, int *selq, int *gscratch, int32_t *ctrl,
                                      unsigned long long *prof = nullptr, WpResume *rs = nullptr, int *alt_lcol = nullptr,
                                      double *alt_lval = nullptr, IdT *alt_lseq = nullptr, int alt_capL = 0)
{
    using A = WpAcc<G>;
    (void)prof;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nL = 0, nU = 0, nK = 0, seq = 0;
    int seen_done = -1, klast = -1;
    double wdiag = 0.0, thr1 = 0.0;
    // The U-slot hash of a row in the wave's GLOBAL table starts on the table's first 4 096 cells (8 KB: 64 lines that stay in the
    // L2 / the MALL between two probes) and moves to the whole table (256 KB and more, every probe a line from HBM: 120 of the
    // kernel's 150 GiB of fetches on C3) only when the row has more than 2 048 - p U slots: the cells of a table are all 0 between rows
    // whatever mask the last row used.  (A row of A with more than 1 024 entries starts on the whole table.)
    int hm = w.hmask;
    if (G && w.hmask > kWpHashSmall - 1 && Aptr[i + 1] - Aptr[i] <= kWpHashSmall / 4) hm = kWpHashSmall - 1;
#ifdef ILUT_PROFILE
    int prof_lvl = 0, prof_parent = -1, prof_lparent = -1, prof_maxl = 0;
    long long prof_wait = 0;
    long long prof_t0 = wall_clock64();
#endif
    // alt_*: a second, larger home for the pool (the wave's global arrays, for the tier with the pool in LDS: the U part, its hash and
    // the kept list are the same arrays in both).  A row whose pool is about to outgrow its home copies it there BETWEEN two
    // eliminations and returns 3; the caller goes on with the row (rs) on the other arrays instead of starting again from A's row --
    // such a row is a long one, and on the chain of deepest dependencies its second start was the longest link of all.
    const bool resume = rs != nullptr && rs->active != 0;
    if (resume) {
        nL = rs->nL; nU = rs->nU; nK = rs->nK; seq = rs->seq; klast = rs->klast; wdiag = rs->wdiag; thr1 = rs->thr1; hm = rs->hm;
#ifdef ILUT_PROFILE
        prof_lvl = rs->lvl; prof_parent = rs->parent; prof_lparent = rs->lparent; prof_maxl = rs->maxl; prof_wait = rs->wait; prof_t0 = rs->t0;
#endif
    }
#define WP_MOVE_POOL()                                                                                                  \
    do {                                                                                                                \
        for (int q = lane; q < nL; q += 64) {                                                                           \
            const int c2 = A::ldi(&w.lcol[q]);                                                                          \
            const double v2 = A::ldd(&w.lval[q]);                                                                       \
            const int s2 = A::ldi(&w.lseq[q]);                                                                          \
            alt_lcol[q] = c2; alt_lval[q] = v2; A::sti(&alt_lseq[q], s2);                                               \
        }                                                                                                               \
        A::sync();                                                                                                      \
        rs->active = 1; rs->nL = nL; rs->nU = nU; rs->nK = nK; rs->seq = seq; rs->klast = klast; rs->wdiag = wdiag; rs->thr1 = thr1; rs->hm = hm; \
        WP_MOVE_PROF();                                                                                                 \
        return 3;                                                                                                       \
    } while (0)
#ifdef ILUT_PROFILE
#define WP_MOVE_PROF() do { rs->lvl = prof_lvl; rs->parent = prof_parent; rs->lparent = prof_lparent; rs->maxl = prof_maxl; rs->wait = prof_wait; rs->t0 = prof_t0; } while (0)
#else
#define WP_MOVE_PROF() do { } while (0)
#endif
    if (!resume) {
    // the U-slot hash starts empty: the LDS table is cleared here; the wave's table in global memory (256 KB and more) is empty between
    // rows -- the host clears it once, every row takes its own cells out again when it is done (clearing all of it per row wrote
    // 190 GB on C3: three rows of four work there)
    if (!G) {
        unsigned long long *t64 = reinterpret_cast<unsigned long long *>(w.uh);
        for (int q = lane; q < (int)((size_t)(w.hmask + 1) * sizeof(IdT) / 8); q += 64) t64[q] = 0ull;
        A::sync();
    }
    // (2.) scatter the row (ILUT.hpp:222-231)
    const int a0 = Aptr[i], a1 = Aptr[i + 1];
    for (int base = a0; base < a1; base += 64) {
        const int q = base + lane;
        const bool valid = q < a1;
        const int c = valid ? Aidx[q] : 0x7fffffff;
        const double v = valid ? Aval[q] : 0.0;
        const bool isL = valid && c < i, isU = valid && c > i, isD = valid && c == i;
        const unsigned long long mL = __ballot(isL), mU = __ballot(isU), mD = __ballot(isD);
        if (seq + __popcll(mL) > WpIdMax<IdT>::value) { wp_uh_clear<G, IdT>(w, hm, lane, nU, gscratch); return 1; }
        if (nL + __popcll(mL) > w.capL || nU + __popcll(mU) > w.capU) { if (!G && lane == 0) atomicAdd(&ctrl[nL + __popcll(mL) > w.capL ? 4 : 5], 1); wp_uh_clear<G, IdT>(w, hm, lane, nU, gscratch); return 1; }
        if (isL) { const int pos = nL + __popcll(mL & lt); A::sti(&w.lcol[pos], c); A::std_(&w.lval[pos], v); A::sti(&w.lseq[pos], seq + __popcll(mL & lt)); }
        if (isU) { const int pos = nU + __popcll(mU & lt); A::sti(&w.ucol[pos], c); A::std_(&w.uval[pos], v); }
        A::sync();
        wp_uh_insert_all<G, IdT>(w, hm, isU, c, nU + __popcll(mU & lt));
        if (mD != 0ull) wdiag = wave_bcast_f64(v, __ffsll((long long)mD) - 1);
        nL += __popcll(mL); seq += __popcll(mL); nU += __popcll(mU);
    }
    A::sync();
    {
        double z = 0.0;
        for (int q = 0; q < nL; ++q) { const double v = A::ldd(&w.lval[q]); const double sq = v * v; z = z + sq; }
        thr1 = tau * sqrt(z);
    }
    WP_ACC(7, prof_t0, wall_clock64());
    }
    // (3.-9.) eliminate in ascending column order (ILUT.hpp:234-255)
    unsigned best = 0x7fffffffu;
    int bq = -1, nd = 0;
    bool have = false;
    int bc[16];
    // one pass over the pool: every lane's smallest live column (best, at bq), the places of the entries that went with the last
    // elimination (column <= klast: dlist, nd of them) and, for N > 0, the update with the U row at hand (columns bc[1..N), products bpr).
    // Four chunks of 64 entries are asked for together: a chunk per trip was most of such a pass -- the pool is reached by flat or
    // global loads, 200 cycles and more each.
#define WP_SCAN(N)                                                                                                      \
    best = 0x7fffffffu; bq = -1; nd = 0;                                                                                \
    for (int base_ = 0; base_ < nL; base_ += 256) {                                                                     \
        int c4[4];                                                                                                      \
        double v4[4];                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                 \
            const int q = base_ + 64 * u + lane;                                                                        \
            c4[u] = q < nL ? A::ldi(&w.lcol[q]) : 0x7fffffff;                                                           \
            v4[u] = q < nL ? A::ldd(&w.lval[q]) : 0.0;                                                                  \
        }                                                                                                               \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                 \
            const int q = base_ + 64 * u + lane;                                                                        \
            const bool valid = q < nL;                                                                                  \
            const int c_ = c4[u];                                                                                       \
            double v_ = v4[u];                                                                                          \
            if ((N) > 0) {                                                                                              \
                int ma = -1;                                                                                            \
                _Pragma("unroll") for (int jj = 1; jj < ((N) > 0 ? (N) : 1); ++jj) ma = c_ == bc[jj] ? jj : ma;         \
                if (ma >= 0) { v_ = v_ - bpr[ma]; A::std_(&w.lval[q], v_); bfound[ma] = 1; }                           \
            }                                                                                                           \
            const bool dead = valid && c_ <= klast;                                                                     \
            const unsigned long long md = __ballot(dead);                                                               \
            if (md != 0ull) {                                                                                           \
                if (dead) { const int pos = nd + __popcll(md & lt); if (pos < 63) dlist[pos] = q; }                     \
                nd += __popcll(md);                                                                                     \
            }                                                                                                           \
            const bool live = valid && !dead && v_ != 0.0 && !(fabs(v_) < thr1);                                        \
            if (live && (unsigned)c_ < best) { best = (unsigned)c_; bq = q; }                                           \
        }                                                                                                               \
    }
    for (;;) {
        WP_T(tp0);
#ifdef ILUT_PROFILE
        prof_maxl = nL > prof_maxl ? nL : prof_maxl;
#endif
        // (an elimination appends at most p - 1 entries -- a U row has p at most)
        if (G && hm != w.hmask && nU + p > (hm + 1) / 2) {
            // the U-slot hash moves to the whole table: its cells out of the small one, the slots in again
            wp_uh_clear<G, IdT>(w, hm, lane, nU, gscratch);
            hm = w.hmask;
            for (int base = 0; base < nU; base += 64) {
                const int q = base + lane;
                const int c2 = q < nU ? A::ldi(&w.ucol[q]) : 0;
                wp_uh_insert_all<G, IdT>(w, hm, q < nU, c2, q);
                A::sync();
            }
        }
        if (G && alt_lcol != nullptr && nL + p > w.capL && nL + p <= alt_capL) WP_MOVE_POOL();
        // The next column that is ELIMINATED: the smallest one whose entry is neither zero (ILUT.hpp:239-240) nor below the stage-1
        // threshold (:244-245).  The reference pops every column in ascending order and forgets those; a forgotten entry has no
        // effect on anything, and an entry's value only changes when a smaller column is eliminated -- so every entry left of the
        // next eliminated column has, now, the value it would have when popped: all of them go at once (on C3 78 % of the pops,
        // each a pass over the pool, end this way: 600 passes per row became 134).
        // ONE pass over the pool per elimination: the minimum over the live entries, and the places of the entries that went with the
        // previous elimination (column <= klast: popped or forgotten).  Nothing depends on an entry's place in the pool (its order of
        // insertion is lseq), so those places are filled from the pool's tail instead of moving everything up -- a rewrite of the whole
        // pool, with two hand-overs per 64 entries, was the larger half of this phase.
        // The pass that updates the pool with a U row (WP_SCAN(N > 0), below) does this pass's work for the NEXT elimination on the way
        // -- it has every entry's column and new value in hand -- and the entries appended behind it join in: `have`.
        if (!have) { WP_SCAN(0) }
        have = false;
        WP_T(tpa); WP_SUB(2, 5, tp0, tpa);
        const unsigned g = wave_min_u32(best);
        if (g == 0x7fffffffu) break;                                         // (what is left would be popped and forgotten)
        const unsigned long long who = __ballot(best == g);
        const int qs = __builtin_amdgcn_readlane(bq, __ffsll((long long)who) - 1);
        const int k = (int)g;
        // Everything the top of an elimination reads is asked for TOGETHER: row k of U (the first attempt of the fetch below), the popped
        // entry, and the tail entries that may have to move into freed places -- one trip where there were five in a row.
        const size_t ubi = (size_t)k * ul_.si, ubv = (size_t)k * ul_.sv, ubl = (size_t)k * ul_.sl;      // (row k's record: common.h, UrowLayout)
#ifdef ILUT_FIRST_PLAIN
        // (experiment: the first attempt with loads that may be served by this XCD's L2 -- a record is written once, so whatever a stale
        // line holds is either "not there" or final; the retries below go past the L2)
        int ul = __hip_atomic_load(&Ulen[ubl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        int c0 = lane < p ? __hip_atomic_load(&Urow_idx[ubi + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : -1;
        unsigned long long v0 = lane < p ? __hip_atomic_load(reinterpret_cast<const unsigned long long *>(&Urow_val[ubv + lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0ull;
#else
        int ul = ld_agent_i32(&Ulen[ubl]);
        int c0 = lane < p ? ld_agent_i32(&Urow_idx[ubi + lane]) : -1;
        unsigned long long v0 = lane < p ? ld_agent_u64(reinterpret_cast<const unsigned long long *>(&Urow_val[ubv + lane])) : 0ull;
#endif
        const double wkv = A::ldd(&w.lval[qs]);
        const int sk = A::ldi(&w.lseq[qs]);
        const bool fill_holes = nd < 63;
        const int nLn = nL - (nd + 1);
        const int mq = nLn + lane;
        const bool mvalid = fill_holes && lane <= nd;
        const int mc = mvalid ? A::ldi(&w.lcol[mq]) : 0;
        const double mval = mvalid ? A::ldd(&w.lval[mq]) : 0.0;
        const int msq = mvalid ? A::ldi(&w.lseq[mq]) : 0;
        A::sync();
        if (fill_holes) {
            // (the popped entry goes too) the r-th free place below the new end takes the r-th surviving entry behind it
            if (lane == 0) dlist[nd] = qs;
            ++nd;
            __builtin_amdgcn_wave_barrier();
            const int hq = lane < nd ? dlist[lane] : 0x7fffffff;
            const bool is_hole = lane < nd && hq < nLn;
            const bool is_mover = lane < nd && mc > klast && mq != qs;
            const unsigned long long mh = __ballot(is_hole), mm = __ballot(is_mover);
            if (is_hole) bfound[__popcll(mh & lt)] = hq;
            __builtin_amdgcn_wave_barrier();
            if (is_mover) {
                const int dst = bfound[__popcll(mm & lt)];
                A::sti(&w.lcol[dst], mc); A::std_(&w.lval[dst], mval); A::sti(&w.lseq[dst], msq);
            }
            nL = nLn;
            // (the moved entries are read again behind the fetch below, whose wait covers these stores)
            __builtin_amdgcn_wave_barrier();
        } else {
            // (more went than the list holds: the pool keeps the entries right of column k, in place -- a chunk's entries are in
            // registers before any of them is written, and they move to positions at or before their own)
            int kept = 0;
            for (int base = 0; base < nL; base += 64) {
                const int q = base + lane;
                const bool valid = q < nL;
                const int c = valid ? A::ldi(&w.lcol[q]) : 0;
                const double v = valid ? A::ldd(&w.lval[q]) : 0.0;
                const int sq = valid ? A::ldi(&w.lseq[q]) : 0;
                const bool keep = valid && c > k;
                const unsigned long long mk = __ballot(keep);
                A::sync();
                if (keep) { const int pos = kept + __popcll(mk & lt); A::sti(&w.lcol[pos], c); A::std_(&w.lval[pos], v); A::sti(&w.lseq[pos], sq); }
                kept += __popcll(mk);
                A::sync();
            }
            nL = kept;
        }
        klast = k;
        // row k of U, validated against the sentinels
        WP_T(tp1); WP_ACC(0, tp0, tp1);
        unsigned spins = 0, idle = 0;
        (void)spins;
        for (;;) {
            const bool bad = ul <= 0 || (lane < ul && (c0 < 0 || v0 == kSentinel));
            if (__ballot(bad) == 0ull) break;
            ++spins;
            // the limit counts polls during which NO row was finished anywhere (ctrl[7]): a long chain elsewhere is not a hang
            if (++idle > ILUT_SPIN) return 2;
            if ((idle & 4095u) == 0u) { const int f = ld_agent_i32(&ctrl[7]); if (f != seen_done) { seen_done = f; idle = 0; } }
            // (a row that is not there yet is asked for by its length word alone -- one request instead of three per poll, and not
            // more often than a trip takes: the pollers share the L2 with the waves that work)
            __builtin_amdgcn_s_sleep(ILUT_POLL_NAP);
            ul = __builtin_amdgcn_readfirstlane(ld_agent_i32(&Ulen[ubl]));
            if (ul > 0) {
                c0 = lane < p ? ld_agent_i32(&Urow_idx[ubi + lane]) : -1;
                v0 = lane < p ? ld_agent_u64(reinterpret_cast<const unsigned long long *>(&Urow_val[ubv + lane])) : 0ull;
            }
        }
#ifdef ILUT_PROFILE
#ifndef ILUT_PROFILE_SUB
        prof[4] += spins; prof[5] += 1ull;
#endif
#endif
        ul = __builtin_amdgcn_readfirstlane(ul);
        WP_T(tp2); WP_ACC(1, tp1, tp2);
#ifdef ILUT_PROFILE
        {
            const int lk = ld_agent_i32(&g_wp_lvl[k]);
            if (lk > prof_lvl) { prof_lvl = lk; prof_lparent = k; }
            if (spins > 0) { prof_parent = k; prof_wait += tp2 - tp1; }
        }
#endif
        const double ud = wave_bcast_f64(__longlong_as_double((long long)v0), 0);
        const double m = wkv / ud;                                           // :249
        if (nK >= w.capK) { if (!G && lane == 0) atomicAdd(&ctrl[6], 1); wp_uh_clear<G, IdT>(w, hm, lane, nU, gscratch); return 1; }
        if (lane == 0) { w.kcol[nK] = k; w.kval[nK] = m; w.kseq[nK] = (IdT)sk; }
        ++nK;
        for (int base = 0; base < ul; base += 64) {                          // w -= m * U[k, j>k]  (:252-253)
            WP_T(tu0);
            const int j = base + lane;
            int c = c0;
            unsigned long long vb = v0;
            if (base > 0) {
                unsigned sp2 = 0;
                for (;;) {
                    c = j < ul ? ld_agent_i32(&Urow_idx[ubi + j]) : 0;
                    vb = j < ul ? ld_agent_u64(reinterpret_cast<const unsigned long long *>(&Urow_val[ubv + j])) : 0ull;
                    const bool bad = j < ul && (c < 0 || vb == kSentinel);
                    if (__ballot(bad) == 0ull) break;
                    if (++sp2 > ILUT_SPIN) return 2;
                    if ((sp2 & 4095u) == 0u) { const int f = ld_agent_i32(&ctrl[7]); if (f != seen_done) { seen_done = f; sp2 = 0; } }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const bool valid = j < ul && j >= 1;
            const double pr = m * __longlong_as_double((long long)vb);
            const int cnt = ul - base < 64 ? ul - base : 64;
            bcol[lane] = j < ul ? c : 0x7fffffff;
            bpr[lane] = pr;
            bfound[lane] = 0;
            __builtin_amdgcn_wave_barrier();
            const int bmin = bcol[0], bmax = bcol[cnt - 1];
            // (the first probe of the U-slot hash is asked for before the pass over the pool, whose trip it shares)
            const bool uside = valid && c > i;
            unsigned hU = wp_hash(c, hm);
            unsigned eU = uside ? (unsigned)w.uh[hU] : 0u;
            WP_T(tu1); WP_SUB(1, 4, tu0, tu1);
            if (cnt <= 16 && base == 0) {
                // the usual case (p <= 16): the U row's columns sit in scalars and every slot is compared against them -- a binary
                // search per slot is a chain of dependent LDS reads that the whole wave pays for as soon as one lane's slot lies in
                // [bmin, bmax], i.e. always (it was 5-6 us per U row, 3/4 of a row's time).  Only the columns LEFT of the diagonal
                // can be in the pool -- the first cntL of the ascending row, its own diagonal entry (column k, popped) aside: rows
                // eliminated late have few of those or none, and the pass over the pool is as long as that number asks for.
                const int cntL = __popcll(__ballot(j < ul && c < i));
                if (cntL > 1) {
#pragma unroll
                    for (int jj = 1; jj < 16; ++jj) bc[jj] = __builtin_amdgcn_readfirstlane(jj < cntL ? bcol[jj] : -1);
                    if (cntL <= 4) { WP_SCAN(4) } else if (cntL <= 8) { WP_SCAN(8) } else { WP_SCAN(16) }
                    have = true;
                }
            } else {
            for (int q = lane; q < nL; q += 64) {
                const int c2 = A::ldi(&w.lcol[q]);
                if (c2 >= bmin && c2 <= bmax) {
                    int lo = 0, hi = cnt - 1;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (bcol[mid] < c2) lo = mid + 1; else hi = mid; }
                    if (bcol[lo] == c2) { const double o = A::ldd(&w.lval[q]); A::std_(&w.lval[q], o - bpr[lo]); bfound[lo] = 1; }
                }
            }
            }
            // right of the diagonal: the entry's own lane looks its column up in the hash (the reference's occupancy[] access)
            WP_T(tu2); WP_SUB(1, 5, tu1, tu2);
            bool ufound = false;
            if (uside) {
                // (the walk ends at the slot of column c or at an empty cell -- hU: where the column goes if it is appended below)
                while (eU != 0u) {
                    const int cc = A::ldi(&w.ucol[eU - 1]);
                    const double o = A::ldd(&w.uval[eU - 1]);
                    if (cc == c) { A::std_(&w.uval[eU - 1], o - pr); ufound = true; break; }
                    hU = (hU + 1) & (unsigned)hm;
                    eU = (unsigned)w.uh[hU];
                }
            }
            WP_T(ts1); WP_SUB(2, 4, tu2, ts1);
            const unsigned long long md = __ballot(valid && c == i);
            if (md != 0ull) wdiag = wdiag - wave_bcast_f64(pr, __ffsll((long long)md) - 1);
            // (bfound is LDS, in order per wave.  Waiting here for the stores of the update to be acknowledged, and again before and
            // inside the hash insertion, was three trips per elimination for nothing: what is appended below goes to new places, and
            // the one hand-over through memory -- a lane walking on from a cell another lane has just taken -- waits by itself)
            __builtin_amdgcn_wave_barrier();
            const bool nf = valid && c != i && (c < i ? bfound[lane] == 0 : !ufound);
            const bool isL = nf && c < i, isU = nf && c > i;
            const unsigned long long mL = __ballot(isL), mU = __ballot(isU);
            if (seq + __popcll(mL) > WpIdMax<IdT>::value) { wp_uh_clear<G, IdT>(w, hm, lane, nU, gscratch); return 1; }
            if (nL + __popcll(mL) > w.capL || nU + __popcll(mU) > w.capU) { if (!G && lane == 0) atomicAdd(&ctrl[nL + __popcll(mL) > w.capL ? 4 : 5], 1); wp_uh_clear<G, IdT>(w, hm, lane, nU, gscratch); return 1; }
            if (isL) {
                const int pos = nL + __popcll(mL & lt);
                const double fv = 0.0 - pr;
                A::sti(&w.lcol[pos], c); A::std_(&w.lval[pos], fv); A::sti(&w.lseq[pos], seq + __popcll(mL & lt));
                if (have && fv != 0.0 && !(fabs(fv) < thr1) && (unsigned)c < best) { best = (unsigned)c; bq = pos; }
            }
            if (isU) { const int pos = nU + __popcll(mU & lt); A::sti(&w.ucol[pos], c); A::std_(&w.uval[pos], 0.0 - pr); }
            wp_uh_insert_all<G, IdT>(w, hm, isU, c, nU + __popcll(mU & lt), hU);
            nL += __popcll(mL); seq += __popcll(mL); nU += __popcll(mU);
            A::sync();
            WP_T(ts2); WP_ACC(6, ts1, ts2);
        }
        WP_T(tp3); WP_ACC(2, tp2, tp3);
    }
    WP_T(tq0);
    // (10.-12.) dropping (ILUT.hpp:259,261).  The U row FIRST: it is what other rows wait for; the L row is nobody's dependency and is
    // selected behind the publication (the capacity test of its staging comes before anything is published)
    if (G && alt_lcol != nullptr && nK > w.capL && nK <= alt_capL) { nL = 0; WP_MOVE_POOL(); }      // (the pool is empty: only its home changes)
    if (nK > w.capL) { if (!G && lane == 0) atomicAdd(&ctrl[6], 1); wp_uh_clear<G, IdT>(w, hm, lane, nU, gscratch); return 1; }
    const size_t lb = (size_t)i * p;
    __builtin_amdgcn_s_waitcnt(0);
    // (12.) U row = (i, w[i]) then kept entries; every datum write-through, the length last is not required
    WP_T(tq0a);
    const size_t lbi = (size_t)i * ul_.si, lbv = (size_t)i * ul_.sv;
    const int nUk = wp_select<G, true, SELG>(lane, w.ucol, w.uval, nU, p - 1, tau, selq, gscratch, Urow_idx + lbi + 1, Urow_val + lbv + 1);
    WP_T(tq0b); WP_SUB(3, 4, tq0a, tq0b);
    if (lane == 0) {
        double piv = wdiag;
        if (piv == 0.0) atomicMin(&ctrl[2], i);                                  // ILUT.hpp:269-270 (reported after the sweep)
        if ((unsigned long long)__double_as_longlong(piv) == kSentinel) piv = __longlong_as_double((long long)kCanonNaN);
        st_agent_f64(&Urow_val[lbv], piv);
        st_agent_i32(&Urow_idx[lbi], i);
#ifdef ILUT_PROFILE
        st_agent_i32(&g_wp_lvl[i], prof_lvl + 1);
        g_wp_parent[i] = prof_parent; g_wp_lparent[i] = prof_lparent; g_wp_tstart[i] = prof_t0;
        g_wp_size[i] = (prof_maxl > 4095 ? 4095 : prof_maxl) | ((nU > 1023 ? 1023 : nU) << 12) | ((nK > 511 ? 511 : nK) << 22) | (G ? (1 << 31) : 0); g_wp_wait[i] = prof_wait; g_wp_tfin[i] = wall_clock64();
        __threadfence();
#endif
        st_agent_i32(&Ulen[(size_t)i * ul_.sl], nUk + 1);
        atomicAdd(&ctrl[7], 1);                                                  // rows finished (what a waiting wave watches)
    }
    A::sync();
    // (10.) the multipliers back in insertion order (the pool is empty now: its arrays take them)
    WP_T(tq0c);
    // (an entry's rank among the sequence numbers: those are read 64 at a time and compared out of the registers -- a load per
    // comparison was 68 us per row on C3)
    for (int q0 = 0; q0 < nK; q0 += 64) {
        const int q = q0 + lane;
        const int s = q < nK ? (int)w.kseq[q] : 0;
        int r = 0;
        for (int b2 = 0; b2 < nK; b2 += 64) {
            const int t = b2 + lane < nK ? (int)w.kseq[b2 + lane] : 0x7fffffff;
            const int rem = nK - b2 < 64 ? nK - b2 : 64;
            for (int l = 0; l < rem; ++l) r += (__builtin_amdgcn_readlane(t, l) < s) ? 1 : 0;
        }
        if (q < nK) {
            A::sti(&w.lcol[r], w.kcol[q]);
            A::std_(&w.lval[r], w.kval[q]);
        }
    }
    A::sync();
    WP_T(tq0d); WP_SUB(3, 5, tq0c, tq0d);
    // (11.) L row = kept entries then (i, 1.0)
    const int nLk = wp_select<G, false, SELG>(lane, w.lcol, w.lval, nK, p - 1, tau, selq, gscratch, Lrow_idx + lb, Lrow_val + lb);
    if (lane == 0) { Lrow_idx[lb + nLk] = i; Lrow_val[lb + nLk] = 1.0; Llen[i] = nLk + 1; }
    A::sync();
    WP_T(tq0e); WP_SUB(4, 4, tq0d, tq0e);
    wp_uh_clear<G, IdT>(w, hm, lane, nU, gscratch);
    WP_T(tq1); WP_ACC(3, tq0, tq1); WP_SUB(4, 5, tq0e, tq1);
    return 0;
}

// ctrl: [0] next row, [1] error (1 timeout, 3 capacity -> the host runs the largest capacity class), [2] smallest row with a zero pivot,
// [3..6] statistics, [7] rows finished
// (four waves per SIMD -- 128 VGPRs -- is what the launch of 16 waves per CU counts on; left alone the compiler took 156-169 registers for the
// scalars it spills into vector registers and three or two waves were resident: 370 -> 309 ms on C3.  The class with 16 KB of LDS per wave
// holds 10 waves per CU anyway and keeps its registers: that is the remark silenced here)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <int kWpCapU, int kWpHashLds>
#ifndef ILUT_WPE
#define ILUT_WPE 4
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ILUT_WPE, ILUT_WPE)))
k_ilut_rows_wp(int32_t n, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval,
               int32_t p, double tau, WpArrays gw, int *gscratch_all,
               int32_t *Lrow_idx, double *Lrow_val, int32_t *Llen,
               int32_t *Urow_idx, double *Urow_val, int32_t *Ulen, UrowLayout ul_, int32_t *ctrl, int tier2)
{
    constexpr int kWpCapL = kWpCapU;
    // One block of LDS per wave, carved twice.  Tier 1: pool, U slots and hash, kWpCapU entries each.  Tier 2 (a row that outgrew tier 1:
    // three of four on C3, mostly by their U part -- 508 U slots on average against a pool of at most 202): the whole block is the POOL,
    // which every elimination passes over several times, and the U slots, their hash and the kept list live in the wave's global arrays,
    // touched once per entry of a fetched row.  Tier 3: everything in global memory.
#ifndef ILUT_RAW
#define ILUT_RAW 7168
#endif
    constexpr int kRaw = kWpCapU == 128 ? ILUT_RAW : 14336;
    constexpr int kCap2 = kRaw / 14;                                         // 512 / 1024 pool entries (8 + 4 + 2 bytes each)
    static_assert(kWpCapU * 26 + kWpHashLds * 2 <= kRaw, "tier 1 does not fit the block");
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kRaw];
    __shared__ int s_selq[kWpSel];
    __shared__ int bcol[64], bfound[64], s_dlist[64];
    __shared__ double bpr[64];
    double *s_uval = reinterpret_cast<double *>(s_raw), *s_lval = s_uval + kWpCapU;
    int *s_ucol = reinterpret_cast<int *>(s_lval + kWpCapL), *s_lcol = s_ucol + kWpCapU;
    unsigned short *s_lseq = reinterpret_cast<unsigned short *>(s_lcol + kWpCapL), *s_uh = s_lseq + kWpCapL;
    double *t_lval = reinterpret_cast<double *>(s_raw);
    int *t_lcol = reinterpret_cast<int *>(t_lval + kCap2);
    unsigned short *t_lseq = reinterpret_cast<unsigned short *>(t_lcol + kCap2);
    const int lane = threadIdx.x;
    const size_t wv = blockIdx.x;
    WpArrays g = gw;
    g.uh += wv * (size_t)kWpHashG;
    g.ucol += wv * (size_t)gw.capU; g.uval += wv * (size_t)gw.capU;
    g.lcol += wv * (size_t)gw.capL; g.lval += wv * (size_t)gw.capL; g.lseq += wv * (size_t)gw.capL;
    g.kcol += wv * (size_t)gw.capK; g.kval += wv * (size_t)gw.capK; g.kseq += wv * (size_t)gw.capK;
    int *gscratch = gscratch_all + wv * (size_t)gw.capU;
    const WpArrays lw = {s_uh, kWpHashLds - 1, s_ucol, s_uval, kWpCapU, s_lcol, s_lval, s_lseq, kWpCapL, g.kcol, g.kval, g.kseq, gw.capK};
    const WpArrays hw = {g.uh, kWpHashG - 1, g.ucol, g.uval, gw.capU, t_lcol, t_lval, t_lseq, kCap2, g.kcol, g.kval, g.kseq, gw.capK};
    static_assert(kWpHashLds * 2 % 8 == 0 && (kWpCapU * 26) % 8 == 0, "the LDS hash is cleared in 8-byte words");
#ifdef ILUT_PROFILE
    unsigned long long prof[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#else
    unsigned long long *prof = nullptr;
#endif
    for (;;) {
        int i = 0;
        if (lane == 0) i = atomicAdd(&ctrl[0], 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= n) break;
        // (starting every row in tier 2 costs the same as trying tier 1 first: 494 against 496 ms on C3)
        WpResume rs;
        rs.active = 0;
        int rc = wp_row<false>(lane, i, n, p, tau, Aptr, Aidx, Aval, Lrow_idx, Lrow_val, Llen, Urow_idx, Urow_val, Ulen, ul_,
                               lw, bcol, bpr, bfound, s_dlist, s_selq, gscratch, ctrl, prof);
        rc = __builtin_amdgcn_readfirstlane(rc);
        if (rc == 1 && tier2) {
            if (lane == 0) atomicAdd(&ctrl[3], 1);          // statistics: rows that outgrew tier 1
            rc = wp_row<true>(lane, i, n, p, tau, Aptr, Aidx, Aval, Lrow_idx, Lrow_val, Llen, Urow_idx, Urow_val, Ulen, ul_,
                              hw, bcol, bpr, bfound, s_dlist, s_selq, gscratch, ctrl, prof, &rs, g.lcol, g.lval, g.lseq, gw.capL);
            rc = __builtin_amdgcn_readfirstlane(rc);
            if ((rc == 1 || rc == 3) && lane == 0) atomicAdd(&ctrl[24], 1);      // ... and tier 2
        } else if (rc == 1) {
            if (lane == 0) atomicAdd(&ctrl[3], 1);
        }
        if (rc == 1 || rc == 3) {
            // (3: the row goes on where it was, its pool in the global arrays now)
            if (rc == 1) rs.active = 0;
            rc = wp_row<true>(lane, i, n, p, tau, Aptr, Aidx, Aval, Lrow_idx, Lrow_val, Llen, Urow_idx, Urow_val, Ulen, ul_,
                              g, bcol, bpr, bfound, s_dlist, s_selq, gscratch, ctrl, prof, &rs);
            rc = __builtin_amdgcn_readfirstlane(rc);
        }
        if (rc != 0) {
            // give up: publish a poisoned row so that nobody waits for it, and report
            if (lane == 0) {
                atomicMax(&ctrl[1], rc == 1 ? 3 : 1);
                st_agent_f64(&Urow_val[(size_t)i * ul_.sv], 1.0);
                st_agent_i32(&Urow_idx[(size_t)i * ul_.si], i);
                st_agent_i32(&Ulen[(size_t)i * ul_.sl], 1);
                Llen[i] = 0;
            }
            // (the row left its cells in the wave's table: the rows this wave still takes start from an empty one)
            for (size_t q = lane; q < (size_t)kWpHashG * sizeof(unsigned short) / 8; q += 64) reinterpret_cast<unsigned long long *>(g.uh)[q] = 0ull;
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
#ifdef ILUT_PROFILE
    if (lane == 0)
        for (int q = 0; q < 8; ++q) atomicAdd(reinterpret_cast<unsigned long long *>(ctrl + 8) + q, prof[q]);
#endif
}

#pragma clang diagnostic pop

// the largest capacity class: every piece as long as the matrix is wide, 32-bit slot ids and sequence numbers, the selection
// queue in global memory too (budgets beyond kWpSel) -- rows of any length and any fill budget, on a few waves
__global__ void __launch_bounds__(64)
k_ilut_rows_wp_big(int32_t n, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval,
                   int32_t p, double tau, WpArraysT<unsigned int> gw, int *gscratch_all, int *selq_all,
                   int32_t *Lrow_idx, double *Lrow_val, int32_t *Llen,
                   int32_t *Urow_idx, double *Urow_val, int32_t *Ulen, UrowLayout ul_, int32_t *ctrl)
{
    __shared__ int bcol[64], bfound[64], s_dlist[64];
    __shared__ double bpr[64];
    const int lane = threadIdx.x;
    const size_t wv = blockIdx.x;
    WpArraysT<unsigned int> g = gw;
    g.uh += wv * (size_t)(gw.hmask + 1);
    g.ucol += wv * (size_t)gw.capU; g.uval += wv * (size_t)gw.capU;
    g.lcol += wv * (size_t)gw.capL; g.lval += wv * (size_t)gw.capL; g.lseq += wv * (size_t)gw.capL;
    g.kcol += wv * (size_t)gw.capK; g.kval += wv * (size_t)gw.capK; g.kseq += wv * (size_t)gw.capK;
    int *gscratch = gscratch_all + wv * (size_t)gw.capU;
#ifdef ILUT_PROFILE
    unsigned long long prof[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#else
    unsigned long long *prof = nullptr;
#endif
    int *selq = selq_all + wv * (size_t)(p + 1);
    for (;;) {
        int i = 0;
        if (lane == 0) i = atomicAdd(&ctrl[0], 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= n) break;
        int rc = wp_row<true, unsigned int, true>(lane, i, n, p, tau, Aptr, Aidx, Aval, Lrow_idx, Lrow_val, Llen, Urow_idx, Urow_val, Ulen, ul_,
                                                  g, bcol, bpr, bfound, s_dlist, selq, gscratch, ctrl, prof);
        rc = __builtin_amdgcn_readfirstlane(rc);
        if (rc != 0) {
            // give up: publish a poisoned row so that nobody waits for it, and report
            if (lane == 0) {
                atomicMax(&ctrl[1], rc == 1 ? 3 : 1);
                st_agent_f64(&Urow_val[(size_t)i * ul_.sv], 1.0);
                st_agent_i32(&Urow_idx[(size_t)i * ul_.si], i);
                st_agent_i32(&Ulen[(size_t)i * ul_.sl], 1);
                Llen[i] = 0;
            }
            for (size_t q = lane; q < (size_t)(gw.hmask + 1) * sizeof(unsigned int) / 8; q += 64) reinterpret_cast<unsigned long long *>(g.uh)[q] = 0ull;
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
}

// every record starts as "not there": length 0, columns -1, values the sentinel
__global__ void k_urec_init(const long long words, const int rw, const int vw, const int p, unsigned long long *__restrict__ rec)
{
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    const int o = (int)(w % rw);                                  // 8-byte word of its record
    rec[w] = o == 0 ? 0xffffffff00000000ull : o < vw ? ~0ull : o < vw + p ? kSentinel : 0ull;
}
static void wp_init_slabs(hipStream_t st, int32_t n, int32_t p, double *Urv, int32_t *Ulen, UrowLayout ul_, int32_t *ctrl)
{
    const long long words = (long long)n * ul_.sv;
    const int vw = (int)(Urv - reinterpret_cast<double *>(Ulen));
    hipLaunchKernelGGL(k_urec_init, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, words, ul_.sv, vw, p,
                       reinterpret_cast<unsigned long long *>(Ulen));
    ILUPP_HIP(hipGetLastError());
    const int32_t init[32] = {0, 0, 0x7fffffff};
    ILUPP_HIP(hipMemcpyAsync(ctrl, init, 128, hipMemcpyHostToDevice, st));
}
struct WpEvents { hipEvent_t a = nullptr, b = nullptr; ~WpEvents() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } };

// 0 = rows computed, 1 = not even this class has room (a matrix too wide for the memory budget), ILUPP_ERR_TIMEOUT
static int ilut_rows_wp_big(hipStream_t st, const DevMat &A, int32_t p, double threshold,
                            int32_t *Lri, double *Lrv, int32_t *Llen, int32_t *Uri, double *Urv, int32_t *Ulen, UrowLayout ul_, int32_t *ctrl, float *kernel_ms)
{
    const int32_t n = A.n;
    const int cap = n + 64;
    size_t hashN = 1024;
    while (hashN < 2 * (size_t)cap) hashN *= 2;
    const size_t per_wave = hashN * 4 + (size_t)cap * (12 + 16 + 16 + 4) + (size_t)(p + 1) * 4;
    int workers = device_cu_count() * 4;
    while (workers > 1 && (size_t)workers * per_wave > ((size_t)16 << 30)) workers >>= 1;
    if ((size_t)workers * per_wave > ((size_t)64 << 30)) return 1;
    if (workers > n) workers = n;
    WpArraysT<unsigned int> g = {nullptr, (int)hashN - 1, nullptr, nullptr, cap, nullptr, nullptr, nullptr, cap, nullptr, nullptr, nullptr, cap};
    PoolBlock b_uh, b_ucol, b_uval, b_lcol, b_lval, b_lseq, b_kcol, b_kval, b_kseq, b_scr, b_selq;
    ILUPP_HIP(b_uh.alloc(sizeof(unsigned int) * (size_t)workers * hashN));
    ILUPP_HIP(b_ucol.alloc(sizeof(int) * (size_t)workers * cap));
    ILUPP_HIP(b_uval.alloc(sizeof(double) * (size_t)workers * cap));
    ILUPP_HIP(b_lcol.alloc(sizeof(int) * (size_t)workers * cap));
    ILUPP_HIP(b_lval.alloc(sizeof(double) * (size_t)workers * cap));
    ILUPP_HIP(b_lseq.alloc(sizeof(unsigned int) * (size_t)workers * cap));
    ILUPP_HIP(b_kcol.alloc(sizeof(int) * (size_t)workers * cap));
    ILUPP_HIP(b_kval.alloc(sizeof(double) * (size_t)workers * cap));
    ILUPP_HIP(b_kseq.alloc(sizeof(unsigned int) * (size_t)workers * cap));
    ILUPP_HIP(b_scr.alloc(sizeof(int) * (size_t)workers * cap));
    ILUPP_HIP(b_selq.alloc(sizeof(int) * (size_t)workers * (size_t)(p + 1)));
    g.uh = b_uh.as<unsigned int>(); g.ucol = b_ucol.as<int>(); g.uval = b_uval.as<double>();
    ILUPP_HIP(hipMemsetAsync(g.uh, 0, sizeof(unsigned int) * (size_t)workers * hashN, st));
    g.lcol = b_lcol.as<int>(); g.lval = b_lval.as<double>(); g.lseq = b_lseq.as<unsigned int>();
    g.kcol = b_kcol.as<int>(); g.kval = b_kval.as<double>(); g.kseq = b_kseq.as<unsigned int>();
    wp_init_slabs(st, n, p, Urv, Ulen, ul_, ctrl);
    WpEvents ev;
    ILUPP_HIP(hipEventCreate(&ev.a));
    ILUPP_HIP(hipEventCreate(&ev.b));
    ILUPP_HIP(hipEventRecord(ev.a, st));
    hipLaunchKernelGGL(k_ilut_rows_wp_big, dim3((unsigned)workers), dim3(64), 0, st, n, A.ptr, A.idx, A.val, p, threshold, g, b_scr.as<int>(),
                       b_selq.as<int>(), Lri, Lrv, Llen, Uri, Urv, Ulen, ul_, ctrl);
    ILUPP_HIP(hipEventRecord(ev.b, st));
    ILUPP_HIP(hipGetLastError());
    int32_t h[8];
    ILUPP_HIP(hipMemcpyAsync(h, ctrl, 32, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, ev.a, ev.b));
    if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] ilut_wp: largest capacity class, %d waves, status %d, kernel %.3f ms\n", workers, h[1], kernel_ms ? *kernel_ms : 0.f);
    if (h[1] == 3) return 1;
    if (h[1] == 1) return ILUPP_ERR_TIMEOUT;
    return 0;
}

// returns 0 = rows computed (ctrl holds zero-pivot info), 1 = a row that fits no capacity class, ILUPP_ERR_TIMEOUT
int ilut_rows_wp(hipStream_t st, const DevMat &A, int32_t p, double threshold,
                 int32_t *Lri, double *Lrv, int32_t *Llen, int32_t *Uri, double *Urv, int32_t *Ulen, UrowLayout ul_, int32_t *ctrl, float *kernel_ms)
{
    const int32_t n = A.n;
    const bool force_big = getenv("ILUPP_ILUT_BIG") != nullptr;                // (tests: A/B of the capacity classes)
    // fill budgets beyond the LDS selection queue: the largest class at once
    if (p - 1 >= kWpSel || force_big) return ilut_rows_wp_big(st, A, p, threshold, Lri, Lrv, Llen, Uri, Urv, Ulen, ul_, ctrl, kernel_ms);
    const bool small_pieces = p <= 32;
    static const int waves_env = getenv("ILUPP_ILUT_WAVES") ? atoi(getenv("ILUPP_ILUT_WAVES")) : 0;     // (experiments: waves per CU)
    int workers = device_cu_count() * (waves_env > 0 ? waves_env : (small_pieces ? 16 : 8));          // (what the LDS block of a wave lets a CU hold)
    if (workers > n) workers = n;
    WpArrays g = {nullptr, kWpHashG - 1, nullptr, nullptr, kWpGCapU, nullptr, nullptr, nullptr, kWpGCapL, nullptr, nullptr, nullptr, kWpGCapK};
    int32_t h[32];
    {
        PoolBlock b_uh, b_ucol, b_uval, b_lcol, b_lval, b_lseq, b_kcol, b_kval, b_kseq, b_scr;
        ILUPP_HIP(b_uh.alloc(sizeof(unsigned short) * (size_t)workers * kWpHashG));
        ILUPP_HIP(b_ucol.alloc(sizeof(int) * (size_t)workers * g.capU));
        ILUPP_HIP(b_uval.alloc(sizeof(double) * (size_t)workers * g.capU));
        ILUPP_HIP(b_lcol.alloc(sizeof(int) * (size_t)workers * g.capL));
        ILUPP_HIP(b_lval.alloc(sizeof(double) * (size_t)workers * g.capL));
        ILUPP_HIP(b_lseq.alloc(sizeof(unsigned short) * (size_t)workers * g.capL));
        ILUPP_HIP(b_kcol.alloc(sizeof(int) * (size_t)workers * g.capK));
        ILUPP_HIP(b_kval.alloc(sizeof(double) * (size_t)workers * g.capK));
        ILUPP_HIP(b_kseq.alloc(sizeof(unsigned short) * (size_t)workers * g.capK));
        ILUPP_HIP(b_scr.alloc(sizeof(int) * (size_t)workers * g.capU));
        g.uh = b_uh.as<unsigned short>(); g.ucol = b_ucol.as<int>(); g.uval = b_uval.as<double>();
        ILUPP_HIP(hipMemsetAsync(g.uh, 0, sizeof(unsigned short) * (size_t)workers * kWpHashG, st));
        g.lcol = b_lcol.as<int>(); g.lval = b_lval.as<double>(); g.lseq = b_lseq.as<unsigned short>();
        g.kcol = b_kcol.as<int>(); g.kval = b_kval.as<double>(); g.kseq = b_kseq.as<unsigned short>();
        int *gscratch = b_scr.as<int>();
        wp_init_slabs(st, n, p, Urv, Ulen, ul_, ctrl);
#ifdef ILUT_PROFILE
        PoolBlock pb_lvl, pb_par, pb_fin, pb_start, pb_wait, pb_lpar;
        ILUPP_HIP(pb_lpar.alloc(sizeof(int) * (size_t)n));
        { int *f = pb_lpar.as<int>(); ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wp_lparent), &f, sizeof(f))); }
        PoolBlock pb_size;
        ILUPP_HIP(pb_size.alloc(sizeof(int) * (size_t)n));
        { int *f = pb_size.as<int>(); ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wp_size), &f, sizeof(f))); }
        ILUPP_HIP(pb_lvl.alloc(sizeof(int) * (size_t)n)); ILUPP_HIP(pb_par.alloc(sizeof(int) * (size_t)n));
        ILUPP_HIP(pb_fin.alloc(sizeof(long long) * (size_t)n)); ILUPP_HIP(pb_start.alloc(sizeof(long long) * (size_t)n)); ILUPP_HIP(pb_wait.alloc(sizeof(long long) * (size_t)n));
        ILUPP_HIP(hipMemsetAsync(pb_lvl.p, 0, sizeof(int) * (size_t)n, st));
        {
            int *a = pb_lvl.as<int>(), *b = pb_par.as<int>();
            long long *c = pb_fin.as<long long>(), *d = pb_start.as<long long>(), *e = pb_wait.as<long long>();
            ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wp_lvl), &a, sizeof(a))); ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wp_parent), &b, sizeof(b)));
            ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wp_tfin), &c, sizeof(c))); ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wp_tstart), &d, sizeof(d)));
            ILUPP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wp_wait), &e, sizeof(e)));
        }
        ILUPP_HIP(hipMemsetAsync(ctrl + 8, 0, 64, st));
#endif
        WpEvents ev;
        ILUPP_HIP(hipEventCreate(&ev.a));
        ILUPP_HIP(hipEventCreate(&ev.b));
        ILUPP_HIP(hipEventRecord(ev.a, st));
        static const int cap_env = getenv("ILUPP_ILUT_CAP") ? atoi(getenv("ILUPP_ILUT_CAP")) : 0;           // (experiments: 256 = the larger LDS class for every budget)
        static const int tier2 = getenv("ILUPP_ILUT_NO_TIER2") ? 0 : 1;                                     // (tests, A/B: rows that outgrow LDS go to global memory at once)
        if (small_pieces && cap_env != 256)
            hipLaunchKernelGGL((k_ilut_rows_wp<128, 512>), dim3((unsigned)workers), dim3(64), 0, st, n, A.ptr, A.idx, A.val, p, threshold, g, gscratch,
                               Lri, Lrv, Llen, Uri, Urv, Ulen, ul_, ctrl, tier2);
        else
            hipLaunchKernelGGL((k_ilut_rows_wp<256, 1024>), dim3((unsigned)workers), dim3(64), 0, st, n, A.ptr, A.idx, A.val, p, threshold, g, gscratch,
                               Lri, Lrv, Llen, Uri, Urv, Ulen, ul_, ctrl, tier2);
        ILUPP_HIP(hipEventRecord(ev.b, st));
        ILUPP_HIP(hipGetLastError());
        ILUPP_HIP(hipMemcpyAsync(h, ctrl, 128, hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, ev.a, ev.b));
#ifdef ILUT_PROFILE
        if (h[1] == 0) {
            unsigned long long t[8];
            ILUPP_HIP(hipMemcpy(t, ctrl + 8, 64, hipMemcpyDeviceToHost));
            float kms = 0.f;
            ILUPP_HIP(hipEventElapsedTime(&kms, ev.a, ev.b));
            const double tick = 1e-8;            // wall_clock64: 100 MHz
            const double wave_s = (double)workers * kms * 1e-3;
            fprintf(stderr, "[ilut profile] n %d, p %d, %d waves (%d per CU), kernel %.3f ms: wave-seconds %.2f\n", n, p, workers, workers / device_cu_count(), kms, wave_s);
            const char *nm[8] = {"pop the next column (minimum over the pool) + stage-1 drops", "wait for / fetch the U row", "update with the U row", "drop, sort, store the row",
                                 "", "", "  (of the update: append fill, hash)", "scatter A's row, clear the hash"};
            for (int q : {7, 0, 1, 2, 6, 3}) fprintf(stderr, "[ilut profile]   %-62s %9.3f wave-s  %5.1f %%\n", nm[q], t[q] * tick, 100.0 * t[q] * tick / wave_s);
            fprintf(stderr, "[ilut profile]   U rows fetched %llu (%.1f per row), retries while waiting %llu\n", t[5], (double)t[5] / n, t[4]);
            std::vector<int> lvl((size_t)n), par((size_t)n);
            std::vector<long long> fin((size_t)n), sta((size_t)n), wai((size_t)n);
            ILUPP_HIP(hipMemcpy(lvl.data(), pb_lvl.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
            ILUPP_HIP(hipMemcpy(par.data(), pb_par.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
            ILUPP_HIP(hipMemcpy(fin.data(), pb_fin.p, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost));
            ILUPP_HIP(hipMemcpy(sta.data(), pb_start.p, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost));
            ILUPP_HIP(hipMemcpy(wai.data(), pb_wait.p, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost));
            int depth = 0, last = 0;
            long long tmin = sta[0], tmax = fin[0];
            double busy = 0.0, waited = 0.0;
            size_t nwait = 0;
            for (int r = 0; r < n; ++r) {
                depth = lvl[r] > depth ? lvl[r] : depth;
                if (fin[r] > tmax) { tmax = fin[r]; last = r; }
                if (sta[r] < tmin) tmin = sta[r];
                busy += (double)(fin[r] - sta[r]); waited += (double)wai[r];
                nwait += par[r] >= 0 ? 1 : 0;
            }
            fprintf(stderr, "[ilut profile]   dependency depth of the factor (levels) %d; first start -> last finish %.3f ms\n", depth, (tmax - tmin) * tick * 1e3);
            fprintf(stderr, "[ilut profile]   a row is in its wave %.1f us on average, %.1f us of that waiting for rows not finished yet (%.1f %% of the rows wait at all)\n",
                    busy / n * tick * 1e6, waited / n * tick * 1e6, 100.0 * nwait / n);
            // the chain of last-awaited rows back from the row that finished last
            int links = 0; double chain_wait = 0.0, chain_own = 0.0;
            std::vector<double> link_us;
            for (int r = last; r >= 0;) {
                const int q = par[r];
                if (q < 0) { chain_own += (double)(fin[r] - sta[r]); break; }
                const double d = (double)(fin[r] - fin[q]);             // from the awaited row's publication to this row's
                link_us.push_back(d * tick * 1e6);
                chain_own += d; chain_wait += (double)wai[r];
                ++links; r = q;
            }
            std::sort(link_us.begin(), link_us.end());
            fprintf(stderr, "[ilut profile]   critical chain (row %d back through the rows it last waited for): %d links, %.3f ms = %.1f %% of the kernel; a link (awaited row published -> this row published) "
                            "median %.1f us, mean %.1f us, p90 %.1f us; its rows waited %.3f ms in all\n", last, links, chain_own * tick * 1e3, 100.0 * chain_own * tick * 1e3 / kms,
                    link_us.empty() ? 0.0 : link_us[link_us.size() / 2], links ? chain_own * tick * 1e6 / links : 0.0, link_us.empty() ? 0.0 : link_us[link_us.size() * 9 / 10],
                    chain_wait * tick * 1e3);
            {
                // the chain of DEEPEST dependencies back from a row of the last level: one row per level
                std::vector<int> lpar((size_t)n);
                ILUPP_HIP(hipMemcpy(lpar.data(), pb_lpar.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
                int deepest = 0;
                for (int r = 0; r < n; ++r) if (lvl[r] > lvl[deepest] || (lvl[r] == lvl[deepest] && fin[r] > fin[deepest])) deepest = r;
                std::vector<double> gap, own, wt;
                int cnt = 0, root = deepest;
                for (int r = deepest; r >= 0 && lpar[r] >= 0; r = lpar[r]) {
                    const int q = lpar[r];
                    gap.push_back((double)(fin[r] - fin[q]) * tick * 1e6);
                    own.push_back((double)(fin[r] - sta[r]) * tick * 1e6);
                    wt.push_back((double)wai[r] * tick * 1e6);
                    ++cnt; root = q;
                }
                auto mean = [](const std::vector<double> &v) { double z = 0; for (double x : v) z += x; return v.empty() ? 0.0 : z / v.size(); };
                {
                    // the working rows' sizes: all rows / the rows of the chain
                    std::vector<int> sz((size_t)n);
                    ILUPP_HIP(hipMemcpy(sz.data(), pb_size.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
                    auto stats = [&](const char *what, const std::vector<int> &rows) {
                        double pl = 0, us = 0, el = 0, gl = 0; int pmax = 0, umax = 0, p128 = 0, p512 = 0, u128 = 0;
                        for (int r : rows) {
                            const unsigned v = (unsigned)sz[(size_t)r];
                            const int a = v & 4095, b = (v >> 12) & 1023, c = (v >> 22) & 511;
                            pl += a; us += b; el += c; gl += v >> 31;
                            pmax = a > pmax ? a : pmax; umax = b > umax ? b : umax;
                            p128 += a > 128; p512 += a > 512; u128 += b > 128;
                        }
                        const double m = rows.empty() ? 1.0 : (double)rows.size();
                        fprintf(stderr, "[ilut profile]     %-18s %8zu rows: pool max %.0f on average (largest %d; > 128: %.1f %%, > 512: %.1f %%), U slots %.0f (largest >= %d; > 128: %.1f %%), eliminations %.0f, in the global pieces %.1f %%\n",
                                what, rows.size(), pl / m, pmax, 100.0 * p128 / m, 100.0 * p512 / m, us / m, umax, 100.0 * u128 / m, el / m, 100.0 * gl / m);
                    };
                    std::vector<int> all((size_t)n), ch;
                    for (int r = 0; r < n; ++r) all[(size_t)r] = r;
                    for (int r = deepest; r >= 0 && lpar[r] >= 0; r = lpar[r]) ch.push_back(r);
                    stats("all rows", all);
                    stats("rows of the chain", ch);
                }
                std::vector<double> g2 = gap; std::sort(g2.begin(), g2.end());
                fprintf(stderr, "[ilut profile]   chain of deepest dependencies: row %d (level %d) back to row %d: %d links; from the first row's publication to the last's %.3f ms = %.1f %% of the kernel\n",
                        deepest, lvl[deepest], root, cnt, (double)(fin[deepest] - fin[root]) * tick * 1e3, 100.0 * (double)(fin[deepest] - fin[root]) * tick * 1e3 / kms);
                fprintf(stderr, "[ilut profile]     a link (the deepest dependency published -> this row published): mean %.1f us, median %.1f us, p90 %.1f us; the rows of the chain are %.1f us in their waves, %.1f us of that waiting\n",
                        mean(gap), g2.empty() ? 0.0 : g2[g2.size() / 2], g2.empty() ? 0.0 : g2[g2.size() * 9 / 10], mean(own), mean(wt));
            }
            fprintf(stderr, "[ilut profile]   throughput bound: %.2f wave-s of non-waiting work / %d waves = %.1f ms; latency bound: the chain above\n",
                    (busy - waited) * tick, workers, (busy - waited) * tick / workers * 1e3);
        }
#endif
    }
    if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] ilut_wp: %d of %d rows outgrew LDS (pool %d, U slots %d, kept %d), %d of them the pool-in-LDS tier too, status %d, kernel %.3f ms\n", h[3], n, h[4], h[5], h[6], h[24], h[1], kernel_ms ? *kernel_ms : 0.f);
    // a row that outgrew the 64 K pieces: the whole factorisation once more in the largest class
    if (h[1] == 3) return ilut_rows_wp_big(st, A, p, threshold, Lri, Lrv, Llen, Uri, Urv, Ulen, ul_, ctrl, kernel_ms);
    if (h[1] == 1) return ILUPP_ERR_TIMEOUT;
    return 0;
}

}  // namespace ilupp
