// ilupp_amd/csrc/st_direct.hip -- ILU(0) on the static level-major form, fed from A's CSR values directly (gfx950).
//
// st.hip's factor kernel reads level-major factor records that a rows pass (k_st_rows) makes from A: A -> records -> kernel is
// 5.75 GB of traffic for 3.21 GB of work, and the rows pass is the longest kernel of a construction.  Here the records are gone.
// What makes that possible is a stronger statement about the lanes than st.hip's (proven row by row, pattern only: k_sd_proof):
//
//   a lane's rows (a chain of consecutive rows, CSR rows first .. first+cnt-1) ALL have exactly the lane's template entries --
//   except that the own-chain entries are missing where the chain ends (column r-1 in the lane's first row, r+1 in its last).
//
// Then a lane's part of A's value array is ONE contiguous stream with a fixed pitch of m entries per row (one entry less in the
// first row), nothing about a row has to be looked up, and the column indices are never read again:
//
//   * PRODUCER waves (3 per workgroup) copy, for every lane and every block of two steps, the 128 bytes of A.val that hold the
//     lane's two rows -- 8 threads x 16 bytes per lane, aligned pieces, through a register file that holds the pieces of the next
//     4 blocks (the read-ahead) -- verbatim into an LDS ring of two blocks;
//   * the CONSUMER lanes (4 waves, one row per lane and step, as in st.hip) read their row from the ring at lane-constant
//     addresses: entries left of the diagonal from the row's start, the diagonal and the entries right of it from the diagonal's
//     place (the two rows at a chain's ends shift one of the two by one entry);
//   * the transposed entry a(k,r) an elimination needs (ILU0.hpp:8-23 on such rows: u_rr -= (a_rk / u_kk) a_kr) is an entry RIGHT
//     of the diagonal of the pivot row k, i.e. something the lane of row k knows as soon as it has read its row: the hand-off
//     array in LDS carries, next to the pivots, every row's three entries right of the diagonal.  For pivot rows of earlier
//     workgroups the courier wave reads a(k,r) from A itself (it polls the pivot a few steps ahead anyway);
//   * behind the barrier of a step a lane reads its (at most three) PIVOTS and nothing else: its own row and the transposed entries
//     were read during the step before (the row's piece of the ring is complete one step early; the entries right of the diagonal
//     are handed over as soon as their row has been read, a step before its pivot exists).  The chain of a step is then
//     barrier -> three 8-byte LDS reads -> three divisions -> one 8-byte LDS write;
//   * the kernel writes the complete records of both sweeps (l_rk and 1; A's entries right of the diagonal and u_rr).
//
// Per row: 56 B of A read, 64 B of records written -- the algorithmic traffic of SURVEY section 8(d) without the column indices.
// Arithmetic, order of operations and the records' formats are st.hip's (bit-identical results; tests A/B the two).
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "st_common.h"

#define SD_STORE(v, p) __builtin_nontemporal_store(v, p)

#ifndef SD_CSLEEP
#define SD_CSLEEP 1
#endif
namespace ilupp {

static constexpr int kSdH = kSdHist;                               // hand-off slots: an in-workgroup dependency lies at most kSdH-1 steps back
static constexpr int kSdHoU = (kThreads + 64) * 8;                 // a slot, first part: the lanes' pivots, then the courier's
static constexpr int kSdHoAt = 24;                                 // second part, bytes per lane: its row's three entries right of the diagonal
static constexpr int kSdHoRow = kSdHoU + kThreads * kSdHoAt + 64 * 8;      // ... then the courier's transposed entries
static constexpr int kSdPitch = 136;                               // bytes of a lane's piece of a block (128 loaded; 17 x 8: conflict-free 8-byte reads)
static constexpr int kSdRing = kThreads * kSdPitch;                // a block of two steps
static constexpr int kSdProd = 3;                                  // producer waves
static constexpr int kSdPer = 11;                                  // groups of 8 lanes per producer wave (3 x 11 x 8 >= 256)
static constexpr int kSdRA = 4;                                    // blocks the producers read ahead
static constexpr int kSdThreads = kThreads + 64 + 64 * kSdProd;
static constexpr int kSdLds = 2 * kSdRing + kSdH * kSdHoRow;
static constexpr unsigned kSdOob = 0xfffffff0u;                    // a buffer offset beyond any array: the load returns zeros and touches nothing
static_assert(kSdProd * kSdPer * 8 >= kThreads, "every lane needs a producer");
static_assert(kSdHoRow % 8 == 0 && kSdRing % 8 == 0, "alignment of the LDS regions");
static constexpr unsigned kSdHo = 2u * kSdRing;                    // where the hand-off slots begin

#ifdef SD_STAMP
// diagnostics build only: 100 MHz ticks of one consumer wave ([0..7]) and one producer wave ([16..23]) of workgroup SD_STAMP_WG, per segment of a step
__device__ unsigned long long g_sd_stamp[32];
#ifndef SD_STAMP_WG
#define SD_STAMP_WG 120
#endif
#define SD_T(i) do { if (stamp_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_[i] += n_ - last_; last_ = n_; } } while (0)
#define SD_T_DECL(cond) const bool stamp_on = (cond); unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long last_ = __builtin_amdgcn_s_memtime(); unsigned long long nst_ = 0
#define SD_T_END(off) do { if (stamp_on && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 6; ++i_) g_sd_stamp[(off) + i_] = acc_[i_]; g_sd_stamp[(off) + 6] = nst_; } } while (0)
#else
#define SD_T(i) do { } while (0)
#define SD_T_DECL(cond) do { } while (0)
#define SD_T_END(off) do { } while (0)
#endif

struct SdArgs {
    const int32_t *ltab, *wtab;               // forward schedule
    const double *val;                        // A's values, the pointer rounded down to 16 bytes
    uint32_t val_bytes;                       // bytes readable from there
    int32_t val_shift;                        // what the rounding took off (0 or 8)
    v2d *pkL, *pkU;                           // 2 x 64 x 16 B per chunk: {l0,l1}{l2,1} / {u1,u2}{u3,u0}, both in the forward schedule's order
    const int32_t *xe, *xw;                   // the forward schedule's exchange between workgroups: pivots of exported lanes
    double *xch;
    int32_t *ctrl;                            // [0] ticket, [1] error
};

// a (lane, dependency) pair whose pivot row belongs to an earlier workgroup: the pivot of the consumer's row k is element
// idx0 + (k + sk) * stride of the exchange; the transposed entry is 8 bytes at at0 + k * atm (- sh in the producer's last row)
struct SdPair { int idx0, stride, sk, cnt; unsigned at0; int atm, klast, sh, hasT; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sd_rsrc(const SdArgs &A)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.val), 0, (int)A.val_bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------------
// the 256 lanes of the schedule
// ---------------------------------------------------------------------------------------------
template <bool EX>
__device__ __forceinline__ void sd_consumer(const SdArgs &A, unsigned char *lds, const int wg, const unsigned (&Ru)[3][kSdH], const unsigned (&Rt)[3][kSdH],
                                            const bool (&hasT)[3], const int tlo, const int thi)
{
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], sk = T[ST_SKEW], nd = T[ST_ND], fl = T[ST_DFL], p0 = T[ST_P0];
    const int ndU = fl & 3, ownL = (fl >> 2) & 1, ownU = (fl >> 3) & 1, m = fl >> 4;
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    const int xe = A.xe[slot];
    const bool exports = cnt > 0 && xe >= 0;
    const int xE = __builtin_amdgcn_readfirstlane(A.xw[wg * 4]);
    const int xoff = A.xw[wg * 4 + 3] - A.xw[wg * 4 + 1] * xE + xe;          // + step * xE: where this lane's pivot of a step goes
    unsigned char *pl = reinterpret_cast<unsigned char *>(A.pkL);
    unsigned char *pu = reinterpret_cast<unsigned char *>(A.pkU);
    const unsigned lo16 = (unsigned)ln * 16u;
    // where the lane's rows sit in its piece of a block: the piece starts at the 16-byte boundary at or below the (virtual) start
    // of the block's first row; row start of the second row: 8 m further
    const unsigned Cu = 8u * (unsigned)(p0 - ownL - sk * m) + (unsigned)A.val_shift;
    const unsigned rowA = (unsigned)t * kSdPitch + (Cu & 15u);
    const unsigned aL_[2] = {rowA, rowA + 8u * (unsigned)m};                             // entries left of the diagonal: + 8 j
    const unsigned aD_[2] = {rowA + 8u * (unsigned)nd, rowA + 8u * (unsigned)(m + nd)};  // the diagonal; right of it: + 8 (1 + q)
    const int k0L = ownL ? 0 : -1;                          // the row without its own-chain entry on the left / on the right
    const int kEU = ownU ? cnt - 1 : -1;
    const unsigned hoU = kSdHo + (unsigned)t * 8u;                                        // this lane's pivot: + slot * kSdHoRow
    const unsigned hoA = kSdHo + kSdHoU + (unsigned)t * kSdHoAt;                          // ... its entries right of the diagonal
    const bool inL[3] = {0 < nd, 1 < nd, 2 < nd}, inU[3] = {0 < ndU, 1 < ndU, 2 < ndU};
    const bool lastL[3] = {nd == 1, nd == 2, nd == 3};
    const double absent = st_dbl(kAbsent);

    // the row of a step (its seven values and the transposed entries of its eliminations) is read during the step BEFORE
    double nav[3], nd_, nup[3], nat[3];
#define SD_PREREAD(k_, ui_, par_, ri_)                                                                  \
    do {                                                                                                \
        const unsigned aLn_ = aL_[ui_] + ((k_) == k0L ? 8u : 0u), aUn_ = aD_[ui_] - ((k_) == kEU ? 8u : 0u); \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) nav[j] = st_lds(lds, aLn_ + (par_) + 8u * j);     \
        nd_ = st_lds(lds, aD_[ui_] + (par_));                                                           \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) nup[q] = st_lds(lds, aUn_ + (par_) + 8u + 8u * q); \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) nat[j] = st_lds(lds, Rt[j][ri_]);                  \
    } while (0)
    // values that look like one of the two markers (NaNs with a payload no arithmetic makes) are canonicalised, as the rows pass of
    // st.hip does; then the entries right of the diagonal go to the hand-off slot of their step
#define SD_CLEAN_AND_HAND(ri_)                                                                          \
    do {                                                                                                \
        bool nan_ = nd_ != nd_;                                                                         \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) nan_ = nan_ || nav[j] != nav[j] || nup[j] != nup[j] || nat[j] != nat[j]; \
        if (__any(nan_)) {                                                                              \
            nd_ = st_clean(nd_);                                                                        \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) { nav[j] = st_clean(nav[j]); nup[j] = st_clean(nup[j]); nat[j] = st_clean(nat[j]); } \
        }                                                                                               \
        _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                   \
            *reinterpret_cast<double *>(lds + hoA + (unsigned)(ri_) * kSdHoRow + 8u * q) = nup[q];      \
    } while (0)

    ST_BARRIER();                                           // (the producers' first block and the courier's first entries are in place)
    SD_PREREAD(tlo - sk, 0, 0u, 0);
    SD_CLEAN_AND_HAND(0);
    SD_T_DECL(wv == 0 && wg == SD_STAMP_WG);
    for (int tb = tlo; tb < thi; tb += 8) {
        const int kb = tb - sk;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = kb + u;
            const bool valid = (unsigned)k < (unsigned)cnt;
#ifdef SD_STAMP
            ++nst_;
#endif
            SD_T(0);
            const bool c0 = k == k0L, cE = k == kEU;
            double av[3], up[3], at[3];
            const double d = nd_;
#pragma unroll
            for (int j = 0; j < 3; ++j) { av[j] = nav[j]; up[j] = nup[j]; at[j] = nat[j]; }
            ST_BARRIER();
            SD_T(1);
            // the pivots: what the step waits for.  Behind them, in flight while the divisions run, the row of the next step
            double piv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) piv[j] = st_lds(lds, Ru[j][u % kSdH]);
            SD_PREREAD(k + 1, (u + 1) & 1, (unsigned)(((u + 1) >> 1) & 1) * kSdRing, (u + 1) % kSdH);
#ifdef SD_STAMP
            asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
#endif
            SD_T(2);
            bool pj[3], pq[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) pj[j] = valid && inL[j] && !(lastL[j] && c0);
#pragma unroll
            for (int q = 0; q < 3; ++q) pq[q] = valid && inU[q] && !(q == 0 && cE);
            // u_ii = a_ii - sum (a_ik / u_kk) a_ki, eliminations in ascending k (ILU0.hpp:47-62 for rows whose eliminations
            // meet them on the diagonal only)
            double w3 = d;
            double l[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                l[j] = av[j] / piv[j];
                const double pr = l[j] * at[j];
                const double nw = w3 - pr;
                w3 = (pj[j] && hasT[j]) ? nw : w3;
            }
            {
                const unsigned long long wb = st_bits(w3);
                if (wb == kSentinel || wb == kAbsent) w3 = st_dbl(kCanonNaN);
            }
#ifdef SD_STAMP
            asm volatile("" :: "v"(w3));
#endif
            SD_T(3);
            *reinterpret_cast<double *>(lds + hoU + (unsigned)(u % kSdH) * kSdHoRow) = w3;
            // pivots that other workgroups read: write-through, to the exchange
            if (EX) { if (exports && valid) st_agent_f64(A.xch + (xoff + (tb + u) * xE), w3); }
            // (everything below is off the chain of the step)
            SD_CLEAN_AND_HAND((u + 1) % kSdH);
            const int cw = tb + u - tminw;
#ifdef SD_NOSTORE
            if (valid && (unsigned)cw < (unsigned)nchw && w3 == 1.2345e-300) {
#else
            if (valid && (unsigned)cw < (unsigned)nchw) {
#endif
                unsigned char *o = pl + (size_t)(base + cw) * 2048;
                v2d la, lb;
                la.x = pj[0] ? l[0] : absent; la.y = pj[1] ? l[1] : absent;
                lb.x = pj[2] ? l[2] : absent; lb.y = 1.0;
                SD_STORE(la, reinterpret_cast<v2d *>(o + lo16));
                SD_STORE(lb, reinterpret_cast<v2d *>(o + lo16 + 1024));
                unsigned char *ou = pu + (size_t)(base + cw) * 2048;
                v2d ua, ub;
                ua.x = pq[0] ? up[0] : absent; ua.y = pq[1] ? up[1] : absent;
                ub.x = pq[2] ? up[2] : absent; ub.y = w3;
                SD_STORE(ua, reinterpret_cast<v2d *>(ou + lo16));
                SD_STORE(ub, reinterpret_cast<v2d *>(ou + lo16 + 1024));
            }
            SD_T(4);
        }
    }
#undef SD_PREREAD
#undef SD_CLEAN_AND_HAND
    SD_T_END(0);
#ifdef SD_STAMP
    if (t == 0 && wg == (int)gridDim.x - 1) { g_sd_stamp[10] = __builtin_amdgcn_s_memtime(); g_sd_stamp[11] = __builtin_amdgcn_s_memrealtime(); }
    if (t == 0 && wg == 0) { g_sd_stamp[12] = __builtin_amdgcn_s_memtime(); }
#endif
}

// ---------------------------------------------------------------------------------------------
// the courier: lane p brings pair p -- the pivot of a step from the exchange (polled kStPF steps ahead; all-sentinel before the
// kernel) into the hand-off slot of that step, the transposed entry from A into the slot of its step one step EARLY (the lanes
// read their transposed entries during the step before)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sd_courier(const SdArgs &A, const unsigned long long *idle, unsigned char *lds, const SdPair P,
                                           const int tlo, const int thi)
{
    constexpr int NP = 8;                                 // (steps ahead: the pivots polled, the transposed entries read from A -- the latter want the distance)
    const int ln = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = sd_rsrc(A);
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(A.xch);
    const unsigned span = (unsigned)P.cnt;
    const unsigned hoU = kSdHo + (unsigned)(kThreads + ln) * 8u;
    const unsigned hoA = kSdHo + kSdHoU + (unsigned)kThreads * kSdHoAt + (unsigned)ln * 8u;
    unsigned long long gq[NP];
    double ga[NP];                                        // ga[s % NP]: the transposed entry of step s
#define SDC_ADDR(k_) ((unsigned)(k_) < span ? src + (P.idx0 + ((k_) + P.sk) * P.stride) : idle)
#define SDC_AT(k_) (((unsigned)(k_) < span && P.hasT) ? P.at0 + (unsigned)((k_) * P.atm) - ((k_) == P.klast ? (unsigned)P.sh : 0u) : kSdOob)
#define SDC_LDAT(k_) __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, SDC_AT(k_), 0, 0))
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        gq[g] = ld_agent_u64(SDC_ADDR(tlo + g - P.sk));
        ga[g] = SDC_LDAT(tlo + g - P.sk);
        asm volatile("" ::: "memory");
    }
    // the transposed entry of the first step, before anybody reads it
    *reinterpret_cast<double *>(lds + hoA) = st_clean(ga[0]);
    ga[0] = SDC_LDAT(tlo + NP - P.sk);
    ST_BARRIER();
    bool dead = false;
    for (int tb = tlo; tb < thi; tb += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = tb + u - P.sk;
            const bool need = (unsigned)k < span;
            unsigned long long v = gq[u % NP];
            if (!dead) {
                unsigned spins = 0;
                while (__builtin_amdgcn_ballot_w64(need && v == kSentinel) != 0) {
                    if (need && v == kSentinel) v = ld_agent_u64(SDC_ADDR(k));
                    __builtin_amdgcn_s_waitcnt(0x0F70);          // retired here, not at the join after the loop
                    __builtin_amdgcn_s_sleep(SD_CSLEEP);
                    if ((++spins & 255u) == 0) {
                        if (spins > kStSpinLimit) atomicExch(&A.ctrl[1], 1);
                        const int e = ld_agent_i32(&A.ctrl[1]);
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        if (spins > kStSpinLimit || e != 0) { dead = true; break; }
                    }
                }
            }
            *reinterpret_cast<unsigned long long *>(lds + hoU + (unsigned)(u % kSdH) * kSdHoRow) = v;
            *reinterpret_cast<double *>(lds + hoA + (unsigned)((u + 1) % kSdH) * kSdHoRow) = st_clean(ga[(u + 1) % NP]);
            gq[u % NP] = ld_agent_u64(SDC_ADDR(k + NP));
            ga[(u + 1) % NP] = SDC_LDAT(k + 1 + NP);
            ST_BARRIER();
        }
    }
#undef SDC_ADDR
#undef SDC_AT
#undef SDC_LDAT
    if (dead && ln == 0) atomicExch(&A.ctrl[1], 1);
}

// ---------------------------------------------------------------------------------------------
// a producer wave: 8 threads per lane, 16 bytes each; kSdPer groups of 8 lanes per wave
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sd_producer(const SdArgs &A, unsigned char *lds, const int wg, const int pw, const int tlo, const int thi)
{
    const int ln = threadIdx.x & 63, sub = ln & 7, lg = ln >> 3;
    const __amdgpu_buffer_rsrc_t rs = sd_rsrc(A);
    unsigned g[kSdPer], S[kSdPer];
    const int b0 = tlo >> 1;
#pragma unroll
    for (int i = 0; i < kSdPer; ++i) {
        const int l = (pw * kSdPer + i) * 8 + lg;
        const bool live = l < kThreads;
        const int32_t *T = A.ltab + (size_t)(wg * kThreads + (live ? l : 0)) * kStTab;
        const int cnt = T[ST_CNT], sk = T[ST_SKEW], fl = T[ST_DFL], p0 = T[ST_P0];
        const int ownL = (fl >> 2) & 1, m = fl >> 4;
        const unsigned Cu = 8u * (unsigned)(p0 - ownL - sk * m) + (unsigned)A.val_shift;
        // (the eighth piece of a block is only read when the rows start in the second half of the first: 2 rows x 56 B + 8 B.
        // Not fetching the blocks before a lane's first row and behind its last one -- 12 % of the pieces -- was measured and made the
        // kernel 5 % SLOWER: two compares and a select per load in a wave that shares its SIMD with a consumer.)
        const bool on = live && cnt > 0 && (sub < 7 || (Cu & 15u) + 16u * (unsigned)m > 112u);
        S[i] = on ? 16u * (unsigned)m : 0u;
        g[i] = on ? (Cu & ~15u) + (unsigned)b0 * S[i] + 16u * (unsigned)sub : kSdOob;
#ifdef SD_NOLOAD
        g[i] = kSdOob; S[i] = 0;
#endif
    }
    const unsigned dst = (unsigned)(pw * kSdPer * 8 + lg) * kSdPitch + 16u * (unsigned)sub;      // group i: + i * 8 * kSdPitch
    v4u ra[kSdRA][kSdPer];
#define SDP_LOAD(rb)                                                                       \
    do {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < kSdPer; ++i) {                               \
            ra[rb][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, g[i], 0, 0);             \
            g[i] += S[i];                                                                  \
        }                                                                                  \
    } while (0)
    // (two 8-byte LDS stores with IMMEDIATE offsets -- the places of a wave's eleven groups lie up to 45 KB apart, beyond the 2 KB reach
    // of ds_write2_b64, and an address add per store in a wave that shares its SIMD with a consumer costs the consumer its issue slots)
    const unsigned dst_lds = (unsigned)reinterpret_cast<uintptr_t>(lds) + dst;
#define SDP_WRITE(rb, parity)                                                              \
    do {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < kSdPer; ++i) {                               \
            if ((pw * kSdPer + i) * 8 < kThreads) {                                        \
                typedef unsigned long long u64_;                                           \
                const v4u x_ = ra[rb][i];                                                  \
                const u64_ lo_ = ((u64_)x_.y << 32) | x_.x, hi_ = ((u64_)x_.w << 32) | x_.z; \
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(dst_lds), "v"(lo_), "n"(i * (8 * kSdPitch) + (parity) * kSdRing) : "memory"); \
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(dst_lds), "v"(hi_), "n"(i * (8 * kSdPitch) + (parity) * kSdRing + 8) : "memory"); \
            }                                                                              \
        }                                                                                  \
    } while (0)
    // the first kSdRA blocks; block b0 (b0 is a multiple of 4: the steps of a trip of the loops are 8) goes to the ring at once
#pragma unroll
    for (int rb = 0; rb < kSdRA; ++rb) { SDP_LOAD(rb); asm volatile("" ::: "memory"); }
    SDP_WRITE(0, 0);
    SDP_LOAD(0);
    ST_BARRIER();                                           // (the lanes read their first row behind this one)
    SD_T_DECL(pw == 0 && wg == SD_STAMP_WG);
    for (int tb = tlo; tb < thi; tb += 8) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            SD_T(0);
            ST_BARRIER();                                       // the consumers read block tb/2 + bb now; the one before it is free
            SD_T(1);
#ifdef SD_STAMP
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kSdRA - 1) * kSdPer) : "memory");
            ++nst_;
#endif
            SD_T(2);
            SDP_WRITE((bb + 1) & 3, (bb + 1) & 1);
            SDP_LOAD((bb + 1) & 3);
            SD_T(3);
            ST_BARRIER();
            SD_T(4);
        }
    }
    SD_T_END(16);
#undef SDP_LOAD
#undef SDP_WRITE
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kSdThreads)
k_ilu0_sd(SdArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ SdPair s_pairs[64];
    __shared__ int s_cnt[4], s_total;
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = (unsigned)atomicAdd(&A.ctrl[0], 1);
    __syncthreads();
    const int wg = (int)s_ticket;
    const int t = threadIdx.x;
#ifdef SD_STAMP
    if (t == 0 && wg == 0) { g_sd_stamp[8] = __builtin_amdgcn_s_memtime(); g_sd_stamp[9] = __builtin_amdgcn_s_memrealtime(); }
#endif
    int tlo = 0x7fffffff, thi = -0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t *w4 = A.wtab + (size_t)(wg * 4 + q) * 4;
        const int a = w4[1], b = w4[2];
        if (b > 0) { tlo = min(tlo, a); thi = max(thi, a + b); }
    }
    tlo = __builtin_amdgcn_readfirstlane(tlo); thi = __builtin_amdgcn_readfirstlane(thi);
    if (thi <= tlo) return;
    tlo &= ~7;                                                        // step % 8 = position in the unrolled loops
    if (t < 64) { SdPair z; z.idx0 = 0; z.stride = 0; z.sk = 0; z.cnt = 0; z.at0 = 0; z.atm = 0; z.klast = -1; z.sh = 0; z.hasT = 0; s_pairs[t] = z; }
    if (t < kThreads) {
        const int slot = wg * kThreads + t;
        const int32_t *T = A.ltab + (size_t)slot * kStTab;
        const int nd = T[ST_ND], cnt = T[ST_CNT];
        // where each dependency's pivot is read -- the hand-off slot of the producer lane's step, dt steps back: one address per
        // residue of the step -- and where its transposed entry is (the producer row's q-th entry right of the diagonal, in the slot of
        // that row's step), or the courier's places of this step
        unsigned Ru[3][kSdH], Rt[3][kSdH];
        bool hasT[3], isg[3];
        SdPair gp[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int sw = T[ST_SRC + j];
            const int ty = (j < nd && cnt > 0) ? (sw & 3) : ST_NONE;
            const int q = (ty != ST_NONE) ? T[ST_Q + j] : -1;
            hasT[j] = q >= 0;
            const int qq = q < 0 ? 0 : q;
            const int os = sw >> 2;
            const int lane = ty == ST_LOCAL ? (os & 255) : t;
            const int dt = ty == ST_LOCAL ? T[ST_DT + j] : 1;
            isg[j] = ty == ST_GHOST;
#pragma unroll
            for (int i = 0; i < kSdH; ++i) {
                const unsigned sl = kSdHo + (unsigned)((i - dt) & (kSdH - 1)) * kSdHoRow;
                Ru[j][i] = sl + (unsigned)lane * 8u;
                Rt[j][i] = sl + kSdHoU + (unsigned)lane * kSdHoAt + 8u * (unsigned)qq;
            }
            SdPair d; d.idx0 = 0; d.stride = 0; d.sk = T[ST_SKEW]; d.cnt = cnt > 0 ? cnt : 0; d.at0 = 0; d.atm = 0; d.klast = -1; d.sh = 0; d.hasT = 0;
            if (isg[j]) {
                const int pw = os >> 8;
                const int32_t *TP = A.ltab + (size_t)os * kStTab;
                const int E = A.xw[pw * 4];
                const int kap = T[ST_KAP + j];
                d.stride = E;
                d.idx0 = A.xw[pw * 4 + 3] + (kap + TP[ST_SKEW] - T[ST_SKEW] - A.xw[pw * 4 + 1]) * E + A.xe[os];
                const int flp = TP[ST_DFL];
                const int mp = flp >> 4;
                d.hasT = hasT[j] ? 1 : 0;
                d.atm = 8 * mp;
                d.at0 = (unsigned)A.val_shift + 8u * (unsigned)(TP[ST_P0] - ((flp >> 2) & 1) + TP[ST_ND] + 1 + qq + kap * mp);
                d.klast = TP[ST_CNT] - 1 - kap;
                d.sh = 8 * ((flp >> 3) & 1);
            }
            gp[j] = d;
        }
        __syncthreads();                                              // (s_pairs zeroed)
        // the pairs of the workgroup, numbered
        {
            const int wv = t >> 6;
            unsigned long long bal[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) bal[j] = __builtin_amdgcn_ballot_w64(isg[j]);
            const int mine = __popcll(bal[0]) + __popcll(bal[1]) + __popcll(bal[2]);
            if ((t & 63) == 0) s_cnt[wv] = mine;
            __syncthreads();
            int before = 0;
            for (int q = 0; q < wv; ++q) before += s_cnt[q];
            if (t == 0) s_total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (isg[j]) {
                    const int p = before + __builtin_amdgcn_mbcnt_hi((unsigned)(bal[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[j], 0));
                    if (p < 64) s_pairs[p] = gp[j];
#pragma unroll
                    for (int i = 0; i < kSdH; ++i) {
                        Ru[j][i] = kSdHo + (unsigned)i * kSdHoRow + (unsigned)(kThreads + min(p, 63)) * 8u;
                        Rt[j][i] = kSdHo + (unsigned)i * kSdHoRow + kSdHoU + (unsigned)kThreads * kSdHoAt + (unsigned)min(p, 63) * 8u;
                    }
                }
                before += __popcll(bal[j]);
            }
        }
        __syncthreads();
        if (t == 0 && s_total > 64) atomicExch(&A.ctrl[1], 1);        // (the analysis does not let such a schedule through)
        const bool wave_exports = __any(cnt > 0 && A.xe[slot] >= 0);
        if (wave_exports) sd_consumer<true>(A, lds, wg, Ru, Rt, hasT, tlo, thi); else sd_consumer<false>(A, lds, wg, Ru, Rt, hasT, tlo, thi);
    } else if (t < kThreads + 64) {
        __syncthreads();
        __syncthreads();
        __syncthreads();
        const SdPair P = s_pairs[t - kThreads];
        // (a poll nobody needs goes to a place of this workgroup's own: the same address for the whole chip would be a hot spot)
        const unsigned long long *idle = reinterpret_cast<const unsigned long long *>(A.ltab + (size_t)wg * kThreads * kStTab);
        sd_courier(A, idle, lds, P, tlo, thi);
    } else {
        __syncthreads();
        __syncthreads();
        __syncthreads();
        sd_producer(A, lds, wg, (t - kThreads - 64) >> 6, tlo, thi);
    }
}

// ---------------------------------------------------------------------------------------------
// analysis: the lane fields the kernel needs, and the statement about the lanes it relies on (lane level)
// ---------------------------------------------------------------------------------------------
// the statement, row by row (pattern only): row k of a lane starts at p0 + k m (- 1 behind the first row when the chain entry
// exists), and its columns are the template's -- r + oL, r, r + oU -- without r - 1 in the first row and r + 1 in the last
__global__ void __launch_bounds__(256)
k_sd_proof(int32_t n, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, int64_t nnz, int32_t B, int32_t nb,
           const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot, const int32_t *__restrict__ ltabF,
           const int32_t *__restrict__ ltabB, const int32_t *__restrict__ uslot, int32_t *__restrict__ dflags)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int b = block_of(r, B, nb, start);
    const int f = blk2slot[b];
    const int32_t *T = ltabF + (size_t)f * kStTab;
    const int su = uslot[f];
    if (su < 0) { atomicOr(dflags, 1); return; }
    const int32_t *TB = ltabB + (size_t)su * kStTab;
    const v4i t0 = *reinterpret_cast<const v4i *>(T);                 // first, cnt, skew, nd
    const int k = r - t0.x, cnt = t0.y, nd = t0.w;
    const int fl = T[ST_DFL], p0 = T[ST_P0];
    const int ndU = fl & 3, ownL = (fl >> 2) & 1, ownU = (fl >> 3) & 1, m = fl >> 4;
    int bad = (k < 0 || k >= cnt) ? 1 : 0;
    const int q0 = Aptr[r], q1 = Aptr[r + 1];
    const int noL = (ownL && k == 0) ? 1 : 0, noU = (ownU && k == cnt - 1) ? 1 : 0;
    if (q0 != p0 + k * m - ((ownL && k > 0) ? 1 : 0)) bad = 1;
    if (q1 - q0 != m - noL - noU) bad = 1;
    if (!bad) {
        const Row8 rc = load_row8(Aidx, q0, q1 - q0, nnz);
        int e = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nd && !(noL && j == nd - 1)) { if (ROW8_AT(rc, e) != r + T[ST_OFF + j]) bad = 1; ++e; }
        if (ROW8_AT(rc, e) != r) bad = 1;
        ++e;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < ndU && !(noU && q == 0)) { if (ROW8_AT(rc, e) != r + TB[ST_OFF + q]) bad = 1; ++e; }
    }
    if (bad) atomicOr(dflags, 4);
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// The checks of the premise.  Row level: every block of the schedule is one whole chain and the rows of every chain are alike
// (found by the first pass over the pattern, symbolic.hip: Schedule::chains) -- then the lane templates, taken from three sampled
// rows, hold for every row.  Lane level: sd_tab_lane (st_common.h; run by k_st_scat); dflags (device, one int, zeroed by the caller)
// is non-zero afterwards when the matrix is not one for this kernel.  ILUPP_SD_VERIFY=1 also runs the row-by-row statement (k_sd_proof).
bool st_direct_prepare(hipStream_t st, const DevMat &A, const Schedule &fwd, int32_t *dflags)
{
    static const bool off = getenv("ILUPP_NO_DIRECT") != nullptr;
    // (the producers address A's values with 32-bit byte offsets)
    if (off || !fwd.chains || !A.val || (A.nnz + 4) * 8 >= 0x7fffffffLL) return false;
    (void)st; (void)dflags;                  // (the flags block was cleared when it was made: st.hip, st_structure)
    return true;
}
// (tests, ILUPP_SD_VERIFY=1: the row-by-row statement next to the light one; after k_st_scat has made the lane fields)
void st_direct_verify(hipStream_t st, const DevMat &A, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu, int32_t *dflags)
{
    static const bool verify = getenv("ILUPP_SD_VERIFY") != nullptr;
    if (!verify) return;
    hipLaunchKernelGGL(k_sd_proof, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, st, A.n, A.ptr, A.idx, (int64_t)A.nnz, fwd.B, fwd.nb,
                       fwd.start, fwd.blk2slot, pl->ltab, pu->ltab, pu->uslot, dflags);
}

int ilu0_numeric_sd(hipStream_t st, const DevMat &A, PackedSweep *pl, PackedSweep *pu, int32_t *d_ctrl, float *kernel_ms,
                    hipEvent_t e0, hipEvent_t e1)
{
    {
        static std::once_flag once[64];      // once per device
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_sd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSdLds));
        });
    }
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    fill_u64(st, reinterpret_cast<unsigned long long *>(pl->xch), pl->xch_len, kSentinel);
    SdArgs a;
    a.ltab = pl->ltab; a.wtab = pl->wtab;
    const uintptr_t vp = reinterpret_cast<uintptr_t>(A.val);
    a.val = reinterpret_cast<const double *>(vp & ~(uintptr_t)15);
    a.val_shift = (int32_t)(vp & 15);
    a.val_bytes = (uint32_t)(A.nnz * 8 + a.val_shift);
    a.pkL = reinterpret_cast<v2d *>(pl->pk); a.pkU = reinterpret_cast<v2d *>(pu->pk);
    a.xe = pl->xe; a.xw = pl->xw; a.xch = pl->xch; a.ctrl = d_ctrl;
    ILUPP_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_ilu0_sd, dim3((unsigned)pl->nwg), dim3(kSdThreads), kSdLds, st, a);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4];
    ILUPP_HIP(d2h_async(st, ctrl, d_ctrl, 16));
    ILUPP_HIP(stream_sync(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

#ifdef SD_STAMP
void sd_read_stamps(unsigned long long *out) { ILUPP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sd_stamp), sizeof(unsigned long long) * 32)); }
#endif

}  // namespace ilupp

#ifdef SD_STAMP
extern "C" int ilupp_hip_debug_sd_stamps(unsigned long long *out)
{
    try { ilupp::sd_read_stamps(out); } catch (...) { return -1; }
    return 0;
}
#endif
