// ilupp_amd/csrc/st_wave.hip -- the sweeps of the static level-major form with the exchange inside the wave (gfx950; round 4).
//
// What round 3's kernels (st.hip) spend a step on, measured on one tile alone on the chip (2048 x 16 x 16, no other workgroup, no
// traffic): 0.29 us = about 600 cycles for about 80 instructions of ONE wave per SIMD -- a step is bound by the instruction issue of
// a lone wave, not by the LDS round trip behind its barrier (a first version that only took that round trip off the chain ran at
// exactly the same pace).  So this file attacks the instruction count of the wave that walks the chain:
//
//   * class-aligned records (format 1, st_common.h: wr_classify): the three coefficients of a row sit where the lane's three
//     sources are -- the unknown of lane - 16 (class C), of lane - 1 (class B), the lane's own previous one (class A) -- in the
//     order the reference accumulates them (ascending column: C, B, A forward; A, B, C backward).  A coefficient that does not
//     exist is +0.0, and it meets an unknown that is +0.0 as well (a cell of zeros in the hand-off array for a whole class; the
//     lane's own "unknown" of a step without a row is forced to +0.0): x - 0.0 * 0.0 is x, bit for bit, for every x.  No per-entry
//     "is it there" compare, no select per entry, no permutation of sources: a step is exchange, three multiply-subtracts, one guard;
//   * the neighbours inside a wave (a wave = 16 x 4 lanes of the 16 x 16 patch) hand their unknown over in registers: DPP row_shr:1
//     for lane - 1, ds_bpermute for lane - 16.  What comes from other waves or other workgroups is at least kWrLag = 2 steps old
//     (the skews are computed with that weight, st.hip: st_link_body) and is read from the hand-off array a step early, behind the
//     barrier that follows its store: no LDS round trip on the chain of a step;
//   * streams through buffer resources: a step outside the wave's chunks is out of range (loads return zero, stores are
//     dropped): no clamps, no dump places, no 64-bit address arithmetic;
//   * the courier wave also EXPORTS: the unknowns other workgroups read leave through it (one 8-byte-per-lane write-through store
//     per step for the whole workgroup instead of one per wave), so the waves on the chain issue no such store.
//
// Arithmetic and its order are st.hip's (sparse_implementation.h:4040-4087: sequential accumulation in stored order, division by
// the diagonal found by position); results are bit-identical (tests A/B the two: ILUPP_NO_WR=1).
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "st_common.h"

namespace ilupp {

static constexpr int kWxRow = kThreads + 64 + 8;          // doubles per slot of the hand-off array: lanes, courier pairs, the cell of zeros (+ padding)
static constexpr int kWxZero = kThreads + 64;             // index of the cell of zeros in a slot
static constexpr int kWxLds = 2 * kStH * kWxRow * 8;

#ifdef WX_STAMP
// diagnostics build only: [0..3] forward, [4..7] backward sweep: shader cycles, 100 MHz ticks, steps of wave 0 of workgroup 0
__device__ unsigned long long g_wx_stamp[16];
#endif

// ---------------------------------------------------------------------------------------------
// format 0 <-> format 1, in place.  One thread per (wave, chunk, lane) of the forward schedule.  TO1: by template position with
// kAbsent -> class-aligned with +0.0, and the zero record {0, 0} {0, 1} wherever a lane has no row at a step of its wave; else back
// (an entry exists where the lane's template has it: every row but the first of a chain has the own-chain entry -- wx_lane_ok).
// The conversion back serves what still reads positions (factors(), the transposed records); the factor kernel writes format 1.
// ---------------------------------------------------------------------------------------------
template <bool TO1>
__global__ void __launch_bounds__(512)
k_wx_convert(const int32_t *__restrict__ ltabF, const int32_t *__restrict__ ltabB, const int32_t *__restrict__ uslot,
             const int32_t *__restrict__ wtab, v2d *__restrict__ pkL, v2d *__restrict__ pkU)
{
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    if (c >= nch) return;
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const int32_t *T = ltabF + (size_t)slot * kStTab;
    const int k = tmin + c - T[ST_SKEW], cnt = T[ST_CNT];
    v2d *pl = pkL + ((size_t)base + c) * 128 + L;
    v2d *pu = pkU + ((size_t)base + c) * 128 + L;
    if (k < 0 || k >= cnt) {
        if (TO1) { v2d z0, z1; z0.x = 0.0; z0.y = 0.0; z1.x = 0.0; z1.y = 1.0; pl[0] = z0; pl[64] = z1; pu[0] = z0; pu[64] = z1; }
        return;
    }
    const int su = uslot[slot];
    const int32_t *TB = ltabB + (size_t)(su < 0 ? 0 : su) * kStTab;
    const double absent = st_dbl(kAbsent);
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        v2d *p = side == 0 ? pl : pu;
        const int32_t *TT = side == 0 ? T : TB;
        const int tl = side == 0 ? (slot & 255) : (su & 255);
        const int kk = side == 0 ? k : cnt - 1 - k;                  // the row's index in its schedule's processing order
        int cls[3]; bool ring[3];
        (void)wr_classify(TT, tl, side == 1, cls, ring);
        const v2d a = p[0], b = p[64];
        const double v[3] = {a.x, a.y, b.x};
        double o[3];
        if (TO1) {
            o[0] = o[1] = o[2] = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (cls[j] != WR_NONE && st_bits(v[j]) != kAbsent) {
                    const int s = wr_slot_of(cls[j], side == 1);
                    if (s == 0) o[0] = v[j]; else if (s == 1) o[1] = v[j]; else o[2] = v[j];
                }
        } else {
            o[0] = o[1] = o[2] = absent;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (cls[j] != WR_NONE && !((TT[ST_SRC + j] & 3) == ST_OWN && kk == 0)) {
                    const int s = wr_slot_of(cls[j], side == 1);
                    o[j] = s == 0 ? v[0] : (s == 1 ? v[1] : v[2]);
                }
        }
        v2d x; x.x = o[0]; x.y = o[1]; p[0] = x;
        x.x = o[2]; x.y = b.y; p[64] = x;
    }
}

void wx_convert_records(hipStream_t st, PackedSweep *pl, PackedSweep *pu, int to_fmt)
{
    if (pl->fmt == to_fmt && pu->fmt == to_fmt) return;
    const dim3 grid((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));
    if (to_fmt == 1)
        hipLaunchKernelGGL((k_wx_convert<true>), grid, dim3(512), 0, st, pl->ltab, pu->ltab, pu->uslot, pl->wtab, reinterpret_cast<v2d *>(pl->pk),
                           reinterpret_cast<v2d *>(pu->pk));
    else
        hipLaunchKernelGGL((k_wx_convert<false>), grid, dim3(512), 0, st, pl->ltab, pu->ltab, pu->uslot, pl->wtab, reinterpret_cast<v2d *>(pl->pk),
                           reinterpret_cast<v2d *>(pu->pk));
    ILUPP_HIP(hipGetLastError());
    pl->fmt = pu->fmt = to_fmt;
}

// ---------------------------------------------------------------------------------------------
// the exchange inside a wave
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wx_dpp_shr1(const double old, const double src)
{
    const long long o = __double_as_longlong(old), v = __double_as_longlong(src);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)v, 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(v >> 32), 0x111, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wx_from_lane(const int byte_addr, const double src)
{
    const long long v = __double_as_longlong(src);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(v >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// what a lane knows about its sources
struct WxLane {
    unsigned aB, aC;              // hand-off array: where the stand-in of class B / C is read (+ (step % 8) slots); the cell of zeros without one
    bool ringC;                   // class C is a hand-off value or nothing (else: lane - 16)
    int src16;                    // byte address of lane - 16 for ds_bpermute
};
__device__ __forceinline__ bool wx_lane_setup(const int32_t *T, const int t, const bool bwd, const unsigned va[3], WxLane *W)
{
    int cls[3]; bool ring[3];
    const bool ok = wx_lane_ok(T, t, bwd);
    (void)wr_classify(T, t, bwd, cls, ring);
    const unsigned zero = (unsigned)((kStH * kWxRow + kWxZero) * 8);
    bool hasB = false, hasC = false;
    W->aB = zero; W->aC = zero; W->ringC = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (cls[j] == WR_B) { hasB = true; if (ring[j]) W->aB = va[j]; }
        if (cls[j] == WR_C) { hasC = true; if (ring[j]) W->aC = va[j]; else W->ringC = false; }
    }
    (void)hasB; (void)hasC;
    W->src16 = ((t - 16) & 63) * 4;
    return ok;
}

// ---------------------------------------------------------------------------------------------
// the 256 lanes of the schedule.  DR = +1 forward, -1 backward; DIV: divide by the record's diagonal
// ---------------------------------------------------------------------------------------------
template <int DR, bool DIV>
__device__ __forceinline__ void wx_sweep_wave(const StSArgs &A, unsigned char *xh, const int wg, const WxLane W, const int tlo, const int thi)
{
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], sk = T[ST_SKEW];
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    const int ysrc_ = DR > 0 ? 0 : A.ysrc[slot];
    // streams.  Forward: the wave's own chunks (records, right-hand side and, in place, the result).  Backward: records and right-hand
    // side lie in the FORWARD schedule's order (chunk and lane of the lane's row 0 from ysrc, one chunk down per row); the result goes
    // to the wave's own chunks of ylm.
    const unsigned char *pkb = reinterpret_cast<const unsigned char *>(A.pk);
    unsigned char *xlb = reinterpret_cast<unsigned char *>(A.xlm);
    const __amdgpu_buffer_rsrc_t rrec = DR > 0
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(pkb + (size_t)base * 2048), 0, nchw * 2048, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(pkb), 0, (int)((unsigned)A.xlm_chunks * 2048u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrhs = DR > 0
        ? __builtin_amdgcn_make_buffer_rsrc(xlb + (size_t)base * 512, 0, nchw * 512, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(xlb, 0, (int)((unsigned)A.xlm_chunks * 512u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = DR > 0 ? rrhs
        : __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char *>(A.ylm) + (size_t)base * 512, 0, nchw * 512, 0x00020000);
    // byte offsets of step tlo (wrapping arithmetic: a step outside the stream is a huge offset = out of range)
    unsigned vrec = DR > 0 ? (unsigned)(tlo - tminw) * 2048u + (unsigned)ln * 16u
                           : (unsigned)((ysrc_ >> 6) + sk - tlo) * 2048u + (unsigned)(ysrc_ & 63) * 16u;
    unsigned vrhs = DR > 0 ? (unsigned)(tlo - tminw) * 512u + (unsigned)ln * 8u
                           : (unsigned)((ysrc_ >> 6) + sk - tlo) * 512u + (unsigned)(ysrc_ & 63) * 8u;
    unsigned vout = (unsigned)(tlo - tminw) * 512u + (unsigned)ln * 8u;
    constexpr unsigned dRec = DR > 0 ? 2048u : 0u - 2048u, dRhs = DR > 0 ? 512u : 0u - 512u;

    typedef double v2dd __attribute__((ext_vector_type(2)));
    typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
    v4u ra[kStRA][2];
    double rr[kStRA];
#ifdef WX_X_NOMEM
#define WXS_LOAD(u) do { } while (0)
#else
#define WXS_LOAD(u)                                                                                    \
    do {                                                                                               \
        ra[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rrec, vrec, 0, 0);                            \
        ra[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rrec, vrec + 1024u, 0, 0);                    \
        rr[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rrhs, vrhs, 0, 0));    \
        vrec += dRec; vrhs += dRhs;                                                                    \
    } while (0)
#endif
#ifdef WX_X_NOMEM
    for (int u = 0; u < kStRA; ++u) { ra[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rrec, vrec, 0, 0); ra[u][1] = ra[u][0]; rr[u] = 1.0; }
#endif
#pragma unroll
    for (int u = 0; u < kStRA; ++u) { WXS_LOAD(u); asm volatile("" ::: "memory"); }
    // (the courier's values of the first two steps are in place)
    ST_BARRIER();
    double xprev = 0.0;
    double bB = st_lds(xh, W.aB), bC = st_lds(xh, W.aC);               // the hand-off values of the first step
    int k = tlo - sk;
#ifdef WX_STAMP
    const unsigned long long st0_ = __builtin_amdgcn_s_memtime(), sr0_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (int tb = tlo; tb < thi; tb += kStRA) {
#pragma unroll
        for (int u = 0; u < kStRA; ++u) {
            // the hand-off values of the NEXT step: stored before the barrier this wave has just passed
            const double nB = st_lds(xh, W.aB + (unsigned)((u + 1) % kStH) * (kWxRow * 8));
            const double nC = st_lds(xh, W.aC + (unsigned)((u + 1) % kStH) * (kWxRow * 8));
            const bool valid = (unsigned)k < (unsigned)cnt;
            const double alt = valid ? st_dbl(kCanonNaN) : 0.0;
            const v2dd c01 = __builtin_bit_cast(v2dd, ra[u][0]), c2d = __builtin_bit_cast(v2dd, ra[u][1]);
            // the exchange inside the wave
            const double sB = wx_dpp_shr1(bB, xprev);
#ifdef WX_X_NOBPERM
            const double pC = xprev + 1.0;
#else
            const double pC = wx_from_lane(W.src16, xprev);
#endif
            const double sC = W.ringC ? bC : pC;
            const double xs0 = DR > 0 ? sC : xprev, xs2 = DR > 0 ? xprev : sC;
            double acc = rr[u];
            acc = acc - c01.x * xs0;
            acc = acc - c01.y * sB;
            acc = acc - c2d.x * xs2;
            double x = DIV ? acc / c2d.y : acc;
            // (the forward sweep does not divide: the register of the diagonal stays taken until here all the same -- a dead quarter of a
            // 16-byte load is a free register to the allocator, and what it puts there has to wait for that load, a load of a later step)
            if (!DIV) asm volatile("" :: "v"(c2d.y));
            x = (valid && x == x) ? x : alt;
#ifndef WX_X_NOLDSW
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)(u % kStH) * (kWxRow * 8)) = x;
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)(u % kStH + kStH) * (kWxRow * 8)) = x;
#endif
            xprev = x;
#ifndef WX_X_NOMEM
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, x), rout, DR > 0 ? vrhs - (unsigned)kStRA * 512u : vout, 0, 2);
#endif
            vout += 512u;
            WXS_LOAD(u);
            bB = nB; bC = nC;
            ++k;
#ifdef WX_X_NOBAR
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            ST_BARRIER();
#endif
        }
    }
#undef WXS_LOAD
#ifdef WX_STAMP
    if (t == 0 && wg == 0) {
        const int o = DR > 0 ? 0 : 4;
        g_wx_stamp[o] = __builtin_amdgcn_s_memtime() - st0_; g_wx_stamp[o + 1] = __builtin_amdgcn_s_memrealtime() - sr0_; g_wx_stamp[o + 2] = (unsigned long long)(thi - tlo);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// the courier.  Inbound as st.hip's, but two barriers early: lane p brings pair p's value of step s into the hand-off array before
// the barrier that ends step s - 2 (the lanes read it at the start of step s - 1).  Outbound: behind the barrier that ends step s it
// reads the unknowns of the exported lanes from the hand-off array and stores them to the exchange, write-through.
// ---------------------------------------------------------------------------------------------
template <int RA, int NP>
__device__ __forceinline__ void wx_courier(const unsigned long long *src, const unsigned long long *idle, unsigned char *xh, const StPair P,
                                           const int tlo, const int thi, int32_t *err, double *xch, const int xrow0, const int E, const int nexp,
                                           const int *s_exp)
{
    constexpr int SH = 2;
    const int ln = threadIdx.x & 63;
#if defined(WX_X_NOCOURIER) || defined(WX_X_NOBAR)
#ifndef WX_X_NOBAR
    ST_BARRIER();
    for (int tb = tlo; tb < thi; ++tb) ST_BARRIER();
#endif
    return;
#endif
    const unsigned span = (unsigned)(P.khi - P.klo);
    unsigned long long gq[NP];
    // exports: lane e stores the unknown of the exported lane with ordinal e -- ONE store instruction per step, issued by every lane
    // (a lane without an export stores out of range): a store under a condition is one the compiler's wait counts cannot be sure of,
    // and every wait for a polled value would then also wait for the write-through stores behind it
    const int elane = (ln < E) ? s_exp[ln] : -1;
    const unsigned ea = (unsigned)((kStH * kWxRow + (elane >= 0 ? elane : kWxZero)) * 8);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xch + xrow0, 0, (int)((unsigned)(thi - tlo) * (unsigned)E * 8u), 0x00020000);
    unsigned vx = elane >= 0 ? (unsigned)ln * 8u : 0xfffffff0u;
    const unsigned dvx = elane >= 0 ? (unsigned)E * 8u : 0u;
    (void)nexp;
#define WXC_ADDR(k_) ((unsigned)((k_) - P.klo) < span ? src + (P.idx0 + ((k_) + P.sk) * P.stride) : idle)
#pragma unroll
    for (int g = 0; g < NP; ++g) { gq[g] = ld_agent_u64(WXC_ADDR(tlo + g - P.sk)); asm volatile("" ::: "memory"); }
    bool dead = false;
#define WXC_DELIVER(i_)                                                                                              \
    do {                                                                                                             \
        const int k = tlo_ + (i_) - P.sk;                                                                            \
        const bool need = (unsigned)(k - P.klo) < span;                                                              \
        unsigned long long v = gq[(i_) % NP];                                                                        \
        if (!dead) {                                                                                                 \
            unsigned spins = 0;                                                                                      \
            while (__builtin_amdgcn_ballot_w64(need && v == kSentinel) != 0) {                                       \
                if (need && v == kSentinel) v = ld_agent_u64(WXC_ADDR(k));                                           \
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                                  \
                __builtin_amdgcn_s_sleep(ST_CSLEEP);                                                                 \
                if ((++spins & 255u) == 0) {                                                                         \
                    if (spins > kStSpinLimit) atomicExch(err, 1);                                                    \
                    const int e = ld_agent_i32(err);                                                                 \
                    __builtin_amdgcn_s_waitcnt(0x0F70);                                                              \
                    if (spins > kStSpinLimit || e != 0) { dead = true; break; }                                      \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        *reinterpret_cast<unsigned long long *>(xh + (unsigned)(kThreads + ln) * 8 + (unsigned)((i_) % kStH + kStH) * (kWxRow * 8)) = v; \
        gq[(i_) % NP] = ld_agent_u64(WXC_ADDR(k + NP));                                                              \
    } while (0)
    {
        const int tlo_ = tlo;
#pragma unroll
        for (int i = 0; i < SH; ++i) {
            typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
            WXC_DELIVER(i);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, 0.0), rx, 0xfffffff0u, 0, 16);
        }
    }
    ST_BARRIER();
    for (int tb = tlo; tb < thi; tb += RA) {
        const int tlo_ = tb;
#pragma unroll
        for (int u = 0; u < RA; ++u) {
            WXC_DELIVER(u + SH);
            ST_BARRIER();
            {
                typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
                const double v = st_lds(xh, ea + (unsigned)(u % kStH) * (kWxRow * 8));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, v), rx, vx, 0, 16);          // sc1: write-through, as st_agent_f64
                vx += dvx;
            }
        }
    }
#undef WXC_DELIVER
#undef WXC_ADDR
    if (dead && ln == 0) atomicExch(err, 1);
}

template <int DR, bool DIV>
__global__ void __launch_bounds__(kStWgThreads)
k_sptrsv_wx(StSArgs A)
{
    __shared__ __attribute__((aligned(16))) unsigned char xh[kWxLds];
    __shared__ StPair s_pairs[64];
    __shared__ int s_exp[kThreads];
    __shared__ int s_cnt[4], s_total;
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = (unsigned)atomicAdd(A.ticket, 1);
    __syncthreads();
    const int wg = (int)s_ticket;
    const int t = threadIdx.x;
    int tlo = 0x7fffffff, thi = -0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t *w4 = A.wtab + (size_t)(wg * 4 + q) * 4;
        const int a = w4[1], b = w4[2];
        if (b > 0) { tlo = min(tlo, a); thi = max(thi, a + b); }
    }
    tlo = __builtin_amdgcn_readfirstlane(tlo); thi = __builtin_amdgcn_readfirstlane(thi);
    if (thi <= tlo) return;
    tlo &= ~(kStRA - 1);
    if (t < 64) { StPair z; z.idx0 = 0; z.stride = 0; z.sk = 0; z.klo = 0; z.khi = 0; s_pairs[t] = z; }
    if (t < kThreads) {
        const int slot = wg * kThreads + t;
        const int32_t *T = A.ltab + (size_t)slot * kStTab;
        bool isg[3]; int idx0[3], stride[3]; unsigned va[3];
        st_lane_sources(T, t, A.ltab, A.xe, A.xw, isg, idx0, stride, va, kWxRow);
        s_exp[t] = -1;
        __syncthreads();                                              // (s_pairs zeroed, s_exp cleared)
        st_number_pairs(T, t, isg, idx0, stride, va, s_pairs, s_cnt, &s_total, kWxRow);
        WxLane W;
        const bool ok = wx_lane_setup(T, t, DR < 0, va, &W);
        {
            const int xe = A.xe[slot];
            if (T[ST_CNT] > 0 && xe >= 0 && xe < kThreads) s_exp[xe] = t;
        }
        // the hand-off array starts all +0.0: the cells of zeros stay that way, the rest is read before it is written only by lanes
        // whose coefficient for it is +0.0
        for (int i = t; i < 2 * kStH * kWxRow; i += kThreads) reinterpret_cast<double *>(xh)[i] = 0.0;
        __syncthreads();
        if ((t == 0 && s_total > 64) || !ok) atomicExch(A.err, 1);    // (the analysis does not let such a schedule through)
        wx_sweep_wave<DR, DIV>(A, xh, wg, W, tlo, thi);
    } else {
        __syncthreads();
        __syncthreads();                                              // (the one inside st_number_pairs)
        __syncthreads();
        const StPair P = s_pairs[t - kThreads];
        const unsigned long long *idle = reinterpret_cast<const unsigned long long *>(A.ltab + (size_t)wg * kThreads * kStTab);
        const int E = __builtin_amdgcn_readfirstlane(A.xw[wg * 4]);
        // (the exchange rows of this workgroup start at its first step rounded down to kStXAlign = 16 = what tlo is rounded to)
        const int xrow0 = A.xw[wg * 4 + 3] + (tlo - A.xw[wg * 4 + 1]) * E;
        const int nexp = 1;
        if (E > 64 && (threadIdx.x & 63) == 0) atomicExch(A.err, 1);                  // (the analysis does not let such a schedule through)
        wx_courier<kStRA, kStPS>(reinterpret_cast<const unsigned long long *>(A.xch), idle, xh, P, tlo, thi, A.err, A.xch, xrow0, E, nexp, s_exp);
    }
}

// One sweep of an apply on format-1 records (sptrsv_st's interface)
int sptrsv_wx(hipStream_t st, const PackedSweep &ps, int32_t n, const double *rhs, double *out, int32_t *d_ticket, int32_t *d_err,
              double *ypk_out, const double *ypk_in, const int32_t *ysrc)
{
    const bool fwd = ps.kind == (int)SWEEP_FWD_LAST_ASC;
    double *lml = fwd ? ypk_out : const_cast<double *>(ypk_in);        // level-major, forward order
    if (!lml || (!fwd && (!ysrc || !ps.xlm))) { set_error("static sweep without its level-major vector"); return ILUPP_ERR_INVALID; }
    StSArgs a;
    a.pk = reinterpret_cast<const v2d *>(ps.pk); a.ltab = ps.ltab; a.wtab = ps.wtab; a.n = n;
    a.nchY = (int32_t)ps.nchunks;
    a.xlm = lml; a.ylm = fwd ? nullptr : ps.xlm;
    a.ysrc = ysrc; a.xlm_chunks = (int32_t)(fwd ? ps.nchunks : ps.y_chunks);
    a.xe = ps.xe; a.xw = ps.xw; a.xch = ps.xch; a.ticket = d_ticket; a.err = d_err;
    fill_u64(st, reinterpret_cast<unsigned long long *>(ps.xch), ps.xch_len, kSentinel);
    {
        static std::once_flag once[64];      // once per device
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wx<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wx<-1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
        });
    }
    const dim3 grid((unsigned)ps.nwg);
    if (fwd) {
        st_vec_to_lm(st, ps, rhs, lml);
        hipLaunchKernelGGL((k_sptrsv_wx<1, false>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
    } else {
        hipLaunchKernelGGL((k_sptrsv_wx<-1, true>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
        st_vec_from_lm(st, ps, out);
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

#ifdef WX_STAMP
void wx_read_stamps(unsigned long long *out) { ILUPP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wx_stamp), sizeof(unsigned long long) * 16)); }
#endif

}  // namespace ilupp

#ifdef WX_STAMP
extern "C" int ilupp_hip_debug_wx_stamps(unsigned long long *out)
{
    try { ilupp::wx_read_stamps(out); } catch (...) { return -1; }
    return 0;
}
#endif
