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
// ... and per workgroup of the factor kernel (by ticket): entry, lane 0's first row, lane 0's last row, exit (100 MHz)
__device__ unsigned long long g_wf_tl[4096 * 4];
// ... and per workgroup and wave (16 slots): cycles spent waiting at the barriers of the main loop, [12..15]: loop cycles of wave 0, the
// courier's spin cycles, the producers' cycles between barriers (wave 5), steps
__device__ unsigned long long g_wf_wait[4096 * 16];
#define WF_BARRIER(acc_) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long b0_ = __builtin_amdgcn_s_memtime(); asm volatile("s_barrier" ::: "memory"); (acc_) += __builtin_amdgcn_s_memtime() - b0_; } while (0)
#else
#define WF_BARRIER(acc_) ST_BARRIER()
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
             const int32_t *__restrict__ wtab, v2d *__restrict__ pkL, v2d *__restrict__ pkU, const int descB)
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
                    // (descB: a backward sweep that accumulates in DESCENDING column order -- IChol0's T4 -- takes its coefficients in the
                    // forward sweeps' slot order C, B, A)
                    const int s = wr_slot_of(cls[j], side == 1 && !descB);
                    if (s == 0) o[0] = v[j]; else if (s == 1) o[1] = v[j]; else o[2] = v[j];
                }
        } else {
            o[0] = o[1] = o[2] = absent;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (cls[j] != WR_NONE && !((TT[ST_SRC + j] & 3) == ST_OWN && kk == 0)) {
                    const int s = wr_slot_of(cls[j], side == 1 && !descB);
                    // (a value that looks like one of the two markers of format 0 -- a NaN with that payload among A's values, or what
                    // the arithmetic made of one -- must not read as "no entry")
                    o[j] = st_clean(s == 0 ? v[0] : (s == 1 ? v[1] : v[2]));
                }
        }
        v2d x; x.x = o[0]; x.y = o[1]; p[0] = x;
        x.x = o[2]; x.y = TO1 ? b.y : st_clean(b.y); p[64] = x;
    }
}

// format 2 -> format 1 for the L records: a wave per chunk, {lA} at 1024 + 8 lane becomes {lA, 1} at 1024 + 16 lane (every lane has read
// before any lane writes: the loads of a wave's instruction are issued before its next instruction's stores, which need their data)
__global__ void __launch_bounds__(512)
k_wx_expand_l(const int32_t *__restrict__ wtab, unsigned char *__restrict__ pkL)
{
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], nch = wtab[(size_t)w * 4 + 2];
    if (c >= nch) return;
    unsigned char *ch = pkL + ((size_t)base + c) * 2048 + 1024;
    const double lA = *reinterpret_cast<const double *>(ch + 8 * L);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    v2d x; x.x = lA; x.y = 1.0;
    *reinterpret_cast<v2d *>(ch + 16 * L) = x;
}

void wx_convert_records(hipStream_t st, PackedSweep *pl, PackedSweep *pu, int to_fmt)
{
    if (to_fmt == 2) to_fmt = 1;                     // (nobody but the factor kernel makes compact records: an object that left them stays on format 1)
    if (pl->fmt == 2) {
        const dim3 grid((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));
        hipLaunchKernelGGL(k_wx_expand_l, grid, dim3(512), 0, st, pl->wtab, reinterpret_cast<unsigned char *>(pl->pk));
        ILUPP_HIP(hipGetLastError());
        pl->fmt = 1;
    }
    if (pl->fmt == to_fmt && pu->fmt == to_fmt) return;
    const dim3 grid((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));
    if (to_fmt == 1)
        hipLaunchKernelGGL((k_wx_convert<true>), grid, dim3(512), 0, st, pl->ltab, pu->ltab, pu->uslot, pl->wtab, reinterpret_cast<v2d *>(pl->pk),
                           reinterpret_cast<v2d *>(pu->pk), (pu->pair && pu->desc) ? 1 : 0);
    else
        hipLaunchKernelGGL((k_wx_convert<false>), grid, dim3(512), 0, st, pl->ltab, pu->ltab, pu->uslot, pl->wtab, reinterpret_cast<v2d *>(pl->pk),
                           reinterpret_cast<v2d *>(pu->pk), (pu->pair && pu->desc) ? 1 : 0);
    ILUPP_HIP(hipGetLastError());
    pl->fmt = pu->fmt = to_fmt;
}

// the records of the transposed apply (st.hip: k_st_transpose writes them by template position): class-aligned, the backward ones --
// L^T, accumulated in descending column order -- in the forward sweeps' slot order
void wx_convert_transposed(hipStream_t st, PackedSweep *pl, PackedSweep *pu)
{
    if (!pl->pkT || !pu->pkT || (pl->fmtT == 1 && pu->fmtT == 1)) return;
    const dim3 grid((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));
    hipLaunchKernelGGL((k_wx_convert<true>), grid, dim3(512), 0, st, pl->ltab, pu->ltab, pu->uslot, pl->wtab, reinterpret_cast<v2d *>(pl->pkT),
                       reinterpret_cast<v2d *>(pu->pkT), 1);
    ILUPP_HIP(hipGetLastError());
    pl->fmtT = pu->fmtT = 1;
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
// VEC: the caller's vector (natural order) travels through a ring in LDS that a sixth wave of the workgroup fills (forward: the
// right-hand side) or drains (backward: the result) in whole 128-byte lines -- wx_vector below: no level-major copy of the vector, no
// conversion kernels, one vector-memory instruction less per step in the waves on the chain.
static constexpr int kVecPitch = 34 * 8;                  // bytes of a lane's 32 ring entries (+ 2: the lanes of a wave spread over the banks)
static constexpr int kVecRing = kThreads * kVecPitch;

// CL: the forward sweep of a unit lower factor on COMPACT records (format 2: the second piece of a record is the one coefficient lA, 8 bytes
// per lane at 1024 + 8 lane of the chunk; the constant 1 of format 1's {lA, 1} is neither stored nor fetched: 12 memory lines per chunk
// instead of 16)
template <int DR, bool DIV, bool VEC, bool DESC = false, bool CL = false>
__device__ __forceinline__ void wx_sweep_wave(const StSArgs &A, unsigned char *xh, const int wg, const WxLane W, const int tlo, const int thi,
                                              unsigned char *vr)
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
    static_assert(!CL || (DR > 0 && !DIV), "compact records: the forward sweep of a unit lower factor");
    const unsigned cl2 = 1024u - (unsigned)ln * 8u;                       // (CL) from a lane's first piece to its second

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
        if (CL) { const v2u_ c_ = __builtin_amdgcn_raw_buffer_load_b64(rrec, vrec + cl2, 0, 0); ra[u][1].x = c_.x; ra[u][1].y = c_.y; } \
        else ra[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rrec, vrec + 1024u, 0, 0);               \
        if (!(VEC && DR > 0)) rr[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rrhs, vrhs, 0, 0)); \
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
#ifdef WX_PRIO
    __builtin_amdgcn_s_setprio(WX_PRIO);
#endif
    double xprev = 0.0;
    double bB = st_lds(xh, W.aB), bC = st_lds(xh, W.aC);               // the hand-off values of the first step
    int k = tlo - sk;
    const unsigned vbase = (unsigned)t * kVecPitch;                    // (VEC) this lane's ring
    double bR = (VEC && DR > 0) ? st_lds(vr, vbase + (((unsigned)k & 31u) << 3)) : 0.0;
#ifdef WX_STAMP
    const unsigned long long st0_ = __builtin_amdgcn_s_memtime(), sr0_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (int tb = tlo; tb < thi; tb += kStRA) {
#pragma unroll
        for (int u = 0; u < kStRA; ++u) {
            // the hand-off values of the NEXT step: stored before the barrier this wave has just passed
            const double nB = st_lds(xh, W.aB + (unsigned)((u + 1) % kStH) * (kWxRow * 8));
            const double nC = st_lds(xh, W.aC + (unsigned)((u + 1) % kStH) * (kWxRow * 8));
            // (VEC, forward) the right-hand side of the next step: in the ring since at least kVecLead steps
            const double nR = (VEC && DR > 0) ? st_lds(vr, vbase + (((unsigned)(k + 1) & 31u) << 3)) : 0.0;
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
            // (accumulation in ascending column order: C, B, A forward, A, B, C backward; DESC: a backward sweep in descending order)
            const double xs0 = (DR > 0 || DESC) ? sC : xprev, xs2 = (DR > 0 || DESC) ? xprev : sC;
            double acc = (VEC && DR > 0) ? bR : rr[u];
            acc = acc - c01.x * xs0;
            acc = acc - c01.y * sB;
            acc = acc - c2d.x * xs2;
            double x = DIV ? acc / c2d.y : acc;
            // (the forward sweep does not divide: the register of the diagonal stays taken until here all the same -- a dead quarter of a
            // 16-byte load is a free register to the allocator, and what it puts there has to wait for that load, a load of a later step)
            if (!DIV && !CL) asm volatile("" :: "v"(c2d.y));
            x = (valid && x == x) ? x : alt;
#ifndef WX_X_NOLDSW
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)(u % kStH) * (kWxRow * 8)) = x;
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)(u % kStH + kStH) * (kWxRow * 8)) = x;
#endif
            xprev = x;
#ifndef WX_X_NOMEM
            if (VEC && DR < 0) *reinterpret_cast<double *>(vr + vbase + (((unsigned)k & 31u) << 3)) = x;      // (the vector wave takes it from there)
            else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, x), rout, vout, 0, 2);
#endif
            vout += 512u;
            WXS_LOAD(u);
            bB = nB; bC = nC; bR = nR;
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

// ---------------------------------------------------------------------------------------------
// the vector wave (VEC).  A lane's rows 16 g .. 16 g + 15 (group g) are ONE 128-byte line of the caller's vector: eight threads move it,
// 16 bytes each, as whole lines.  The ring holds two groups per lane (entry k mod 32).  Lane t reaches group g at step 16 g + skew(t), so
// the lanes whose skew is congruent to the step modulo 16 are served at that step ("phase"; 16 lanes of a 16 x 16 patch per phase = two
// instructions of 8 lanes):
//   forward:  load group g + 1 (its half of the ring was read for the last time two steps ago), store it into the ring kVecLead steps
//             later -- 8 steps before the lane reads its first entry;
//   backward: group g - 1 has just been completed: read it from the ring, store it (descending rows: the pairs swapped).
// Per step: 2 loads + 2 LDS stores (forward), 2 LDS loads + 2 stores (backward; + 2 for chains of odd length).
// ---------------------------------------------------------------------------------------------
static constexpr int kVecSlots = 2;                       // instructions of 8 lanes per phase
struct VecLds { int first[kThreads], cnt[kThreads], sk[kThreads]; int plist[kThreads]; int pstart[17]; int pcnt[16]; int odd; };

static constexpr int kVecRecs = 16 * kVecSlots * 64;                // one 16-byte record per (phase, instruction, thread) ...
static constexpr int kVecDyn = kVecRing + 64 + kVecRecs * 16;       // ... behind the ring (and its dump place) in the dynamic LDS

template <int DR>
__device__ __forceinline__ void wx_vector(const StSArgs &A, unsigned char *vr, VecLds *V, const int tlo, const int thi)
{
    typedef double v2dd __attribute__((ext_vector_type(2)));
    typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
    const int ln = threadIdx.x & 63, j = ln & 7, grp = ln >> 3;
    // the phases' lane lists (this wave alone: LDS operations of one wave are in order)
    if (ln < 16) V->pcnt[ln] = 0;
    if (ln == 0) V->odd = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int myp[4], mypos[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tt = ln + 64 * i;
        myp[i] = V->cnt[tt] > 0 ? (V->sk[tt] & 15) : -1;
        mypos[i] = myp[i] >= 0 ? atomicAdd(&V->pcnt[myp[i]], 1) : 0;
        if (V->cnt[tt] > 0 && (V->cnt[tt] & 1)) V->odd = 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ln == 0) { int run = 0; for (int p = 0; p < 16; ++p) { V->pstart[p] = run; run += V->pcnt[p]; } V->pstart[16] = run; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bool over = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (myp[i] >= 0) {
            if (mypos[i] >= 8 * kVecSlots) over = true; else V->plist[V->pstart[myp[i]] + mypos[i]] = ln + 64 * i;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (__builtin_amdgcn_ballot_w64(over) != 0 && ln == 0) atomicExch(A.err, 1);      // (the analysis does not let such a schedule through: flags[9] & 8)
    const bool odd = __builtin_amdgcn_readfirstlane(V->odd) != 0;
    const int thiR = tlo + ((thi - tlo + kStRA - 1) & ~(kStRA - 1));      // (the other waves run whole trips of kStRA steps: as many barriers here)
    const __amdgpu_buffer_rsrc_t rnat = __builtin_amdgcn_make_buffer_rsrc(A.nat, 0, (int)((unsigned)A.n * 8u), 0x00020000);
    // What this thread does for instruction q of a step of phase p, as ONE record {c1, c2, cnt, c3} (a step then costs this wave one
    // 16-byte LDS load per instruction, not a chain of table look-ups).  With lane tt of the phase, its skew sk and this thread's pair j:
    //   forward (phase of step s: (s + 8) mod 16; the lane starts group m = (s + 8 - sk) / 16 eight steps later and group m + 1 is asked
    //     for): first row of the pair k0 = s + c2, c2 = 24 + 2 j - sk; vector offset c1 + 8 s, c1 = 8 (first + c2); ring place
    //     c3 + 8 (k0 mod 32), c3 = tt * pitch;
    //   backward (phase s mod 16; group (s - sk) / 16 - 1 is complete): lower row of the pair kk = s + c2, c2 = -sk - 2 - 2 j; vector
    //     offset c1 - 8 s, c1 = 8 (first - 1 - c2); ring place as above.
    int4 *rec = reinterpret_cast<int4 *>(vr + kVecRing + 64);
    for (int pq = 0; pq < 16 * kVecSlots; ++pq) {
        const int p = pq / kVecSlots, q = pq % kVecSlots;
        int4 r = make_int4(0, 0, 0, 0);
        if (8 * q + grp < V->pcnt[p]) {
            const int tt = V->plist[V->pstart[p] + 8 * q + grp];
            const int sk = V->sk[tt], c2 = DR > 0 ? 24 + 2 * j - sk : -sk - 2 - 2 * j;
            r.x = DR > 0 ? 8 * (V->first[tt] + c2) : 8 * (V->first[tt] - 1 - c2);
            r.y = c2; r.z = V->cnt[tt]; r.w = tt * kVecPitch;
        }
        rec[pq * 64 + ln] = r;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (DR > 0) {
        // groups 0 and 1 of every lane before the first step (what the steps before tlo would have asked for)
#pragma unroll 1
        for (int b0 = 0; b0 < kThreads / 8; b0 += 4) {
            v4u hold[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int tt = (b0 + (i & 3)) * 8 + grp;
                const int k0 = 2 * j + 16 * (i >> 2);
                unsigned off = (V->cnt[tt] > 0 && k0 < V->cnt[tt]) ? (unsigned)(V->first[tt] + k0) * 8u : 0xfffffff0u;
                asm volatile("" : "+v"(off));
                hold[i] = __builtin_amdgcn_raw_buffer_load_b128(rnat, off, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int tt = (b0 + (i & 3)) * 8 + grp;
                *reinterpret_cast<v4u *>(vr + (unsigned)tt * kVecPitch + (unsigned)(2 * j + 16 * (i >> 2)) * 8u) = hold[i];
            }
        }
        ST_BARRIER();
        v4u ring[kStRA][kVecSlots];
        unsigned dst[kStRA][kVecSlots];
#pragma unroll
        for (int u = 0; u < kStRA; ++u)
#pragma unroll
            for (int q = 0; q < kVecSlots; ++q) { ring[u][q] = v4u{0, 0, 0, 0}; dst[u][q] = (unsigned)kVecRing; }     // (the dump place behind the ring)
        static_assert(kStRA == 16, "a request goes into the ring one trip of the unrolled loop (16 steps) after it was made");
        for (int tb = tlo; tb < thiR; tb += kStRA) {
#pragma unroll
            for (int u = 0; u < kStRA; ++u) {
                const int s = tb + u;
                int4 r[kVecSlots];
#pragma unroll
                for (int q = 0; q < kVecSlots; ++q) r[q] = rec[(((u + 8) & 15) * kVecSlots + q) * 64 + ln];
#pragma unroll
                for (int q = 0; q < kVecSlots; ++q) {
                    // what was asked for 16 steps ago goes into the ring (its half was read for the last time 9 steps ago, its first entry
                    // is read in 7), the next request takes its place
                    *reinterpret_cast<v4u *>(vr + dst[u][q]) = ring[u][q];
                    const int k0 = s + r[q].y;
                    const bool ok = (unsigned)k0 < (unsigned)r[q].z;
                    // (ONE load instruction whatever `ok` says: as two loads in two branches the compiler can no longer count what is in
                    // flight and waits for everything, every step -- the offsets are pinned as values before the instruction)
                    unsigned lo_ = ok ? (unsigned)(r[q].x + 8 * s) : 0xfffffff0u;
                    unsigned ds_ = ok ? (unsigned)r[q].w + (((unsigned)k0 & 31u) << 3) : (unsigned)kVecRing;
                    asm volatile("" : "+v"(lo_), "+v"(ds_));
                    ring[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rnat, lo_, 0, 0);
                    dst[u][q] = ds_;
                }
                ST_BARRIER();
            }
        }
    } else {
        ST_BARRIER();
        // at step s the groups completed by step s - 1 leave; after the last step the rest (no more barriers: everything is in the ring)
        const int tend = thiR + 16;
        for (int tb = tlo; tb < tend; tb += kStRA) {
#pragma unroll
            for (int u = 0; u < kStRA; ++u) {
                const int s = tb + u;
#pragma unroll
                for (int q = 0; q < kVecSlots; ++q) {
                    const int4 r = rec[((u & 15) * kVecSlots + q) * 64 + ln];
                    const int kk = s + r.y;                               // the pair's rows: kk (higher address), kk + 1 (lower address)
                    const bool hi_ok = (unsigned)kk < (unsigned)r.z, lo_ok = hi_ok && kk + 1 < r.z;
                    const v2dd pr = *reinterpret_cast<const v2dd *>(vr + (unsigned)r.w + (((unsigned)kk & 31u) << 3));
                    v2dd sw; sw.x = pr.y; sw.y = pr.x;                    // {x(kk + 1), x(kk)}: ascending addresses
                    const unsigned fo = (unsigned)(r.x - 8 * s);
                    unsigned o2_ = lo_ok ? fo : 0xfffffff0u, o1_ = (hi_ok && !lo_ok) ? fo + 8u : 0xfffffff0u;
                    asm volatile("" : "+v"(o2_), "+v"(o1_));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, sw), rnat, o2_, 0, 2);
                    if (odd) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, pr.x), rnat, o1_, 0, 2);
                }
                if (s < thiR) ST_BARRIER();
            }
        }
    }
}

template <int DR, bool DIV, bool VEC, bool DESC = false, bool CL = false>
__device__ __forceinline__ void wx_sweep_body(const StSArgs &A, unsigned char *vr)
{
    __shared__ __attribute__((aligned(16))) unsigned char xh[kWxLds];
    __shared__ VecLds s_vec;
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
        if (VEC) { s_vec.first[t] = T[ST_FIRST]; s_vec.cnt[t] = T[ST_CNT]; s_vec.sk[t] = T[ST_SKEW]; }
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
        wx_sweep_wave<DR, DIV, VEC, DESC, CL>(A, xh, wg, W, tlo, thi, vr);
    } else if (VEC && t >= kThreads + 64) {
        __syncthreads();                                              // (the lanes' fields are in s_vec)
        __syncthreads();
        __syncthreads();
        wx_vector<DR>(A, vr, &s_vec, tlo, thi);
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

template <int DR, bool DIV, bool DESC = false>
__global__ void __launch_bounds__(kStWgThreads)
k_sptrsv_wx(StSArgs A)
{
    wx_sweep_body<DR, DIV, false, DESC>(A, nullptr);
}
// ... with the vector wave: the caller's vector where it lies (A.nat)
template <int DR, bool DIV, bool DESC = false, bool CL = false>
__global__ void __launch_bounds__(kStWgThreads + 64)
k_sptrsv_wv(StSArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wv_ring[];
    wx_sweep_body<DR, DIV, true, DESC, CL>(A, wv_ring);
}

bool wx_vec_on()
{
    static const bool on = getenv("ILUPP_NO_VECWAVE") == nullptr;
    return on;
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
    // the caller's vector where it lies, through the vector wave (the forward sweep reads `rhs`, the backward sweep writes `out`);
    // ILUPP_NO_VECWAVE=1, or a schedule with more than 16 lanes in a phase: through the level-major copies that k_st_vec makes (round 4)
    const bool vec = wx_vec_on() && ps.vec_ok;
    if (ps.fmt == 2 && !(vec && fwd && !ps.pair)) { set_error("compact records without their sweep"); return ILUPP_ERR_INTERNAL; }
    a.nat = fwd ? const_cast<double *>(rhs) : out;
    if (ps.xch_armed) ps.xch_armed = false;
    else fill_u64(st, reinterpret_cast<unsigned long long *>(ps.xch), ps.xch_len, kSentinel);
    {
        static std::once_flag once[64];      // once per device
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wx<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wx<-1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wx<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wv<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kVecDyn));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wx<-1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wv<-1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kVecDyn));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wv<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kVecDyn));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wv<1, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kVecDyn));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_wv<-1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kVecDyn));
        });
    }
    const dim3 grid((unsigned)ps.nwg);
    if (vec) {
        if (fwd && ps.pair) hipLaunchKernelGGL((k_sptrsv_wv<1, true>), grid, dim3(kStWgThreads + 64), kVecDyn, st, a);
        else if (fwd && ps.fmt == 2) hipLaunchKernelGGL((k_sptrsv_wv<1, false, false, true>), grid, dim3(kStWgThreads + 64), kVecDyn, st, a);
        else if (fwd) hipLaunchKernelGGL((k_sptrsv_wv<1, false>), grid, dim3(kStWgThreads + 64), kVecDyn, st, a);
        else if (ps.pair && ps.desc) hipLaunchKernelGGL((k_sptrsv_wv<-1, true, true>), grid, dim3(kStWgThreads + 64), kVecDyn, st, a);
        else hipLaunchKernelGGL((k_sptrsv_wv<-1, true>), grid, dim3(kStWgThreads + 64), kVecDyn, st, a);
    } else if (fwd) {
        st_vec_to_lm(st, ps, rhs, lml);
        // (a pair of stored factors -- an LL^T object -- has a diagonal of its own in the forward factor)
        if (ps.pair) hipLaunchKernelGGL((k_sptrsv_wx<1, true>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
        else hipLaunchKernelGGL((k_sptrsv_wx<1, false>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
    } else {
        if (ps.pair && ps.desc) hipLaunchKernelGGL((k_sptrsv_wx<-1, true, true>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
        else hipLaunchKernelGGL((k_sptrsv_wx<-1, true>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
        st_vec_from_lm(st, ps, out);
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

// =============================================================================================
// ILU(0), wave-exchange form: the direct-feed factor kernel (st_direct.hip: producer waves stream A's values into LDS, consumer
// waves run the pivot recurrence u_rr = a_rr - sum_k (a_rk / u_kk) a_kr, ILU0.hpp:47-62 for rows whose eliminations meet them on
// the diagonal only) rebuilt around a lean consumer:
//   * the PRODUCERS put every entry of a row where its class says (a canonical record of eight doubles per lane and step --
//     {aC, aB} {aA, d} {a'A, a'B} {a'C, -}: the entries left of the diagonal by forward class, the diagonal, the entries right of it
//     by backward class; a place no entry goes to stays +0.0): each producer thread's two 8-byte stores of a block simply go to
//     thread-constant places.  The consumer reads its row with four 16-byte LDS loads at a lane-constant address: nothing about a
//     row is looked up or selected, only the own-chain entries of a chain's first and last row are masked;
//   * pivots travel as the sweeps' unknowns do (DPP / ds_bpermute inside the wave, the hand-off array two steps old otherwise, a
//     cell of ONES for a class without an entry: 0 / 1 = +0.0); the transposed entries a(k, r) -- entries right of the diagonal of
//     the pivot row -- are re-published by the lane that owns the pivot row when it pre-reads that row, one step before its pivot;
//   * records leave in format 1 as they are computed: {lC, lB} {lA, 1} and {a'A, a'B} {a'C, u_rr}; no "absent" selects, stores
//     through buffer resources (a step outside the wave's chunks is dropped);
//   * the courier imports pivots (polled) and transposed entries (from A) of earlier workgroups two barriers early and exports.
// Arithmetic and its order are st_direct.hip's (bit-identical results; ILUPP_NO_WR=1 runs the old kernels).
// =============================================================================================
static constexpr int kWfH = 4;                            // steps of hand-off history (kept twice: slot s and s + 4)
#ifdef WF_HALF
// experiment: a workgroup runs the first 128 lanes of its slots only (schedules made with ILUPP_TILE_TZ=8: patches of 16 x 8 lines), with
// half the LDS, so that two workgroups share a CU
static constexpr int kWfLanes = 128, kWfPairs = 32;
#else
static constexpr int kWfLanes = kThreads, kWfPairs = 64;
#endif
#ifndef WF_PSLEEP
#define WF_PSLEEP 4
#endif
static constexpr int kWfProdNap = WF_PSLEEP;              // s_sleep units (64 cycles) a producer waits behind each barrier before it issues loads
static constexpr int kWfRow = kWfLanes + kWfPairs + 16;   // doubles per slot: lanes, courier pairs, [320] a cell of ones / zeros (+ padding: 4 rows = 21 x 512 B)
static constexpr int kWfCell = kWfLanes + kWfPairs;
static constexpr int kWfArr = 2 * kWfH * kWfRow * 8;      // bytes of one hand-off array
static constexpr int kWfPitch = 80;                       // bytes of a lane's record in the row ring: 8 doubles + 16 (a wave's 16-byte loads and the producers' 8-byte stores then spread over the banks)
static constexpr int kWfSlot = kWfLanes * kWfPitch;       // bytes of a step of the row ring
static constexpr int kWfRing = 4 * kWfSlot;               // two blocks of two steps
static constexpr unsigned kWfX = kWfRing;                 // pivots
static constexpr unsigned kWfTB = kWfRing + kWfArr;       // a'B of every row (and the courier's transposed entries)
static constexpr unsigned kWfTC = kWfRing + 2 * kWfArr;   // a'C
static constexpr int kWfLds = kWfRing + 3 * kWfArr + 64;
#ifdef WF_HALF
static constexpr int kWfProd = 3, kWfPer = 6, kWfRA = 4;
#else
#ifndef WF_RA
#define WF_RA 4
#endif
static constexpr int kWfProd = 6, kWfPer = 6, kWfRA = WF_RA;  // producer waves, groups of 8 lanes per wave, blocks read ahead (1, 2 or 4)
static_assert(WF_RA == 1 || WF_RA == 2 || WF_RA == 4, "the producers' ring of registers is walked with (bb + 2) % kWfRA inside trips of four blocks");
#endif
static constexpr int kWfThreads = kWfLanes + 64 + 64 * kWfProd;
static_assert(kWfProd * kWfPer * 8 >= kWfLanes, "every lane needs a producer");
static_assert((kWfH * kWfRow * 8) % 512 == 0, "the two copies of a hand-off value are stored with one ds_write2st64_b64");

struct WfArgs {
    const int32_t *ltab, *ltabB, *uslot, *wtab;   // forward lane table, backward lane table, forward -> backward slot, chunk table
    const double *val;                            // A's values, the pointer rounded down to 16 bytes
    uint32_t val_bytes;
    int32_t val_shift;
    unsigned char *pkL, *pkU;                     // format-1 records, both in the forward schedule's order
    const int32_t *xe, *xw;
    double *xch;
    int64_t xch_len;                              // doubles of the exchange: a workgroup's export window never reaches past it
    int32_t *ctrl;                                // [0] ticket, [1] error; k_ilu0_wa: [2], [3], [9] .. [14] tickets by XCD
    int32_t flags;                                // k_ilu0_wa: 2 = every workgroup of the launch is resident at once; 1 = ... and tiles are handed out by XCD
    int32_t *prog;                                // k_ilu0_wa: [tile] steps done, [nwg + tile] blocks of eight steps somebody has asked for, [2 nwg + tile] blocks somebody finishes (or null)
};
struct WfPair { int idx0, stride, sk, cnt; unsigned at0; int atm, klast, sh, hasT; int astart, pw; };    // (st_direct.hip: SdPair; k_ilu0_wa: astart, where the producer's workgroup exports its first step; pw, that workgroup)

// what a consumer lane knows
struct WfLane {
    unsigned xB, xC;              // pivot hand-off: stand-in of class B / C (the cell of ones without an entry)
    unsigned tB, tC;              // transposed entries of the class B / C elimination (the cell of zeros without one)
    bool ringC;
    int src16;
    bool hasB, hasC, hasUB, hasUC;   // the lane's rows have an entry of class B / C left of the diagonal; of (backward) class B / C right of it
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wf_rsrc(const WfArgs &A)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.val), 0, (int)A.val_bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------------
// the 256 lanes of the schedule
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wf_consumer(const WfArgs &A, unsigned char *lds, const int wg, const WfLane W, const int tlo, const int thi)
{
    typedef double v2dd __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], sk = T[ST_SKEW];
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(A.pkL + (size_t)base * 2048, 0, nchw * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(A.pkU + (size_t)base * 2048, 0, nchw * 2048, 0x00020000);
    unsigned vout = (unsigned)(tlo - tminw) * 2048u + (unsigned)ln * 16u;          // (a step outside the wave's chunks: out of range)
    const unsigned rowa = (unsigned)t * kWfPitch;                                  // this lane's record in a step of the row ring
    const unsigned xown = kWfX + (unsigned)t * 8u, tbown = kWfTB + (unsigned)t * 8u, tcown = kWfTC + (unsigned)t * 8u;

    // the row of a step is read during the step BEFORE (and its entries right of the diagonal handed on at once)
#define WF_ROW(s4_, r0_, r1_, r2_, r3_)                                                                           \
    do {                                                                                                           \
        r0_ = *reinterpret_cast<const v2dd *>(lds + rowa + (unsigned)(s4_) * kWfSlot);                             \
        r1_ = *reinterpret_cast<const v2dd *>(lds + rowa + (unsigned)(s4_) * kWfSlot + 16u);                       \
        r2_ = *reinterpret_cast<const v2dd *>(lds + rowa + (unsigned)(s4_) * kWfSlot + 32u);                       \
        r3_ = *reinterpret_cast<const v2dd *>(lds + rowa + (unsigned)(s4_) * kWfSlot + 48u);                       \
    } while (0)
#define WF_HAND_T(h4_, r2_, r3_)                                                                                   \
    do {                                                                                                           \
        *reinterpret_cast<double *>(lds + tbown + (unsigned)(h4_) * (kWfRow * 8)) = (r2_).y;                       \
        *reinterpret_cast<double *>(lds + tbown + (unsigned)((h4_) + kWfH) * (kWfRow * 8)) = (r2_).y;              \
        *reinterpret_cast<double *>(lds + tcown + (unsigned)(h4_) * (kWfRow * 8)) = (r3_).x;                       \
        *reinterpret_cast<double *>(lds + tcown + (unsigned)((h4_) + kWfH) * (kWfRow * 8)) = (r3_).x;              \
    } while (0)

    // The producers place a row's entries by their distance from the row's (virtual) start.  A chain's first row has no own-chain
    // entry left of its diagonal and its last row none right of it: there the neighbours' entries sit one place nearer to the diagonal
    // (and the place at the far end holds an entry of another row).  Put right once per row, when the row is handed on; only the two
    // rows at the ends of a chain need it, so the selects sit behind a branch the whole wave takes or skips.
    const int fl = T[ST_DFL];
    const int kF = ((fl >> 2) & 1) ? 0 : -1, kL = ((fl >> 3) & 1) ? cnt - 1 : -1;
    // first row: the entry of template position j lies at the place of position j + 1; last row: of position q at the place of q - 1
    const bool fCB = W.hasC && W.hasB, fCA = W.hasC && !W.hasB, fBA = W.hasB;
    const bool lCB = W.hasUC && W.hasUB, lCA = W.hasUC && !W.hasUB, lBA = W.hasUB;
#define WF_ENDS(kk_, r0_, r1_, r2_, r3_)                                                                          \
    do {                                                                                                           \
        const bool f_ = (kk_) == kF, l_ = (kk_) == kL;                                                             \
        if (__builtin_amdgcn_ballot_w64(f_ || l_) != 0) {                                                          \
            const double c_ = (r0_).x, b_ = (r0_).y, a_ = (r1_).x, ua_ = (r2_).x, ub_ = (r2_).y, uc_ = (r3_).x;    \
            (r0_).x = f_ ? (fCB ? b_ : (fCA ? a_ : c_)) : c_;                                                      \
            (r0_).y = f_ ? (fBA ? a_ : b_) : b_;                                                                   \
            (r1_).x = f_ ? 0.0 : a_;                                                                               \
            (r2_).x = l_ ? 0.0 : ua_;                                                                              \
            (r2_).y = l_ ? (lBA ? ua_ : ub_) : ub_;                                                                \
            (r3_).x = l_ ? (lCB ? ub_ : (lCA ? ua_ : uc_)) : uc_;                                                  \
        }                                                                                                          \
    } while (0)
    ST_BARRIER();                                           // (the producers' first two blocks and the courier's first entries are in place)
#ifdef WF_PRIO
    __builtin_amdgcn_s_setprio(WF_PRIO);                    // (experiment: the waves on the chain before the couriers and producers that share their SIMDs)
#endif
    v2dd c0_, c1_, c2_, c3_;                                // the row of the current step: {aC, aB} {aA, d} {a'A, a'B} {a'C, -}
    v2dd n0_, n1_, n2_, n3_;                                // ... of the next step
    WF_ROW(0, c0_, c1_, c2_, c3_);
    WF_ROW(1, n0_, n1_, n2_, n3_);
    WF_ENDS(tlo - sk, c0_, c1_, c2_, c3_);
    WF_HAND_T(0, c2_, c3_);
    double bB = st_lds(lds, W.xB), bC = st_lds(lds, W.xC);  // pivots of other waves / workgroups for the first step
    ST_BARRIER();                                           // (everybody's transposed entries of the first step are handed on)
    double tB = st_lds(lds, W.tB), tC = st_lds(lds, W.tC);
    double w3prev = 1.0, upA = 0.0;                         // the pivot of the lane's previous row; that row's own-chain entry right of the diagonal
    double qC = 1.0;                                        // the pivot of lane - 16 (asked for at the end of the step before)
    int k = tlo - sk;
    unsigned long long wacc_ = 0;
#ifdef WX_STAMP
    const unsigned long long wt0_ = __builtin_amdgcn_s_memtime();
#endif
    (void)wacc_;
    for (int tb = tlo; tb < thi; tb += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // the pivots: inside the wave from registers
            const double pC = W.ringC ? bC : qC;
            const double pB = wx_dpp_shr1(bB, w3prev);
            // for the steps to come (nothing of it is used before the next barrier): the row of the step after the next, the pivots and
            // the transposed entries of the next step that do not come through the wave's registers
            v2dd m0_, m1_, m2_, m3_;
            WF_ROW((u + 2) & 3, m0_, m1_, m2_, m3_);
            const double nB = st_lds(lds, W.xB + (unsigned)((u + 1) & 3) * (kWfRow * 8));
            const double nC = st_lds(lds, W.xC + (unsigned)((u + 1) & 3) * (kWfRow * 8));
            const double ntB = st_lds(lds, W.tB + (unsigned)((u + 1) & 3) * (kWfRow * 8));
            const double ntC = st_lds(lds, W.tC + (unsigned)((u + 1) & 3) * (kWfRow * 8));
            const bool valid = (unsigned)k < (unsigned)cnt;
#ifdef WX_STAMP
            if (t == 0 && wg < 4096 && k == 0) g_wf_tl[wg * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            if (t == 0 && wg < 4096 && k == cnt - 1) g_wf_tl[wg * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
            // (the lane's previous step had no row: what it left in upA is not an entry of the matrix)
            const double aA = c1_.x, tA = k == 0 ? 0.0 : upA;
            const double uA = c2_.x;
            // u_rr = a_rr - sum (a_rk / u_kk) a_kr, eliminations in ascending k: classes C, B, A
#ifdef WF_X_NODIV
            const double lC = c0_.x * pC, lB = c0_.y * pB, lA = aA * w3prev;
#else
            const double lC = c0_.x / pC, lB = c0_.y / pB, lA = aA / w3prev;
#endif
            double w = c1_.y;
            w = w - lC * tC;
            w = w - lB * tB;
            w = w - lA * tA;
            {
                // (a pivot must not look like the marker of the exchange; a lane without a row hands on 1)
                const unsigned long long wb = st_bits(w);
                if ((wb & ~3ull) == (kSentinel & ~3ull)) w = st_dbl(kCanonNaN);
            }
            const double w3 = valid ? w : 1.0;
            *reinterpret_cast<double *>(lds + xown + (unsigned)(u & 3) * (kWfRow * 8)) = w3;
            *reinterpret_cast<double *>(lds + xown + (unsigned)((u & 3) + kWfH) * (kWfRow * 8)) = w3;
            qC = wx_from_lane(W.src16, w3);
            w3prev = w3; upA = c2_.x;
            {
                typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
                v2dd la, lb, ua, ub;
                la.x = lC; la.y = lB; lb.x = lA; lb.y = 1.0;
                ua.x = uA; ua.y = c2_.y; ub.x = c3_.x; ub.y = w3;
#ifdef WF_X_NOSTORE
                if (w3 == 1.2345e-300) {
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, la), rL, vout, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, lb), rL, vout + 1024u, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, ua), rU, vout, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, ub), rU, vout + 1024u, 0, 2);
#ifdef WF_X_NOSTORE
                }
#endif
                vout += 2048u;
            }
            // the next row (read a step ago): its ends put right, its entries right of the diagonal handed on -- a step before its pivot
            WF_ENDS(k + 1, n0_, n1_, n2_, n3_);
            WF_HAND_T((u + 1) & 3, n2_, n3_);
            c0_ = n0_; c1_ = n1_; c2_ = n2_; c3_ = n3_;
            n0_ = m0_; n1_ = m1_; n2_ = m2_; n3_ = m3_;
            bB = nB; bC = nC; tB = ntB; tC = ntC;
            ++k;
            WF_BARRIER(wacc_);
        }
    }
#ifdef WX_STAMP
    if (ln == 0 && wg < 4096) { g_wf_wait[wg * 16 + wv] = wacc_; if (wv == 0) { g_wf_wait[wg * 16 + 12] = __builtin_amdgcn_s_memtime() - wt0_; g_wf_wait[wg * 16 + 15] = (unsigned long long)(thi - tlo); } }
#endif
#undef WF_ROW
#undef WF_HAND_T
#undef WF_ENDS
}

// ---------------------------------------------------------------------------------------------
// the courier: lane p serves pair p.  Inbound, before the barrier that ends step s - 2: the pivot of step s from the exchange (polled
// kStPF steps ahead) and the transposed entry of step s from A.  Outbound, behind the barrier that ends step s: the pivots of the
// exported lanes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wf_courier(const WfArgs &A, const unsigned long long *idle, unsigned char *lds, const WfPair P,
                                           const int tlo, const int thi, const int wg, const int *s_exp)
{
    // (the pivots are polled NP steps ahead -- a tile settles that many steps further behind the one it reads from --, the transposed
    // entries, which are A's and wait for nobody, NA steps ahead)
    constexpr int NP = kStPF, NA = 8, SH = 2;
    typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
    const int ln = threadIdx.x & 63;
#ifdef WF_X_NOCOURIER
    ST_BARRIER(); ST_BARRIER();
    for (int tb = tlo; tb < thi; ++tb) ST_BARRIER();
    return;
#endif
    const __amdgpu_buffer_rsrc_t rs = wf_rsrc(A);
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(A.xch);
    const unsigned span = (unsigned)P.cnt;
    // (a courier lane beyond the pair slots delivers to a place of the padding nobody reads)
    const unsigned hoX = kWfX + (unsigned)((kWfH * kWfRow + (ln < kWfPairs ? kWfLanes + ln : kWfCell + 8)) * 8);
    const unsigned hoT = kWfTB + (unsigned)((kWfH * kWfRow + (ln < kWfPairs ? kWfLanes + ln : kWfCell + 8)) * 8);
    // exports (wx_courier)
    const int E = __builtin_amdgcn_readfirstlane(A.xw[wg * 4]);
    const int xrow0 = A.xw[wg * 4 + 3] + (tlo - A.xw[wg * 4 + 1]) * E;
    const int elane = (ln < E) ? s_exp[ln] : -1;
    const unsigned ea = kWfX + (unsigned)((kWfH * kWfRow + (elane >= 0 ? elane : kWfCell)) * 8);
    // (the window is clamped to the allocation: sizes predicted from a box grid's dimensions are only compared with what the device
    // found after this kernel has run)
    const long long xroom = (long long)A.xch_len - (long long)xrow0;
    const long long xwant = (long long)(thi - tlo) * (long long)E;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(A.xch + (xroom > 0 ? xrow0 : 0), 0, (int)(8 * (xroom > 0 ? (xwant < xroom ? xwant : xroom) : 0)), 0x00020000);
    unsigned vx = elane >= 0 ? (unsigned)ln * 8u : 0xfffffff0u;
    const unsigned dvx = elane >= 0 ? (unsigned)E * 8u : 0u;
    if (E > 64 && ln == 0) atomicExch(&A.ctrl[1], 1);                 // (the analysis does not let such a schedule through)
    unsigned long long gq[NP];
    double ga[NA];
#define WFC_ADDR(k_) ((unsigned)(k_) < span ? src + (P.idx0 + ((k_) + P.sk) * P.stride) : idle)
#define WFC_AT(k_) (((unsigned)(k_) < span && P.hasT) ? P.at0 + (unsigned)((k_) * P.atm) - ((k_) == P.klast ? (unsigned)P.sh : 0u) : 0xfffffff0u)
#define WFC_LDAT(k_) __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, WFC_AT(k_), 0, 0))
#pragma unroll
    for (int g = 0; g < NA; ++g) {
        if (g < NP) gq[g] = ld_agent_u64(WFC_ADDR(tlo + g - P.sk));
        ga[g] = WFC_LDAT(tlo + g - P.sk);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, 0.0), rx, 0xfffffff0u, 0, 16);      // (the way in looks like a pass of the loop)
        asm volatile("" ::: "memory");
    }
    bool dead = false;
#define WFC_DELIVER(i_)                                                                                              \
    do {                                                                                                             \
        const int k = tlo_ + (i_) - P.sk;                                                                            \
        const bool need = (unsigned)k < span;                                                                        \
        unsigned long long v = gq[(i_) % NP];                                                                        \
        if (!dead) {                                                                                                 \
            unsigned spins = 0;                                                                                      \
            while (__builtin_amdgcn_ballot_w64(need && v == kSentinel) != 0) {                                       \
                if (need && v == kSentinel) v = ld_agent_u64(WFC_ADDR(k));                                           \
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                                  \
                __builtin_amdgcn_s_sleep(1);                                                                         \
                if ((++spins & 255u) == 0) {                                                                         \
                    if (spins > kStSpinLimit) atomicExch(&A.ctrl[1], 1);                                             \
                    const int e = ld_agent_i32(&A.ctrl[1]);                                                          \
                    __builtin_amdgcn_s_waitcnt(0x0F70);                                                              \
                    if (spins > kStSpinLimit || e != 0) { dead = true; break; }                                      \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        *reinterpret_cast<unsigned long long *>(lds + hoX + (unsigned)((i_) & 3) * (kWfRow * 8)) = v;                \
        *reinterpret_cast<double *>(lds + hoT + (unsigned)((i_) & 3) * (kWfRow * 8)) = ga[(i_) % NA];                \
        gq[(i_) % NP] = ld_agent_u64(WFC_ADDR(k + NP));                                                              \
        ga[(i_) % NA] = WFC_LDAT(k + NA);                                                                            \
    } while (0)
    {
        const int tlo_ = tlo;
#pragma unroll
        for (int i = 0; i < SH; ++i) WFC_DELIVER(i);
    }
    ST_BARRIER();
    ST_BARRIER();
    unsigned long long wacc_ = 0, sacc_ = 0;
    (void)wacc_; (void)sacc_;
    for (int tb = tlo; tb < thi; tb += 8) {
        const int tlo_ = tb;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#ifdef WX_STAMP
            const unsigned long long d0_ = __builtin_amdgcn_s_memtime();
#endif
            WFC_DELIVER(u + SH);
#ifdef WX_STAMP
            sacc_ += __builtin_amdgcn_s_memtime() - d0_;
#endif
            WF_BARRIER(wacc_);
            {
                const double v = st_lds(lds, ea + (unsigned)(u & 3) * (kWfRow * 8));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, v), rx, vx, 0, 16);          // sc1: write-through
                vx += dvx;
            }
        }
    }
#undef WFC_DELIVER
#undef WFC_ADDR
#undef WFC_AT
#undef WFC_LDAT
#ifdef WX_STAMP
    if (ln == 0 && wg < 4096) { g_wf_wait[wg * 16 + 4] = wacc_; g_wf_wait[wg * 16 + 13] = sacc_; }
#endif
    if (dead && ln == 0) atomicExch(&A.ctrl[1], 1);
}

// ---------------------------------------------------------------------------------------------
// a producer wave: 8 threads per lane, 16 bytes of A.val each per block of two steps (st_direct.hip: sd_producer); every thread's two
// entries of a block go to the places of the canonical records they belong to
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wf_producer(const WfArgs &A, unsigned char *lds, const int wg, const int pw, const int tlo, const int thi,
                                            const int4 *s_lane, const unsigned char *s_place)
{
    const int ln = threadIdx.x & 63, sub = ln & 7, lg = ln >> 3;
#ifdef WF_X_NOPROD
    ST_BARRIER(); ST_BARRIER();
    for (int tb = tlo; tb < thi; ++tb) ST_BARRIER();
    return;
#endif
    const __amdgpu_buffer_rsrc_t rs = wf_rsrc(A);
    unsigned g[kWfPer], S[kWfPer], d0[kWfPer], d1[kWfPer];
    const int b0 = tlo >> 1;
#pragma unroll
    for (int i = 0; i < kWfPer; ++i) {
        const int l = (pw * kWfPer + i) * 8 + lg;
        const bool live = l < kWfLanes;
        // (the lane's fields and the canonical places of its row's positions: put into LDS by the lane itself, k_ilu0_wx)
        const int4 lp = s_lane[live ? l : 0];                      // cnt, skew, first entry, flags (ST_DFL)
        const int cnt = lp.x, sk = lp.y, p0 = lp.z, fl = lp.w;
        const int ownL = (fl >> 2) & 1, m = fl >> 4;
        const unsigned Cu = 8u * (unsigned)(p0 - ownL - sk * m) + (unsigned)A.val_shift;
        const bool on = live && cnt > 0 && (sub < 7 || (Cu & 15u) + 16u * (unsigned)m > 112u);
        S[i] = on ? 16u * (unsigned)m : 0u;
        g[i] = on ? (Cu & ~15u) + (unsigned)b0 * S[i] + 16u * (unsigned)sub : 0xfffffff0u;
        // where the thread's two entries of a block belong
        unsigned dd[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = 2 * sub + e - (int)((Cu & 15u) >> 3);          // entry of the block's two (virtual) rows
            int place = -1, ui = 0;
            if (on && idx >= 0 && idx < 2 * m) {
                ui = idx >= m ? 1 : 0;
                const int pos = idx - ui * m;
                place = pos < 8 ? (int)s_place[(live ? l : 0) * 8 + pos] - 1 : -1;
            }
            // (a piece of no row goes to place 7 of the lane's record, which nobody reads)
            if (place < 0) { place = 7; ui = 0; }
            dd[e] = (unsigned)ui * kWfSlot + (unsigned)(live ? l : 0) * kWfPitch + (unsigned)place * 8u;
        }
        d0[i] = dd[0]; d1[i] = dd[1];
    }
    const unsigned lbase = (unsigned)reinterpret_cast<uintptr_t>(lds);
    v4u ra[kWfRA][kWfPer];
#define WFP_LOAD(rb)                                                                       \
    do {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < kWfPer; ++i) {                               \
            ra[rb][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, g[i], 0, 0);             \
            g[i] += S[i];                                                                  \
        }                                                                                  \
    } while (0)
#define WFP_WRITE(rb, parity)                                                              \
    do {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < kWfPer; ++i) {                               \
            if ((pw * kWfPer + i) * 8 < kWfLanes) {                                        \
                typedef unsigned long long u64_;                                           \
                const v4u x_ = ra[rb][i];                                                  \
                const u64_ lo_ = ((u64_)x_.y << 32) | x_.x, hi_ = ((u64_)x_.w << 32) | x_.z; \
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(lbase + d0[i]), "v"(lo_), "n"((parity) * 2 * kWfSlot) : "memory"); \
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(lbase + d1[i]), "v"(hi_), "n"((parity) * 2 * kWfSlot) : "memory"); \
            }                                                                              \
        }                                                                                  \
    } while (0)
    // the first kWfRA blocks; blocks b0 and b0 + 1 go to the ring at once (the lanes read the row of step s during step s - 2)
#pragma unroll
    for (int rb = 0; rb < kWfRA; ++rb) { WFP_LOAD(rb); asm volatile("" ::: "memory"); }
    WFP_WRITE(0, 0);
    WFP_LOAD(0);
    WFP_WRITE(1, 1);
    WFP_LOAD(1);
    ST_BARRIER();                                           // (the lanes read their first two rows behind this one)
    ST_BARRIER();
    unsigned long long wacc_ = 0, pacc_ = 0;
    (void)wacc_; (void)pacc_;
    for (int tb = tlo; tb < thi; tb += 8) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            // steps 2 bb and 2 bb + 1 of this trip: the lanes read the rows of steps 2 bb + 2 and 2 bb + 3 (block bb + 1); block bb + 2
            // takes the place of block bb, whose rows were read two steps ago
            WF_BARRIER(wacc_);
#ifdef WX_STAMP
            const unsigned long long p0_ = __builtin_amdgcn_s_memtime();
#endif
            WFP_WRITE((bb + 2) % kWfRA, bb & 1);
            // (the producers, who have eight steps of slack, let the courier's export and poll of this step into the CU's memory queue first:
            // measured -2 % on the kernel at 256^3 with 2..6, nothing with 8, +5 % with 16)
            __builtin_amdgcn_s_sleep(kWfProdNap);
            WFP_LOAD((bb + 2) % kWfRA);
#ifdef WX_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pacc_ += __builtin_amdgcn_s_memtime() - p0_;
#endif
            WF_BARRIER(wacc_);
        }
    }
#ifdef WX_STAMP
    if (ln == 0 && wg < 4096) { g_wf_wait[wg * 16 + 5 + pw] = wacc_; if (pw == 0) g_wf_wait[wg * 16 + 14] = pacc_; }
#endif
#undef WFP_LOAD
#undef WFP_WRITE
}

__global__ void __launch_bounds__(kWfThreads)
k_ilu0_wx(WfArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ WfPair s_pairs[64];
    __shared__ int s_exp[kWfLanes];
    __shared__ int4 s_lane[kWfLanes];                     // per lane: cnt, skew, first entry of A, ST_DFL -- for the producers
    __shared__ unsigned char s_place[kWfLanes * 8];       // per lane and row position: canonical place + 1 (0: the lane has no such position)
    __shared__ int s_cnt[4], s_total;
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = (unsigned)atomicAdd(&A.ctrl[0], 1);
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
    tlo &= ~7;                                                        // step % 8 = position in the unrolled loops
    // the row ring and the hand-off arrays start all +0.0 (what no producer piece goes to stays that way); the cells of ones
    for (int i = t; i < kWfLds / 8; i += kWfThreads) reinterpret_cast<double *>(lds)[i] = 0.0;
    if (t < 64) { WfPair z; z.idx0 = 0; z.stride = 0; z.sk = 0; z.cnt = 0; z.at0 = 0; z.atm = 0; z.klast = -1; z.sh = 0; z.hasT = 0; z.astart = -1; z.pw = -1; s_pairs[t] = z; }
    if (t < kWfLanes) s_exp[t] = -1;
    if (t < 4) s_cnt[t] = 0;
    __syncthreads();
    if (t < 2 * kWfH) *reinterpret_cast<double *>(lds + kWfX + (unsigned)((t * kWfRow + kWfCell) * 8)) = 1.0;
    if (t < kWfLanes) {
        const int slot = wg * kThreads + t;
        const int32_t *T = A.ltab + (size_t)slot * kStTab;
        const int nd = T[ST_ND], cnt = T[ST_CNT];
        int cls[3]; bool ring[3];
        bool ok = wx_lane_ok(T, t, false) && wf_lane_ok(T, A.ltabB, A.uslot);
        (void)wr_classify(T, t, false, cls, ring);
        WfLane W;
        W.xB = W.xC = kWfX + (unsigned)((kWfH * kWfRow + kWfCell) * 8);
        W.tB = W.tC = kWfTB + (unsigned)((kWfH * kWfRow + kWfCell) * 8);
        W.ringC = true;
        W.src16 = ((t - 16) & 63) * 4;
        bool isg[3];
        WfPair gp[3];
        unsigned xg[3], tg[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int sw = T[ST_SRC + j];
            const int ty = (j < nd && cnt > 0) ? (sw & 3) : ST_NONE;
            const int os = sw >> 2;
            const int q = ty != ST_NONE ? T[ST_Q + j] : -1;
            isg[j] = ty == ST_GHOST;
            xg[j] = 0; tg[j] = 0;
            WfPair d; d.idx0 = 0; d.stride = 0; d.sk = T[ST_SKEW]; d.cnt = cnt > 0 ? cnt : 0; d.at0 = 0; d.atm = 0; d.klast = -1; d.sh = 0; d.hasT = 0; d.astart = -1; d.pw = -1;
            if (ty == ST_LOCAL || ty == ST_GHOST) {
                // the transposed entry: which entry right of the diagonal of the pivot row, and where its owner hands it on
                const int pu = A.uslot[os];
                const int32_t *TPB = A.ltabB + (size_t)(pu < 0 ? 0 : pu) * kStTab;
                int pc[3]; bool pr[3];
                (void)wr_classify(TPB, pu & 255, true, pc, pr);
                const int qs = (q >= 0 && pu >= 0) ? wr_slot_of(q == 0 ? pc[0] : (q == 1 ? pc[1] : pc[2]), true) : -1;
                if (qs != 1 && qs != 2) ok = false;                  // (a'B or a'C of the pivot row: what the lanes hand on)
                if (ty == ST_LOCAL) {
                    const int lane = os & 255, dt = T[ST_DT + j];
                    if (dt < 1 || dt > kWfH - 1) ok = false;
                    xg[j] = kWfX + (unsigned)(((kWfH - dt) * kWfRow + lane) * 8);
                    tg[j] = (qs == 2 ? kWfTC : kWfTB) + (unsigned)(((kWfH - dt) * kWfRow + lane) * 8);
                } else {
                    const int pw = os >> 8;
                    const int32_t *TP = A.ltab + (size_t)os * kStTab;
                    const int E = A.xw[pw * 4];
                    const int kap = T[ST_KAP + j];
                    d.stride = E;
                    d.idx0 = A.xw[pw * 4 + 3] + (kap + TP[ST_SKEW] - T[ST_SKEW] - A.xw[pw * 4 + 1]) * E + A.xe[os];
                    const int flp = TP[ST_DFL];
                    const int mp = flp >> 4;
                    d.hasT = q >= 0 ? 1 : 0;
                    d.atm = 8 * mp;
                    d.at0 = (unsigned)A.val_shift + 8u * (unsigned)(TP[ST_P0] - ((flp >> 2) & 1) + TP[ST_ND] + 1 + (q < 0 ? 0 : q) + kap * mp);
                    d.klast = TP[ST_CNT] - 1 - kap;
                    d.sh = 8 * ((flp >> 3) & 1);
                }
            } else if (ty == ST_OWN) {
                if (q != 0) ok = false;                               // (the own chain: the pivot row's first entry right of its diagonal)
            }
            gp[j] = d;
        }
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
                    if (p < kWfPairs) s_pairs[p] = gp[j];
                    xg[j] = kWfX + (unsigned)((kWfH * kWfRow + kWfLanes + min(p, kWfPairs - 1)) * 8);
                    tg[j] = kWfTB + (unsigned)((kWfH * kWfRow + kWfLanes + min(p, kWfPairs - 1)) * 8);
                }
                before += __popcll(bal[j]);
            }
        }
        W.hasB = W.hasC = W.hasUB = W.hasUC = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (cls[j] == WR_B) { W.hasB = true; W.tB = tg[j]; if (ring[j]) W.xB = xg[j]; }
            if (cls[j] == WR_C) { W.hasC = true; W.tC = tg[j]; if (ring[j]) W.xC = xg[j]; else W.ringC = false; }
        }
        {
            const int su = cnt > 0 ? A.uslot[slot] : -1;
            int bc[3] = {WR_NONE, WR_NONE, WR_NONE};
            if (su >= 0) {
                bool br[3];
                (void)wr_classify(A.ltabB + (size_t)su * kStTab, su & 255, true, bc, br);
#pragma unroll
                for (int q = 0; q < 3; ++q) { if (bc[q] == WR_B) W.hasUB = true; if (bc[q] == WR_C) W.hasUC = true; }
            } else if (cnt > 0) {
                ok = false;
            }
            // for the producers: the lane's fields, and where each position of its rows goes in the canonical record
            const int fld = T[ST_DFL], ndU = fld & 3;
            s_lane[t] = make_int4(cnt, T[ST_SKEW], T[ST_P0], fld);
#pragma unroll
            for (int pos = 0; pos < 8; ++pos) {
                int place = -1;
                if (pos < nd) { const int c = pos == 0 ? cls[0] : (pos == 1 ? cls[1] : cls[2]); place = c == WR_NONE ? -1 : wr_slot_of(c, false); }
                else if (pos == nd) place = 3;
                else if (pos <= nd + ndU && pos - nd - 1 < 3) { const int q = pos - nd - 1; const int c = q == 0 ? bc[0] : (q == 1 ? bc[1] : bc[2]); place = c == WR_NONE ? -1 : 4 + wr_slot_of(c, true); }
                s_place[t * 8 + pos] = (unsigned char)(place + 1);
            }
        }
        {
            const int xe = A.xe[slot];
            if (cnt > 0 && xe >= 0 && xe < kWfLanes) s_exp[xe] = t;
        }
        __syncthreads();
        if ((t == 0 && s_total > kWfPairs) || !ok) atomicExch(&A.ctrl[1], 1);  // (the analysis does not let such a schedule through)
#ifdef WX_STAMP
        if (t == 0 && wg < 4096) g_wf_tl[wg * 4] = __builtin_amdgcn_s_memrealtime();
#endif
        wf_consumer(A, lds, wg, W, tlo, thi);
#ifdef WX_STAMP
        if (t == 0 && wg < 4096) g_wf_tl[wg * 4 + 3] = __builtin_amdgcn_s_memrealtime();
#endif
    } else if (t < kWfLanes + 64) {
        __syncthreads();
        __syncthreads();
        const WfPair P = s_pairs[t - kWfLanes];
        const unsigned long long *idle = reinterpret_cast<const unsigned long long *>(A.ltab + (size_t)wg * kThreads * kStTab);
        wf_courier(A, idle, lds, P, tlo, thi, wg, s_exp);
    } else {
        __syncthreads();
        __syncthreads();
        wf_producer(A, lds, wg, (t - kWfLanes - 64) >> 6, tlo, thi, s_lane, s_place);
    }
}


// =============================================================================================
// ILU(0), wave-exchange form, round 6: k_ilu0_wa -- fed by LDS-DMA, no barrier in its loop, other workgroups reading ahead for it.
//
// k_ilu0_wx (above) took 1.0 ms at 256^3 for three rounds.  What this round's stamps and experiments found about it:
//   (1) a tile at work is bounded by what ONE CU can have in flight at HBM latency: 25-30 GB/s of loads + stores, whether 4 tiles
//       work on an otherwise idle chip or 120 (4096 x 32 x 32: four tiles, 0.73 us per step; 8192 x 16 x 16, one tile whose data the
//       Infinity Cache still holds from the run before: 0.44).  A tile moves 30 KB per step, so a step takes about 1.1 us, and the
//       path through the launch is 30 hand-overs x 21 steps + 289 steps.  Neither the XCD a tile runs on, nor polling by the
//       waiting tiles, nor the shader clock (2.35 GHz throughout) matter;
//   (2) the same tile reads A at nearly twice that pace when A's values come from the Infinity Cache (256 x 96 x 96, A touched just
//       before: 0.49 -> 0.31 ms) -- and at any time most CUs are idle: their tile has not begun or has ended;
//   (3) s_barrier counts every wave of the workgroup: with helper waves that touch memory at the step's barrier, the waves on the
//       chain stood there for whichever of them was late (0.2 of 0.6 us per step).
// So:
//   * four LOADER waves (one per consumer wave) issue buffer_load_dwordx4 ... lds: A's values go from HBM into LDS as they lie, a
//     128-byte window per lane and block of two steps (eight threads x 16 bytes = one memory burst per lane; the window of block b
//     starts at the lane's row of step 2 b rounded down to 16 bytes and advances by the lane's two rows): no register, no
//     ds_write, no wait for data in any wave but a counted vmcnt in the loader.  A ring of four blocks per lane;
//   * a window's eight 16-byte pieces are ROTATED by the lane ((lane >> 1) & 7; the source address of a DMA thread is per thread, its
//     LDS destination is not: thread order), so that the 16 lanes the LDS serves at a time read 16 different bank groups;
//   * the consumers read their row's seven entries where the window holds them: seven ds_read_b64 at lane-constant addresses (by step
//     parity; an entry the lane's rows do not have is read from a cell of zeros) straight into the canonical order {aC, aB} {aA, d}
//     {a'A, a'B} {a'C} -- from there on a step is k_ilu0_wx's, bit for bit;
//   * the courier is three waves, one per kind of traffic (a wave's memory operations retire in issue order: behind a write-through
//     store or a load of A, a poll is late): the POLLER only polls, the EXPORTER stores the border pivots and fetches the imports'
//     transposed entries eight steps ahead, the PREFETCHER (below) reads ahead for other tiles;
//   * NO BARRIER in the loop: every wave runs freely and waits only for what it really needs, through counters in LDS (all in
//     STEPS: "everything up to and including this step is done"):
//         cprog[w]  consumer wave w has finished the step (its LDS writes included)
//         lfull[w]  the rows up to this step are in wave w's ring (loader w, behind its counted vmcnt)
//         ipub      the poller has put the imported pivots up to this step into the import ring
//         tpub      the exporter has put the imports' transposed entries up to this step into their slots
//         eprog     the exporter has read (and stored) the border pivots of this step
//     A consumer wave checks all it depends on with ONE ds_read_b32 and one compare per step: lane i reads counter i and compares
//     it with the step plus the lane's margin (wave w - 1 one step ahead: its hand-over values are two steps old; wave w + 1 and
//     the exporter not more than the history of the hand-off arrays behind; rows two steps ahead, imports one).  The counters are
//     read half a step early: what is checked is a little stale, which only matters when a producer is less than that ahead --
//     then the wave reads again until it is;
//   * the hand-off arrays are kept once (a lane's source of `dt` steps ago is one of four precomputed addresses), which is what
//     lets 128 KB of ring fit beside them.
// MODE 1 (experiment, ILUPP_WD_MODE=1): the pivot recurrence alone -- of the records only {a'C, u_rr} is stored (results are wrong by
// design; what VERDICT r5 asked to be measured: the chain without the record stream).
// =============================================================================================
#ifndef WD_NP
#define WD_NP 2
#endif
#ifndef WA_DMA_AUX
#define WA_DMA_AUX 0
#endif
static constexpr int kWdSH = 2;                            // steps a delivery is ahead of the step that uses it
static constexpr int kWdWaveBlk = 64 * 128 + 16;           // a wave's windows of one block, and 16 bytes of zeros (what a row position the lane does not have reads)
// the value of step (v - dt) of `src` in a hand-off array whose slots are `row` doubles long, as read for step v (mod 4)
__device__ __forceinline__ void wd_addr4(unsigned a[4], const unsigned base, const int row, const int src, const int dt)
{
#pragma unroll
    for (int v = 0; v < 4; ++v) a[v] = base + (unsigned)((((v - dt) & 3) * row + src) * 8);
}

// NCW consumer waves (4: a 16 x 16 patch of lines, one workgroup per CU), D ring blocks of two steps per lane (4).  The loops are
// unrolled over U steps, a multiple of the ring (2 D steps) and of the hand-off arrays and the import ring (4 steps).
// (NCW = 2, D = 3 -- half the lanes per workgroup, two workgroups per CU, on schedules of 16 x 8 patches, to spread the tiles at work
// over more CUs -- was built, ran bit-exact on 3-D grids and was measured at 256^3: 1.2 ms against 1.0.  Two tiles that work at the
// same time on one CU share what it can have in flight, and a path through the launch has 46 hand-overs instead of 30.  The
// launch path for it is gone; the configuration stays a template parameter.)
template <int NCW, int D>
struct WaCfg {
    static constexpr int NL = NCW * 64;                        // lanes
    static constexpr int U = D == 4 ? 8 : 12;
    static constexpr int R = 4;                                // steps of the import ring
    static constexpr int NP = D == 4 ? 8 : 6;                  // steps the poller asks ahead / the exporter loads A's transposed entries ahead
    static constexpr int WaveRing = D * kWdWaveBlk;
    static constexpr int RowL = NL + 8;                        // doubles per slot of the pivot and a'C arrays: lanes, [NL] the cell of ones / zeros
    static constexpr int RowX = NL + 64 + 8;                   // ... of the a'B array: lanes, the imports' transposed entries, [NL + 64] zeros
    static constexpr unsigned X = (unsigned)(NCW * WaveRing);
    static constexpr unsigned XI = X + 4u * RowL * 8u;
    static constexpr unsigned TB = XI + (unsigned)R * 64u * 8u;
    static constexpr unsigned TC = TB + 4u * RowX * 8u;
    static constexpr unsigned Cnt = TC + 4u * RowL * 8u;
    static constexpr int Lds = (int)Cnt + 128;
    static constexpr int Threads = NL + 128 + NCW * 64 + 64;   // consumers, poller, exporter, loaders, prefetcher
    static_assert(U % (2 * D) == 0 && U % 4 == 0 && U % NP == 0, "slots are immediates of the unrolled loops");
    static_assert(Lds * (4 / NCW) <= 160 * 1024 - 256 * (4 / NCW), "the LDS of a CU");
};
enum { WA_CP = 0, WA_LF = 4, WA_IP = 8, WA_TP = 9, WA_EP = 10, WA_DEAD = 11, WA_BIG = 12, WA_WARM = 13,          // (set anew for every unit)
       WA_CHAIN = 14, WA_UM = 16, WA_UA = 17, WA_UB = 18, WA_BARC = 20, WA_BARG = 21 };
static constexpr unsigned kWaSpinLimit = 1u << 24;

// (LDS words other waves write: ordered with compiler barriers -- `volatile` would turn them into flat, system-scope accesses)
__device__ __forceinline__ int wa_ld32(const unsigned char *lds, const unsigned off)
{
    asm volatile("" ::: "memory");
    const int v = *reinterpret_cast<const int *>(lds + off);
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void wa_st32(unsigned char *lds, const unsigned off, const int v)
{
    asm volatile("" ::: "memory");
    *reinterpret_cast<int *>(lds + off) = v;
    asm volatile("" ::: "memory");
}
template <class C> __device__ __forceinline__ int wa_cnt(const unsigned char *lds, const int i) { return wa_ld32(lds, C::Cnt + 4u * (unsigned)i); }
template <class C> __device__ __forceinline__ void wa_set(unsigned char *lds, const int i, const int v) { wa_st32(lds, C::Cnt + 4u * (unsigned)i, v); }
// the waves of the roles meet (s_barrier would count the prefetcher's wave too, which is somewhere else): a count and a generation in LDS
template <class C>
__device__ __forceinline__ void wa_bar(unsigned char *lds, const int nwaves)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) {
        int *cnt = reinterpret_cast<int *>(lds + C::Cnt + 4u * WA_BARC);
        const int g = wa_cnt<C>(lds, WA_BARG);
        if (atomicAdd(cnt, 1) == nwaves - 1) {
            wa_set<C>(lds, WA_BARC, 0);
            wa_set<C>(lds, WA_BARG, g + 1);
        } else {
            unsigned spins = 0;
            while (wa_cnt<C>(lds, WA_BARG) == g && ++spins < (1u << 26)) __builtin_amdgcn_s_sleep(1);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// a helper wave waits until the slowest consumer wave has finished step `s` (lanes 0..NCW-1 read one counter each)
template <class C, int NCW>
__device__ __forceinline__ bool wa_wait_consumers(unsigned char *lds, const int s, int32_t *ctrl)
{
    const int ln = threadIdx.x & 63;
    unsigned spins = 0;
    for (;;) {
        const int c = wa_cnt<C>(lds, ln < NCW ? WA_CP + ln : WA_BIG);
        if (__builtin_amdgcn_ballot_w64(c < s) == 0) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {
            if (spins > kWaSpinLimit) { atomicExch(&ctrl[1], 1); wa_set<C>(lds, WA_DEAD, 1); }
            if (wa_cnt<C>(lds, WA_DEAD) != 0) return false;
        }
    }
}

template <int U>
struct WaLane {
    unsigned xB[4], xC[4];        // pivot hand-off: where the stand-in of class B / C FOR a step = v (mod 4) is read (the cell of ones without an entry)
    unsigned tB[4], tC[4];        // transposed entries of the class B / C elimination FOR a step = v (mod 4) (a cell of zeros without one)
    unsigned ra[2][7];
    unsigned ca; int coff;        // the counter this lane checks and its margin: counter - coff >= step
    bool ringC;
    int src16;
    bool hasB, hasC, hasUB, hasUC;
};

// RP: a REPLAY of steps [tlo, thi) of tile `wg` (MODE 2): the pivots are read, not computed -- the chain has stored them --, and the
// three quarters of the records the chain left out are written: {lC, lB} {lA, 1} and {a'A, a'B}
template <int MODE, int NCW, int D, bool RP>
__device__ __forceinline__ void wa_consumer(const WfArgs &A, unsigned char *lds, const int wg, const WaLane<WaCfg<NCW, D>::U> W, const int tlo, const int thi)
{
    typedef WaCfg<NCW, D> C;
    constexpr int U = C::U;
    typedef double v2dd __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], sk = T[ST_SKEW];
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(A.pkL + (size_t)base * 2048, 0, nchw * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(A.pkU + (size_t)base * 2048, 0, nchw * 2048, 0x00020000);
    unsigned vout = (unsigned)(tlo - tminw) * 2048u + (unsigned)ln * 16u;
    const bool compactL = (A.flags & 8) != 0;
    const unsigned cl2 = 1024u - (unsigned)ln * 8u;
    typedef unsigned int v2w2_ __attribute__((ext_vector_type(2)));
    const unsigned xown = C::X + (unsigned)t * 8u, tbown = C::TB + (unsigned)t * 8u, tcown = C::TC + (unsigned)t * 8u;
    // (RP) the lane's stored pivot of step s_: write-through by the chain, 1 where the lane had no row or the wave no chunk
    typedef unsigned int v4w_ __attribute__((ext_vector_type(4)));
#define WA_LDW(s_) __builtin_amdgcn_raw_buffer_load_b128(rU, (unsigned)((s_) - tminw) * 2048u + 1024u + (unsigned)ln * 16u, 0, 16)
#define WA_W(r_, s_) (((unsigned)((s_) - tminw) < (unsigned)nchw) ? __builtin_bit_cast(v2dd, r_).y : 1.0)
#define WA_ROW(blk_, par_, r0_, r1_, r2_, r3_)                                                                     \
    do {                                                                                                           \
        const unsigned o_ = (unsigned)(blk_) * kWdWaveBlk;                                                         \
        (r0_).x = st_lds(lds, W.ra[par_][0] + o_); (r0_).y = st_lds(lds, W.ra[par_][1] + o_);                      \
        (r1_).x = st_lds(lds, W.ra[par_][2] + o_); (r1_).y = st_lds(lds, W.ra[par_][3] + o_);                      \
        (r2_).x = st_lds(lds, W.ra[par_][4] + o_); (r2_).y = st_lds(lds, W.ra[par_][5] + o_);                      \
        (r3_).x = st_lds(lds, W.ra[par_][6] + o_); (r3_).y = 0.0;                                                  \
    } while (0)
#define WA_HAND_T(h4_, r2_, r3_)                                                                                   \
    do {                                                                                                           \
        *reinterpret_cast<double *>(lds + tbown + (unsigned)(h4_) * (C::RowX * 8)) = (r2_).y;                      \
        *reinterpret_cast<double *>(lds + tcown + (unsigned)(h4_) * (C::RowL * 8)) = (r3_).x;                      \
    } while (0)
    const int fl = T[ST_DFL];
    const int kF = ((fl >> 2) & 1) ? 0 : -1, kL = ((fl >> 3) & 1) ? cnt - 1 : -1;
    const bool fCB = W.hasC && W.hasB, fCA = W.hasC && !W.hasB, fBA = W.hasB;
    const bool lCB = W.hasUC && W.hasUB, lCA = W.hasUC && !W.hasUB, lBA = W.hasUB;
#define WA_ENDS(kk_, r0_, r1_, r2_, r3_)                                                                          \
    do {                                                                                                           \
        const bool f_ = (kk_) == kF, l_ = (kk_) == kL;                                                             \
        if (__builtin_amdgcn_ballot_w64(f_ || l_) != 0) {                                                          \
            const double c_ = (r0_).x, b_ = (r0_).y, a_ = (r1_).x, ua_ = (r2_).x, ub_ = (r2_).y, uc_ = (r3_).x;    \
            (r0_).x = f_ ? (fCB ? b_ : (fCA ? a_ : c_)) : c_;                                                      \
            (r0_).y = f_ ? (fBA ? a_ : b_) : b_;                                                                   \
            (r1_).x = f_ ? 0.0 : a_;                                                                               \
            (r2_).x = l_ ? 0.0 : ua_;                                                                              \
            (r2_).y = l_ ? (lBA ? ua_ : ub_) : ub_;                                                                \
            (r3_).x = l_ ? (lCB ? ub_ : (lCA ? ua_ : uc_)) : uc_;                                                  \
        }                                                                                                          \
    } while (0)
    // (no barrier from here on: the loaders' first two blocks, the first imports and their transposed entries are waited for as
    // every later one is -- the check of a step "tlo - 1")
    bool dead = false;
    {
        unsigned spins = 0;
        for (;;) {
            const int c0 = wa_ld32(lds, W.ca);
            if (__builtin_amdgcn_ballot_w64(c0 - W.coff < tlo - 1) == 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                if (spins > kWaSpinLimit) { atomicExch(&A.ctrl[1], 1); wa_set<C>(lds, WA_DEAD, 1); }
                if (wa_cnt<C>(lds, WA_DEAD) != 0) { dead = true; break; }
            }
        }
    }
    asm volatile("" ::: "memory");
    v2dd c0_, c1_, c2_, c3_;
    v2dd n0_, n1_, n2_, n3_;
    WA_ROW(0, 0, c0_, c1_, c2_, c3_);
    WA_ROW(0, 1, n0_, n1_, n2_, n3_);
    WA_ENDS(tlo - sk, c0_, c1_, c2_, c3_);
    WA_HAND_T(0, c2_, c3_);
    double bB = st_lds(lds, W.xB[0]), bC = st_lds(lds, W.xC[0]);
    // (the transposed entries of the first step: the wave's own lanes' -- written just above, a wave's LDS operations stay in order --,
    // the imports' behind tpub; what another wave would hand on for it is two steps old: before the first step, zeros)
    double tB = st_lds(lds, W.tB[0]), tC = st_lds(lds, W.tC[0]);
    double w3prev = 1.0, upA = 0.0;
    double qC = 1.0;
    v4w_ wq[2];                                             // (RP) the stored pivots of the next two steps
    if (RP) {
        // the two steps before: what the other waves' lanes read of them (two steps old), this lane's own last pivot
        const v4w_ r1 = WA_LDW(tlo - 1), r2 = WA_LDW(tlo - 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) wq[i] = WA_LDW(tlo + i);
        const double w1 = WA_W(r1, tlo - 1), w2 = WA_W(r2, tlo - 2);
        *reinterpret_cast<double *>(lds + xown + (unsigned)((tlo - 1) & 3) * (C::RowL * 8)) = w1;
        *reinterpret_cast<double *>(lds + xown + (unsigned)((tlo - 2) & 3) * (C::RowL * 8)) = w2;
        w3prev = w1;
        qC = wx_from_lane(W.src16, w1);
    }
    asm volatile("" ::: "memory");
    if (ln == 0) wa_set<C>(lds, WA_CP + wv, tlo - 1);
    if (RP) {
        // (bB, bC above were read before the wave below had put its two steps in: again, behind the first step's check)
        unsigned spins = 0;
        for (;;) {
            const int c0 = wa_ld32(lds, W.ca);
            if (__builtin_amdgcn_ballot_w64(c0 - W.coff < tlo) == 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                if (spins > kWaSpinLimit) { atomicExch(&A.ctrl[1], 1); wa_set<C>(lds, WA_DEAD, 1); }
                if (wa_cnt<C>(lds, WA_DEAD) != 0) { dead = true; break; }
            }
        }
        asm volatile("" ::: "memory");
        // (only what another wave of the tile has put there: an import was read above, before this wave said so -- the poller takes
        // the slot back once every wave has)
        if (W.xB[0] < C::XI) bB = st_lds(lds, W.xB[0]);
        if (W.xC[0] < C::XI) bC = st_lds(lds, W.xC[0]);
    }
    int k = tlo - sk;
    int cv = wa_ld32(lds, W.ca);
#ifdef WX_STAMP
    unsigned long long nslow_ = 0, cslow_ = 0;
#endif
    for (int tb = tlo; tb < thi; tb += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = tb + u;
            // everything this step reads from or writes to LDS is there / free (counters read in the middle of the step before)
            if (!dead && __builtin_amdgcn_ballot_w64(cv - W.coff < s) != 0) {
#ifdef WX_STAMP
                const unsigned long long q0_ = __builtin_amdgcn_s_memtime();
                ++nslow_;
#endif
                unsigned spins = 0;
                for (;;) {
                    cv = wa_ld32(lds, W.ca);
                    if (__builtin_amdgcn_ballot_w64(cv - W.coff < s) == 0) break;
                    if ((++spins & 1023u) == 0) {
                        if (spins > kWaSpinLimit) { atomicExch(&A.ctrl[1], 1); wa_set<C>(lds, WA_DEAD, 1); }
                        if (wa_cnt<C>(lds, WA_DEAD) != 0) { dead = true; break; }
                    }
                }
#ifdef WX_STAMP
                cslow_ += __builtin_amdgcn_s_memtime() - q0_;
#endif
            }
            asm volatile("" ::: "memory");                  // (no LDS access of the step in front of its check)
            const double pC = W.ringC ? bC : qC;
            const double pB = wx_dpp_shr1(bB, w3prev);
            v2dd m0_, m1_, m2_, m3_;
            WA_ROW(((u + 2) >> 1) % D, u & 1, m0_, m1_, m2_, m3_);
            const double nB = st_lds(lds, W.xB[(u + 1) & 3]);
            const double nC = st_lds(lds, W.xC[(u + 1) & 3]);
            const double ntB = RP ? 0.0 : st_lds(lds, W.tB[(u + 1) & 3]);
            const double ntC = RP ? 0.0 : st_lds(lds, W.tC[(u + 1) & 3]);
            const bool valid = (unsigned)k < (unsigned)cnt;
#ifdef WX_STAMP
            if (!RP && t == 0 && wg < 4096 && k == 0) g_wf_tl[wg * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            if (!RP && t == 0 && wg < 4096 && k == cnt - 1) g_wf_tl[wg * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
            const double aA = c1_.x, tA = k == 0 ? 0.0 : upA;
            const double uA = c2_.x;
            // u_rr = a_rr - sum (a_rk / u_kk) a_kr, eliminations in ascending k: classes C, B, A (ILU0.hpp:47-62)
            const double lC = c0_.x / pC, lB = c0_.y / pB;
            // (the counters for the next step's check: asked for here, half a step before they are looked at)
            cv = wa_ld32(lds, W.ca);
            const double lA = aA / w3prev;
            double w = c1_.y;
            if (!RP) {
                w = w - lC * tC;
                w = w - lB * tB;
                w = w - lA * tA;
                const unsigned long long wb = st_bits(w);
                if ((wb & ~3ull) == (kSentinel & ~3ull)) w = st_dbl(kCanonNaN);
            }
            double w3 = valid ? w : 1.0;
            if (RP) {
                w3 = WA_W(wq[u & 1], s);
                wq[u & 1] = WA_LDW(s + 2);
            }
            *reinterpret_cast<double *>(lds + xown + (unsigned)(u & 3) * (C::RowL * 8)) = w3;
            qC = wx_from_lane(W.src16, w3);
            w3prev = w3; upA = c2_.x;
            {
                typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
                v2dd la, lb, ua, ub;
                la.x = lC; la.y = lB; lb.x = lA; lb.y = 1.0;
                ua.x = uA; ua.y = c2_.y; ub.x = c3_.x; ub.y = w3;
                if (MODE == 0 || RP) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, la), rL, vout, 0, 2);
                    // (flags bit 3: compact L records -- lA alone, 8 bytes per lane: four memory lines of a step's sixteen less)
                    if (compactL) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2w2_, lA), rL, vout + cl2, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, lb), rL, vout + 1024u, 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, ua), rU, vout, 0, 2);
                }
                if (RP) {
                } else if (MODE == 2) {
                    // (write-through: the finishers on other CUs read it; at most 15 of this wave's stores are ever on their way)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, ub), rU, vout + 1024u, 0, 16);
                    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, ub), rU, vout + 1024u, 0, 2);
                }
                vout += 2048u;
            }
            WA_ENDS(k + 1, n0_, n1_, n2_, n3_);
            if (!RP) WA_HAND_T((u + 1) & 3, n2_, n3_);
            // this wave's step is done: its LDS writes are in front of the counter's (a wave's LDS operations stay in order)
            asm volatile("" ::: "memory");
            if (ln == 0) wa_set<C>(lds, WA_CP + wv, s);
            c0_ = n0_; c1_ = n1_; c2_ = n2_; c3_ = n3_;
            n0_ = m0_; n1_ = m1_; n2_ = m2_; n3_ = m3_;
            bB = nB; bC = nC; tB = ntB; tC = ntC;
            ++k;
        }
    }
#ifdef WX_STAMP
    if (!RP && ln == 0 && wg < 4096) { g_wf_wait[wg * 16 + wv] = cslow_; if (wv == 0) g_wf_wait[wg * 16 + 13] = nslow_; }
#endif
    // (every store of this wave has arrived: the word the exporter's last progress and the finishers of this tile's leftovers wait for)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ln == 0) wa_set<C>(lds, WA_CP + wv, 0x7ffffff8);
#undef WA_ROW
#undef WA_HAND_T
#undef WA_ENDS
#undef WA_LDW
#undef WA_W
}

// the poller: the imported pivots of step j into slot j mod 8 of the import ring, as far ahead of the consumers as the earlier
// workgroups (and the ring) allow
template <int NCW, int D>
__device__ __forceinline__ void wa_poller(const WfArgs &A, const unsigned long long *idle, unsigned char *lds, const WfPair P,
                                          const int tlo, const int thi, const int wg, const bool publish)
{
    typedef WaCfg<NCW, D> C;
    constexpr int NP = C::NP, SH = kWdSH, U = C::U;
    const int ln = threadIdx.x & 63;
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(A.xch);
    const unsigned span = (unsigned)P.cnt;
    const unsigned hoX = C::XI + (unsigned)(ln * 8);
    unsigned long long gq[NP];
#define WAC_ADDR(k_) ((unsigned)(k_) < span ? src + (P.idx0 + ((k_) + P.sk) * P.stride) : idle)
#pragma unroll
    for (int g = 0; g < NP; ++g) { gq[g] = ld_agent_u64(WAC_ADDR(tlo + g - P.sk)); asm volatile("" ::: "memory"); }
    bool dead = false;
#ifdef WX_STAMP
    unsigned long long nmiss_ = 0, nspin_ = 0;
#endif
    // Warm start.  Every workgroup this one reads from has begun (its first export is there): this one begins in some twenty steps.
    // A wave of its own (wa_helper) takes that time to ask for the rows behind the ring's -- a tile's first steps are the ones the
    // next tile waits for, and what one CU has in flight at HBM latency feeds them at half their pace.
    {
        unsigned spins = 0;
        for (;;) {
            const unsigned long long v = P.astart >= 0 ? ld_agent_u64(src + P.astart) : 0ull;
            if (__builtin_amdgcn_ballot_w64(v == kSentinel) == 0) break;
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 255u) == 0) {
                if (spins > kStSpinLimit) { atomicExch(&A.ctrl[1], 1); wa_set<C>(lds, WA_DEAD, 1); }
                const int e = ld_agent_i32(&A.ctrl[1]);
                if (spins > kStSpinLimit || e != 0) { dead = true; break; }
            }
        }
        if (ln == 0) { wa_set<C>(lds, WA_WARM, 1); if (A.prog && publish) st_agent_i32(&A.prog[wg], 0); }
    }
    // the value of step tlo_ + i_: wait for it, put it into the ring, ask for the one eight steps on; a value that was not there at the
    // first look means this workgroup has caught up with the one it reads from: everything asked for meanwhile was asked too early,
    // so the whole window is asked for again (one trip for all of it, not one per step)
#define WAC_DELIVER(i_)                                                                                              \
    do {                                                                                                             \
        const int k = tlo_ + (i_) - P.sk;                                                                            \
        const bool need = (unsigned)k < span;                                                                        \
        unsigned long long v = gq[(i_) % NP];                                                                        \
        bool late_ = false;                                                                                          \
        if (!dead) {                                                                                                 \
            unsigned spins = 0;                                                                                      \
            while (__builtin_amdgcn_ballot_w64(need && v == kSentinel) != 0) {                                       \
                late_ = true;                                                                                        \
                WAC_STAT();                                                                                          \
                if (need && v == kSentinel) v = ld_agent_u64(WAC_ADDR(k));                                           \
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                                  \
                __builtin_amdgcn_s_sleep(1);                                                                         \
                if ((++spins & 255u) == 0) {                                                                         \
                    if (spins > kStSpinLimit) { atomicExch(&A.ctrl[1], 1); wa_set<C>(lds, WA_DEAD, 1); }                \
                    const int e = ld_agent_i32(&A.ctrl[1]);                                                          \
                    __builtin_amdgcn_s_waitcnt(0x0F70);                                                              \
                    if (spins > kStSpinLimit || e != 0 || wa_cnt<C>(lds, WA_DEAD) != 0) { dead = true; break; }         \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        *reinterpret_cast<unsigned long long *>(lds + hoX + (unsigned)((i_) & 3) * 512u) = v;                        \
        if (ln == 0) wa_set<C>(lds, WA_IP, tlo_ + (i_));                                                                \
        if (late_) {                                                                                                 \
            _Pragma("unroll") for (int g_ = 1; g_ < NP; ++g_) gq[((i_) + g_) % NP] = ld_agent_u64(WAC_ADDR(k + g_)); \
        }                                                                                                            \
        gq[(i_) % NP] = ld_agent_u64(WAC_ADDR(k + NP));                                                              \
    } while (0)
#ifdef WX_STAMP
#define WAC_STAT() do { ++nspin_; if (spins == 0) ++nmiss_; } while (0)
#else
#define WAC_STAT() do { } while (0)
#endif
    {
        const int tlo_ = tlo;
#pragma unroll
        for (int i = 0; i < SH; ++i) WAC_DELIVER(i);
    }
    const int thiR = tlo + (thi - tlo + U - 1) / U * U;
    for (int tb = tlo; tb < thiR; tb += U) {
        const int tlo_ = tb;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // the slot of step j held step j - 4, which the consumers read during step j - 5
            if (!dead && !wa_wait_consumers<C, NCW>(lds, tb + u + SH - 5, A.ctrl)) dead = true;
            WAC_DELIVER(u + SH);
        }
    }
#undef WAC_DELIVER
#undef WAC_STAT
#undef WAC_ADDR
#ifdef WX_STAMP
    if (ln == 0 && wg < 4096) { g_wf_wait[wg * 16 + 4] = nspin_; g_wf_wait[wg * 16 + 14] = nmiss_; g_wf_wait[wg * 16 + 15] = (unsigned long long)(thi - tlo); }
#endif
    if (dead && ln == 0) atomicExch(&A.ctrl[1], 1);
}

// the exporter: the border pivots of step s behind the consumers' step s; the imports' transposed entries four steps ahead
template <int NCW, int D>
__device__ __forceinline__ void wa_exporter(const WfArgs &A, unsigned char *lds, const WfPair P, const int tlo, const int thi, const int wg,
                                            const int elane, const bool publish)
{
    typedef WaCfg<NCW, D> C;
    constexpr int NA = C::NP, U = C::U;
    typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
    const int ln = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = wf_rsrc(A);
    const unsigned span = (unsigned)P.cnt;
    const unsigned hoT = C::TB + (unsigned)((C::NL + ln) * 8);
    const int E = __builtin_amdgcn_readfirstlane(A.xw[wg * 4]);
    const int xrow0 = A.xw[wg * 4 + 3] + (tlo - A.xw[wg * 4 + 1]) * E;
    const unsigned ea = C::X + (unsigned)((elane >= 0 ? elane : C::NL) * 8);
    const int thiR = tlo + (thi - tlo + U - 1) / U * U;
    // (clamped to the allocation, as wf_courier's)
    const long long xroom = (long long)A.xch_len - (long long)xrow0;
    const long long xwant = (long long)(thi - tlo) * (long long)E;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(A.xch + (xroom > 0 ? xrow0 : 0), 0, (int)(8 * (xroom > 0 ? (xwant < xroom ? xwant : xroom) : 0)), 0x00020000);
    unsigned vx = elane >= 0 ? (unsigned)ln * 8u : 0xfffffff0u;
    const unsigned dvx = elane >= 0 ? (unsigned)E * 8u : 0u;
    if (E > 64 && ln == 0) atomicExch(&A.ctrl[1], 1);                 // (the analysis does not let such a schedule through)
    double ga[NA];
#define WAC_AT(k_) (((unsigned)(k_) < span && P.hasT) ? P.at0 + (unsigned)((k_) * P.atm) - ((k_) == P.klast ? (unsigned)P.sh : 0u) : 0xfffffff0u)
#define WAC_LDAT(k_) __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, WAC_AT(k_), 0, 0))
    // steps tlo .. tlo + 3 before the consumers start, then one step per pass, four steps ahead
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const double v = WAC_LDAT(tlo + g - P.sk);
        *reinterpret_cast<double *>(lds + hoT + (unsigned)g * (C::RowX * 8)) = v;
    }
#pragma unroll
    for (int g = 0; g < NA; ++g) {
        ga[g] = WAC_LDAT(tlo + 4 + g - P.sk);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, 0.0), rx, 0xfffffff0u, 0, 16);      // (the way in looks like a pass of the loop)
        asm volatile("" ::: "memory");
    }
    if (ln == 0) { wa_set<C>(lds, WA_TP, tlo + 3); }
    bool dead = false;
    for (int tb = tlo; tb < thiR; tb += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = tb + u;
            if (!dead && !wa_wait_consumers<C, NCW>(lds, s, A.ctrl)) dead = true;
            {
                const double v = st_lds(lds, ea + (unsigned)(u & 3) * (C::RowL * 8));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, v), rx, vx, 0, 16);          // sc1: write-through
                vx += dvx;
            }
            // the transposed entries of step s + 4 take the slot of step s (read during step s - 1)
            *reinterpret_cast<double *>(lds + hoT + (unsigned)(u & 3) * (C::RowX * 8)) = ga[u % NA];
            ga[u % NA] = WAC_LDAT(s + 4 + NA - P.sk);
            if (ln == 0) { wa_set<C>(lds, WA_EP, s); wa_set<C>(lds, WA_TP, s + 4); }
            // (for the prefetchers of other workgroups: how far this tile is, in steps from its first)
            if ((u & 7) == 0 && ln == 0 && A.prog && publish) st_agent_i32(&A.prog[wg], s - tlo);
        }
    }
    if (A.prog && publish) { (void)wa_wait_consumers<C, NCW>(lds, 0x7ffffff0, A.ctrl); if (ln == 0) st_agent_i32(&A.prog[wg], 0x3fffffff); }
#undef WAC_AT
#undef WAC_LDAT
}

// a loader wave (wd_loader's windows), paced by its consumer wave's progress instead of the barrier
template <int NCW, int D>
__device__ __forceinline__ void wa_loader(const WfArgs &A, unsigned char *lds, const int wg, const int lw, const int tlo, const int thi)
{
    typedef WaCfg<NCW, D> C;
    constexpr int U = C::U;
    const int ln = threadIdx.x & 63, j = ln & 7, gi = ln >> 3;
    const __amdgpu_buffer_rsrc_t rs = wf_rsrc(A);
    unsigned g[8], S[8];
    const int b0 = tlo >> 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int w = 8 * q + gi;
        const int32_t *T = A.ltab + (size_t)(wg * kThreads + lw * 64 + w) * kStTab;
        const int cnt = T[ST_CNT], sk = T[ST_SKEW], p0 = T[ST_P0], fl = T[ST_DFL];
        const int ownL = (fl >> 2) & 1, m = fl >> 4;
        const unsigned Cu = 8u * (unsigned)(p0 - ownL - sk * m) + (unsigned)A.val_shift;
        const int piece = (j + ((w >> 1) & 7)) & 7;
        const bool on = cnt > 0 && (piece < 7 || (Cu & 15u) + 16u * (unsigned)m > 112u);
        S[q] = on ? 16u * (unsigned)m : 0u;
        g[q] = on ? (Cu & ~15u) + (unsigned)b0 * S[q] + 16u * (unsigned)piece : 0xfffffff0u;
    }
    typedef __attribute__((address_space(3))) void lds_void;
#define WAL_ISSUE(slot_)                                                                                             \
    do {                                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + lw * C::WaveRing + (slot_) * kWdWaveBlk + q * 1024), 16, g[q], 0, 0, WA_DMA_AUX); \
            g[q] += S[q];                                                                                            \
        }                                                                                                            \
        asm volatile("" ::: "memory");                                                                               \
    } while (0)
    // (blocks b0 .. b0 + D - 1; the first two have landed before the consumers' first reads)
    WAL_ISSUE(0); WAL_ISSUE(1); WAL_ISSUE(2); if (D == 4) WAL_ISSUE(3);
    if (D == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (ln == 0) wa_set<C>(lds, WA_LF + lw, tlo + 3);
    const int thiR = tlo + (thi - tlo + U - 1) / U * U;
    bool dead = false;
#ifdef WX_STAMP
    unsigned long long iacc_ = 0, vacc_ = 0, sacc_ = 0;
#endif
#ifndef WA_NO_PAIR
    // Two blocks at a time, the two windows of a lane right behind each other: the memory line they share is asked for once (the
    // windows start where the rows start, so consecutive windows of a lane overlap in a line; profiles/r06_cu_stream.txt: 37 against
    // 30 GB/s of new bytes per CU).  The ring then has two blocks in flight and four steps of lead for the first of a pair instead of
    // three and six -- and the kernel takes 0.84 ms instead of 0.90 at 256^3 (-DWA_NO_PAIR: one block every two steps).
#define WAL_ISSUE2(s0_, s1_)                                                                                         \
    do {                                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + lw * C::WaveRing + (s0_) * kWdWaveBlk + q * 1024), 16, g[q], 0, 0, WA_DMA_AUX); \
            g[q] += S[q];                                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + lw * C::WaveRing + (s1_) * kWdWaveBlk + q * 1024), 16, g[q], 0, 0, WA_DMA_AUX); \
            g[q] += S[q];                                                                                            \
        }                                                                                                            \
        asm volatile("" ::: "memory");                                                                               \
    } while (0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ln == 0) wa_set<C>(lds, WA_LF + lw, tlo + 7);
    for (int tb = tlo; tb < thiR; tb += U) {
#pragma unroll
        for (int bb = 1; bb < U / 2; bb += 2) {
            const int need = tb + 2 * bb - 1;                       // blocks bb - 1 and bb have been read: their slots take blocks + D
            if (!dead) {
                unsigned spins = 0;
                while (wa_cnt<C>(lds, WA_CP + lw) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023u) == 0) {
                        if (spins > kWaSpinLimit) { atomicExch(&A.ctrl[1], 1); wa_set<C>(lds, WA_DEAD, 1); }
                        if (wa_cnt<C>(lds, WA_DEAD) != 0) { dead = true; break; }
                    }
                }
            }
            // the pair before has landed (asked for four steps ago): said so before this pair's sixteen instructions go out
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ln == 0) wa_set<C>(lds, WA_LF + lw, tb + 2 * bb + 5);
            WAL_ISSUE2((bb - 1) % D, bb % D);
        }
    }
#undef WAL_ISSUE2
#else
    for (int tb = tlo; tb < thiR; tb += U) {
#pragma unroll
        for (int bb = 0; bb < U / 2; ++bb) {
            // block b (steps tb + 2 bb, + 1): its slot takes block b + D once the wave has read its rows (during the two steps before);
            // block b + 2 is published when it has landed (the D - 2 blocks behind it may be on their way: eight instructions each)
            const int need = tb + 2 * bb - 1;
#ifdef WX_STAMP
            const unsigned long long s0_ = __builtin_amdgcn_s_memtime();
#endif
            if (!dead) {
                unsigned spins = 0;
                while (wa_cnt<C>(lds, WA_CP + lw) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023u) == 0) {
                        if (spins > kWaSpinLimit) { atomicExch(&A.ctrl[1], 1); wa_set<C>(lds, WA_DEAD, 1); }
                        if (wa_cnt<C>(lds, WA_DEAD) != 0) { dead = true; break; }
                    }
                }
            }
#ifdef WX_STAMP
            const unsigned long long i0_ = __builtin_amdgcn_s_memtime();
            sacc_ += i0_ - s0_;
#endif
            WAL_ISSUE(bb % D);
#ifdef WX_STAMP
            const unsigned long long v0_ = __builtin_amdgcn_s_memtime();
            iacc_ += v0_ - i0_;
#endif
            if (D == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#ifdef WX_STAMP
            vacc_ += __builtin_amdgcn_s_memtime() - v0_;
#endif
            if (ln == 0) wa_set<C>(lds, WA_LF + lw, tb + 2 * bb + 5);
        }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef WX_STAMP
    if (ln == 0 && wg < 4096 && lw == 0) { g_wf_wait[wg * 16 + 5] = sacc_; g_wf_wait[wg * 16 + 6] = iacc_; g_wf_wait[wg * 16 + 7] = vacc_; }
#endif
#undef WAL_ISSUE
}

// The prefetcher.  What bounds a tile at work is what ONE CU can have in flight at HBM latency (25-30 GB/s: measured with 4 tiles on an
// idle chip as with 256); the same tile reads A at twice that pace when A's values come from the Infinity Cache (256 x 96 x 96, A
// touched just before: 0.49 -> 0.31 ms).  Most CUs are idle at any time -- their tile has not begun or has ended --, so one wave per
// workgroup spends that time reading ahead FOR THE TILES AT WORK: one 4-byte load per 128-byte line of the rows a tile needs in the
// next kPfLead steps brings the lines into the memory-side cache, where the tile's own loads find them.  Who asks for what: every
// tile publishes its progress (prog[tile], by its exporter, every eight steps); a block of eight steps of a tile is claimed with an
// atomic add on claim[tile] (from -1; block 0 is what the tile's loaders bring before its first step); a tile counts as begun, for
// this purpose, when its poller says that every workgroup it reads from has begun -- some twenty steps before its first.  While
// its own tile works the wave sleeps (it would take from what its CU can have in flight).
#ifndef WA_PF_LEAD
#define WA_PF_LEAD 32
#endif
static constexpr int kPfLead = WA_PF_LEAD;                   // steps ahead of a tile's published progress that are asked for

template <int NCW, int D>
__device__ __forceinline__ unsigned wa_pf_block(const WfArgs &A, const __amdgpu_buffer_rsrc_t rs, const int tile, const int blk)
{
    const int ln = threadIdx.x & 63;
    // the tile's first step (as its own workgroup computes it)
    int tl = 0x7fffffff, th = -0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t *w4 = A.wtab + (size_t)(tile * 4 + q) * 4;
        const int a = w4[1], b = w4[2];
        if (b > 0) { tl = min(tl, a); th = max(th, a + b); }
    }
    tl &= WaCfg<NCW, D>::U == 8 ? ~7 : ~3;
    const int s0 = tl + 8 * blk;
    if (th <= tl || s0 >= th + 2) return 0u;
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
        const int32_t *T = A.ltab + (size_t)(tile * kThreads + 64 * i + ln) * kStTab;
        const int cnt = T[ST_CNT], fl = T[ST_DFL], m = fl >> 4;
        const unsigned Cu = 8u * (unsigned)(T[ST_P0] - ((fl >> 2) & 1) - T[ST_SKEW] * m) + (unsigned)A.val_shift;
        const unsigned lo = Cu + 8u * (unsigned)m * (unsigned)s0, hi = lo + 64u * (unsigned)m;      // eight rows
        unsigned o = cnt > 0 ? (lo & ~127u) : 0xfffffff0u;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const unsigned oo = (cnt > 0 && o < hi) ? o : 0xfffffff0u;
            acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, oo, 0, 0);
            o += 128u;
        }
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// MODE 2: the chains store a quarter of the records, REPLAYS write the rest.  The waves on the chain of a tile store {a'C, u_rr} only
// (write-through: other CUs read it); everything else of a row's records follows from A and the pivots without any recurrence --
// l = a / u_kk, the strict upper part of U is A's (ILU0.hpp:47-62 for rows whose eliminations meet them on the diagonal only) -- and
// is written by a workgroup whose own tile has ended: it takes a unit of kRpSteps steps of any tile that is kFinMargin steps past
// them and runs the same waves over it once more (wa_unit<RP>) -- loaders, poller, the four waves -- with the pivots read instead
// of computed: no hand-over to wait for, the rows mostly still in the Infinity Cache.  That takes 48 of the 64 record bytes per row
// off the CU that is bounded by what it can have in flight while it is on the critical path, and puts them on CUs that would idle.
// (The chain waves keep at most 15 of their stores unacknowledged: what a replay reads has arrived.  The arithmetic of a replay is
// the chain's, the same divisions on the same operands: bit-identical records.)
// ---------------------------------------------------------------------------------------------
static constexpr int kFinMargin = 16;
#ifndef WA_RP_STEPS
#define WA_RP_STEPS 32
#endif
static constexpr int kRpSteps = WA_RP_STEPS;

// the first step of a tile and the number of its blocks of eight steps (as its own workgroup computes them)
template <int NCW, int D>
__device__ __forceinline__ void wa_tile_span(const WfArgs &A, const int tile, int *tl_out, int *nblk_out)
{
    int tl = 0x7fffffff, th = -0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t *w4 = A.wtab + (size_t)(tile * 4 + q) * 4;
        const int a = w4[1], b = w4[2];
        if (b > 0) { tl = min(tl, a); th = max(th, a + b); }
    }
    constexpr int U = WaCfg<NCW, D>::U;
    tl &= U == 8 ? ~7 : ~3;
    *tl_out = tl;
    *nblk_out = th > tl ? ((th - tl + U - 1) / U * U) / 8 : 0;
}

// (wave 0 of a workgroup between units) the next unit to replay -> WA_UM (the tile; -1: none is left anywhere), WA_UA, WA_UB
template <int NCW, int D>
__device__ __forceinline__ void wa_claim_unit(const WfArgs &A, unsigned char *lds, const int wg)
{
    typedef WaCfg<NCW, D> C;
    const int ln = threadIdx.x & 63;
    const int nwg = (int)gridDim.x;
    int32_t *prog = A.prog, *claimR = A.prog + 2 * nwg;
    unsigned spins = 0;
    for (;;) {
        bool open = false;
        for (int t0 = 0; t0 < nwg; t0 += 64) {
            const int tt = t0 + ((ln + wg) & 63);
            bool ready = false;
            int cr = 0, tl = 0, nb = 0;
            if (tt < nwg) {
                const int pr = ld_agent_i32(&prog[tt]);
                cr = ld_agent_i32(&claimR[tt]);
                wa_tile_span<NCW, D>(A, tt, &tl, &nb);
                const int nunits = (nb * 8 + kRpSteps - 1) / kRpSteps;
                if (cr + 1 < nunits) {
                    open = true;
                    ready = pr >= 0x3fffffff || pr >= kRpSteps * (cr + 2) + kFinMargin;
                }
            }
            const unsigned long long bw = __builtin_amdgcn_ballot_w64(ready);
            if (bw != 0) {
                const int L = __builtin_ctzll(bw);
                int got = 0;
                if (ln == L) got = atomicCAS(&claimR[tt], cr, cr + 1) == cr ? 1 : 0;
                got = __builtin_amdgcn_readlane(got, L);
                if (got) {
                    if (ln == L) {
                        const int ua = tl + kRpSteps * (cr + 1);
                        wa_set<C>(lds, WA_UM, tt); wa_set<C>(lds, WA_UA, ua); wa_set<C>(lds, WA_UB, min(ua + kRpSteps, tl + 8 * nb));
                    }
                    return;
                }
                open = true;
            }
        }
        if (__builtin_amdgcn_ballot_w64(open) == 0 || wa_cnt<C>(lds, WA_DEAD) != 0 || ++spins > (1u << 20)) { if (ln == 0) wa_set<C>(lds, WA_UM, -1); return; }
        __builtin_amdgcn_s_sleep(64);
    }
}

// The prefetcher: one wave of a workgroup, on its own from the launch on.  Before its tile begins and after its chain has ended it reads
// ahead for the tiles at work (asleep in between: it would take from what its CU can have in flight while that is on the critical path).
template <int MODE, int NCW, int D, bool PF>
__device__ __forceinline__ void wa_helper(const WfArgs &A, unsigned char *lds, const int wg)
{
    typedef WaCfg<NCW, D> C;
    const int ln = threadIdx.x & 63;
    if (!A.prog) return;
    const int nwg = (int)gridDim.x;
    int32_t *prog = A.prog, *claim = A.prog + nwg;
    const __amdgpu_buffer_rsrc_t rs = wf_rsrc(A);
    unsigned acc = 0;
    bool own_done = false, own_warm = false;
    unsigned idle = 0;
    // the (first two) workgroups this one imports from: they work right before it does, and until it begins it has nothing else to do
    int up1 = -1, up2 = -1;
    {
        int my_pw = -1;
        for (int i = 0; i < NCW && my_pw < 0; ++i) {
            const int32_t *T = A.ltab + (size_t)(wg * kThreads + 64 * i + ln) * kStTab;
            const int nd = T[ST_ND], cnt = T[ST_CNT];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int sw = T[ST_SRC + j];
                if (j < nd && cnt > 0 && (sw & 3) == ST_GHOST && my_pw < 0) my_pw = (sw >> 2) >> 8;
            }
        }
        const unsigned long long b1 = __builtin_amdgcn_ballot_w64(my_pw >= 0);
        if (b1 != 0) {
            up1 = __builtin_amdgcn_readlane(my_pw, __builtin_ctzll(b1));
            const unsigned long long b2 = __builtin_amdgcn_ballot_w64(my_pw >= 0 && my_pw != up1);
            if (b2 != 0) up2 = __builtin_amdgcn_readlane(my_pw, __builtin_ctzll(b2));
        }
    }
#ifdef WX_STAMP
    unsigned long long npf_ = 0, nscan_ = 0; long long lead_ = 0;
#endif
    for (;;) {
#ifdef WX_STAMP
        ++nscan_;
#endif
        // this workgroup's own tile: hands off while its chain works
        if (!own_warm) {
            if (wa_cnt<C>(lds, WA_WARM) != 0) { own_warm = true; continue; }
        } else if (!own_done) {
            if (wa_cnt<C>(lds, WA_DEAD) != 0) break;
            if (wa_cnt<C>(lds, WA_CHAIN) != 2 && !(MODE != 2 && wa_cnt<C>(lds, WA_CP) >= 0x7ffffff0)) { __builtin_amdgcn_s_sleep(127); continue; }
            own_done = true;
            // (a launch of more workgroups than the chip holds: this one's CU is wanted by the next)
            if (!(A.flags & 2)) break;
        }
        // FIVE other tiles, fixed.  The (two) tiles this one imports from: they begin some twenty steps before it and it is idle till
        // then -- a tile's first steps, which the next tile waits for, always have somebody.  And three far ones: half the launch
        // away, an eighth of a line of tiles away, and both (on a box grid in 16 x 16 patches: eight patches on in z, in y, in both --
        // eight or sixteen hand-overs earlier or later: they work mostly while this one does not).  The next block inside a mate's
        // window is claimed with a compare-and-swap on claim[mate] (few waves ask for a tile: no crowd)
        bool did = false, allfin = true;
#pragma unroll 1
        for (int j = 0; j < 5; ++j) {
            const int off = j == 2 ? nwg / 2 : (j == 3 ? nwg / 32 : nwg / 2 + nwg / 32);
            const int mate = j == 0 ? up1 : (j == 1 ? up2 : (wg + off) % nwg);
            if (mate < 0 || mate == wg || (j == 3 && nwg < 64) || (j < 2 && own_done)) continue;
            const int pr = ld_agent_i32(&prog[mate]);
            if (pr < 0x3fffffff) allfin = false;
            if (pr < 0 || pr >= 0x3fffffff) continue;
            int cl = ld_agent_i32(&claim[mate]);
            // (blocks the tile has passed already are not asked for)
            if (8 * (cl + 2) <= pr) cl = pr / 8 - 1;
            if (8 * (cl + 2) > pr + kPfLead) continue;
            int got = 0;
            if (ln == 0) { const int old = ld_agent_i32(&claim[mate]); got = (old <= cl && atomicCAS(&claim[mate], old, cl + 1) == old) ? 1 : 0; }
            got = __builtin_amdgcn_readfirstlane(got);
            if (!got) continue;
            acc ^= wa_pf_block<NCW, D>(A, rs, mate, cl + 2);
            did = true;
#ifdef WX_STAMP
            ++npf_;
            lead_ += (long long)(8 * (cl + 2) - pr);
#endif
        }
        if (allfin && own_done) break;
        if (!did) {
            __builtin_amdgcn_s_sleep(64);
            if (own_done && ++idle > (1u << 18)) break;
            if (!own_warm && wa_cnt<C>(lds, WA_DEAD) != 0) break;
        }
    }
#ifdef WX_STAMP
    if (ln == 0 && wg < 4096) { g_wf_wait[wg * 16 + 9] = npf_; g_wf_wait[wg * 16 + 10] = nscan_; g_wf_wait[wg * 16 + 11] = (unsigned long long)lead_; }
#endif
    if (acc == 0x9e3779b9u && A.val_bytes == 0xfffffff3u) atomicExch(&A.ctrl[1], (int)acc);        // (the loads above are not dead code)
}

// (per-XCD ticket counters in the control words: [2], [3], [9] .. [14])
__device__ __forceinline__ int wa_xcd_word(const int x) { return x < 2 ? 2 + x : 7 + x; }

// one UNIT of a workgroup's work: the chain of its own tile (RP false: tile `wg`, all of its steps), or (RP, MODE 2) the replay of steps
// [ua, ub) of some tile whose chain is past them.  The waves of the roles only (the prefetcher goes its own way): they meet at
// wa_bar, not at s_barrier.
template <int MODE, int NCW, int D, bool RP>
__device__ __forceinline__ void wa_unit(const WfArgs &A, unsigned char *lds, int *s_cnt, int &s_total, const int wg, const int ua, const int ub)
{
    typedef WaCfg<NCW, D> C;
    constexpr int U = C::U, NL = C::NL;
    constexpr int NROLE = NL + 128 + NCW * 64;
    WfPair *s_pairs = reinterpret_cast<WfPair *>(lds);
    int *s_exp = reinterpret_cast<int *>(lds + 64 * sizeof(WfPair));
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
    tlo &= U == 8 ? ~7 : ~3;                                          // step % 4 (% 8) = position in the unrolled loops
    if (RP) { tlo = ua; thi = min(thi, ub); }
    // the hand-off arrays start all +0.0 (the windows are whatever they are: a lane reads its own rows or the cells of zeros behind
    // them, which no DMA touches), the scratch at the ring's start is the set-up's
    for (int i = (int)(C::X / 8) + t; i < (int)(C::Cnt / 8); i += NROLE) reinterpret_cast<double *>(lds)[i] = 0.0;
    if (t < 4) s_cnt[t] = 0;
    wa_bar<C>(lds, NROLE / 64);
    if (t < 64) { WfPair z; z.idx0 = 0; z.stride = 0; z.sk = 0; z.cnt = 0; z.at0 = 0; z.atm = 0; z.klast = -1; z.sh = 0; z.hasT = 0; z.astart = -1; z.pw = -1; s_pairs[t] = z; }
    if (t < NL) s_exp[t] = -1;
    if (t < 4) *reinterpret_cast<double *>(lds + C::X + (unsigned)((t * C::RowL + NL) * 8)) = 1.0;
    // (a consumer wave has "done step tlo - 1" when it has read what its first step starts from)
    if (t < 14) wa_set<C>(lds, t, t == WA_BIG ? 0x7fffffff : ((t == WA_DEAD || t == WA_WARM) ? 0 : (t < WA_LF ? tlo - 2 : ((RP && (t == WA_TP || t == WA_EP)) ? 0x7ffffff0 : tlo - 1))));
    wa_bar<C>(lds, NROLE / 64);
    if (t < NL) {
        const int slot = wg * kThreads + t;
        const int32_t *T = A.ltab + (size_t)slot * kStTab;
        const int nd = T[ST_ND], cnt = T[ST_CNT];
        int cls[3]; bool ring[3];
        bool ok = wx_lane_ok(T, t, false) && wf_lane_ok(T, A.ltabB, A.uslot);
        (void)wr_classify(T, t, false, cls, ring);
        // (a workgroup of fewer lanes than the schedule's slots: the slots left over own no rows)
        for (int l2 = t + NL; l2 < kThreads; l2 += NL) if (A.ltab[(size_t)(wg * kThreads + l2) * kStTab + ST_CNT] > 0) ok = false;
        WaLane<U> W;
#pragma unroll
        for (int v = 0; v < 4; ++v) W.xB[v] = W.xC[v] = C::X + (unsigned)(NL * 8);
#pragma unroll
        for (int v = 0; v < 4; ++v) W.tB[v] = W.tC[v] = C::TB + (unsigned)((NL + 64) * 8);
        W.ringC = true;
        W.src16 = ((t - 16) & 63) * 4;
        bool isg[3];
        WfPair gp[3];
        unsigned xg[3][4], tg[3][4];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int sw = T[ST_SRC + j];
            const int ty = (j < nd && cnt > 0) ? (sw & 3) : ST_NONE;
            const int os = sw >> 2;
            const int q = ty != ST_NONE ? T[ST_Q + j] : -1;
            isg[j] = ty == ST_GHOST;
#pragma unroll
            for (int v = 0; v < 4; ++v) xg[j][v] = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) tg[j][v] = 0;
            WfPair d; d.idx0 = 0; d.stride = 0; d.sk = T[ST_SKEW]; d.cnt = cnt > 0 ? cnt : 0; d.at0 = 0; d.atm = 0; d.klast = -1; d.sh = 0; d.hasT = 0; d.astart = -1; d.pw = -1;
            if (ty == ST_LOCAL || ty == ST_GHOST) {
                const int pu = A.uslot[os];
                const int32_t *TPB = A.ltabB + (size_t)(pu < 0 ? 0 : pu) * kStTab;
                int pc[3]; bool pr[3];
                (void)wr_classify(TPB, pu & 255, true, pc, pr);
                const int qs = (q >= 0 && pu >= 0) ? wr_slot_of(q == 0 ? pc[0] : (q == 1 ? pc[1] : pc[2]), true) : -1;
                if (qs != 1 && qs != 2) ok = false;
                if (ty == ST_LOCAL) {
                    const int lane = os & 255, dt = T[ST_DT + j];
                    // (what crosses a wave is two steps old, not more, and comes from the wave below: the margins of the counters are
                    // made for that)
                    if (dt < 1 || dt > 2) ok = false;
                    if ((lane >> 6) != (t >> 6) && (lane >> 6) != (t >> 6) - 1) ok = false;
                    if (lane >= NL) ok = false;
                    wd_addr4(xg[j], C::X, C::RowL, lane, dt);
                    if (qs == 2) wd_addr4(tg[j], C::TC, C::RowL, lane, dt); else wd_addr4(tg[j], C::TB, C::RowX, lane, dt);
                } else {
                    const int pw = os >> 8;
                    const int32_t *TP = A.ltab + (size_t)os * kStTab;
                    const int E = A.xw[pw * 4];
                    const int kap = T[ST_KAP + j];
                    d.stride = E;
                    d.idx0 = A.xw[pw * 4 + 3] + (kap + TP[ST_SKEW] - T[ST_SKEW] - A.xw[pw * 4 + 1]) * E + A.xe[os];
                    const int flp = TP[ST_DFL];
                    const int mp = flp >> 4;
                    d.hasT = q >= 0 ? 1 : 0;
                    d.atm = 8 * mp;
                    d.at0 = (unsigned)A.val_shift + 8u * (unsigned)(TP[ST_P0] - ((flp >> 2) & 1) + TP[ST_ND] + 1 + (q < 0 ? 0 : q) + kap * mp);
                    d.klast = TP[ST_CNT] - 1 - kap;
                    d.sh = 8 * ((flp >> 3) & 1);
                    {
                        // the producer's workgroup stores its border pivots from its first step on (whether its lanes have rows yet or not)
                        int tl = 0x7fffffff;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) { const int32_t *w4 = A.wtab + (size_t)(pw * 4 + q4) * 4; if (w4[2] > 0) tl = min(tl, w4[1]); }
                        tl &= U == 8 ? ~7 : ~3;
                        d.astart = A.xw[pw * 4 + 3] + (tl - A.xw[pw * 4 + 1]) * E + A.xe[os];
                        d.pw = pw;
                    }
                }
            } else if (ty == ST_OWN) {
                if (q != 0) ok = false;
            }
            gp[j] = d;
        }
        {
            const int wv = t >> 6;
            unsigned long long bal[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) bal[j] = __builtin_amdgcn_ballot_w64(isg[j]);
            const int mine = __popcll(bal[0]) + __popcll(bal[1]) + __popcll(bal[2]);
            if ((t & 63) == 0) s_cnt[wv] = mine;
            wa_bar<C>(lds, NROLE / 64);
            int before = 0;
            for (int q = 0; q < wv; ++q) before += s_cnt[q];
            if (t == 0) s_total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (isg[j]) {
                    const int p = before + __builtin_amdgcn_mbcnt_hi((unsigned)(bal[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[j], 0));
                    if (p < 64) s_pairs[p] = gp[j];
#pragma unroll
                    for (int v = 0; v < 4; ++v) xg[j][v] = C::XI + (unsigned)(v * 512 + min(p, 63) * 8);
                    wd_addr4(tg[j], C::TB, C::RowX, NL + min(p, 63), 0);
                }
                before += __popcll(bal[j]);
            }
        }
        W.hasB = W.hasC = W.hasUB = W.hasUC = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (cls[j] == WR_B) {
                W.hasB = true;
#pragma unroll
                for (int v = 0; v < 4; ++v) W.tB[v] = tg[j][v];
                if (ring[j]) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) W.xB[v] = xg[j][v];
                }
            }
            if (cls[j] == WR_C) {
                W.hasC = true;
#pragma unroll
                for (int v = 0; v < 4; ++v) W.tC[v] = tg[j][v];
                if (ring[j]) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) W.xC[v] = xg[j][v];
                } else {
                    W.ringC = false;
                }
            }
        }
        {
            const int su = cnt > 0 ? A.uslot[slot] : -1;
            int bc[3] = {WR_NONE, WR_NONE, WR_NONE};
            if (su >= 0) {
                bool br[3];
                (void)wr_classify(A.ltabB + (size_t)su * kStTab, su & 255, true, bc, br);
#pragma unroll
                for (int q = 0; q < 3; ++q) { if (bc[q] == WR_B) W.hasUB = true; if (bc[q] == WR_C) W.hasUC = true; }
            } else if (cnt > 0) {
                ok = false;
            }
            const int fld = T[ST_DFL], ndU = fld & 3, ownL = (fld >> 2) & 1, m = fld >> 4;
            const unsigned Cu = 8u * (unsigned)(T[ST_P0] - ownL - T[ST_SKEW] * m) + (unsigned)A.val_shift;
            const int ln = t & 63, wv = t >> 6;
            const int rot = (ln >> 1) & 7;
            int posOf[7] = {-1, -1, -1, -1, -1, -1, -1};
#pragma unroll
            for (int pos = 0; pos < 7; ++pos) {
                int place = -1;
                if (pos < nd) { const int c = pos == 0 ? cls[0] : (pos == 1 ? cls[1] : cls[2]); place = c == WR_NONE ? -1 : wr_slot_of(c, false); }
                else if (pos == nd) place = 3;
                else if (pos <= nd + ndU && pos - nd - 1 < 3) { const int q = pos - nd - 1; const int c = q == 0 ? bc[0] : (q == 1 ? bc[1] : bc[2]); place = c == WR_NONE ? -1 : 4 + wr_slot_of(c, true); }
#pragma unroll
                for (int pl_ = 0; pl_ < 7; ++pl_) if (place == pl_) posOf[pl_] = pos;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int pl_ = 0; pl_ < 7; ++pl_) {
                    const unsigned B = (Cu & 15u) + (unsigned)(e * 8 * m) + 8u * (unsigned)(posOf[pl_] < 0 ? 0 : posOf[pl_]);
                    const unsigned a = (unsigned)(wv * C::WaveRing + ln * 128) + ((((B >> 4) - (unsigned)rot) & 7u) << 4) + (B & 8u);
                    W.ra[e][pl_] = (cnt > 0 && posOf[pl_] >= 0 && B < 128u) ? a : (unsigned)(wv * C::WaveRing + 64 * 128);
                    if (cnt > 0 && posOf[pl_] >= 0 && B >= 128u) ok = false;
                }
            // the counter this lane looks at, and its margin (counter - margin >= step)
            int ci = WA_BIG, co = 0;
            if (ln == 0 && wv > 0) { ci = WA_CP + wv - 1; co = -1; }       // hand-over values of wave w - 1: two steps old, read a step early
            if (ln == 1 && wv < NCW - 1) { ci = WA_CP + wv + 1; co = -2; }       // wave w + 1 still reads what this step overwrites (four slots, two steps old)
            if (ln == 2) { ci = WA_LF + wv; co = 2; }                      // the row of step s + 2
            if (ln == 3) { ci = WA_IP; co = 1; }                           // the imports of step s + 1
            if (ln == 4) { ci = WA_TP; co = 1; }
            if (ln == 5) { ci = WA_EP; co = -4; }                          // the exporter has read the pivots this step overwrites
            W.ca = C::Cnt + 4u * (unsigned)ci; W.coff = co;
        }
        {
            const int xe = A.xe[slot];
            if (cnt > 0 && xe >= 0 && xe < NL) s_exp[xe] = t;
        }
        wa_bar<C>(lds, NROLE / 64);
        if ((t == 0 && s_total > 64) || !ok) atomicExch(&A.ctrl[1], 1);
        wa_bar<C>(lds, NROLE / 64);
        for (int i = t; i < 64 * (int)sizeof(WfPair) / 8 + kThreads / 2; i += NL) reinterpret_cast<double *>(lds)[i] = 0.0;
        wa_bar<C>(lds, NROLE / 64);
#ifdef WX_STAMP
        const unsigned long long cy0_ = __builtin_amdgcn_s_memtime();
        if (!RP && t == 0 && wg < 4096) g_wf_tl[wg * 4] = __builtin_amdgcn_s_memrealtime();
#endif
        wa_consumer<MODE, NCW, D, RP>(A, lds, wg, W, tlo, thi);
#ifdef WX_STAMP
        if (!RP && t == 0 && wg < 4096) {
            g_wf_tl[wg * 4 + 3] = __builtin_amdgcn_s_memrealtime(); g_wf_wait[wg * 16 + 12] = __builtin_amdgcn_s_memtime() - cy0_;
            // where the workgroup ran: HW_ID (wave, SIMD, pipe, CU, SH, SE ...) and XCC_ID
            g_wf_wait[wg * 16 + 8] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0x1814) << 32);
        }
#endif
    } else if (t < NL + 128) {
        wa_bar<C>(lds, NROLE / 64);
        wa_bar<C>(lds, NROLE / 64);
        const WfPair P = s_pairs[t & 63];
        const int E = __builtin_amdgcn_readfirstlane(A.xw[wg * 4]);
        const int elane = ((t & 63) < E) ? s_exp[t & 63] : -1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wa_bar<C>(lds, NROLE / 64);
        wa_bar<C>(lds, NROLE / 64);
        if (t < NL + 64) {
            const unsigned long long *idle = reinterpret_cast<const unsigned long long *>(A.ltab + (size_t)wg * kThreads * kStTab);
            wa_poller<NCW, D>(A, idle, lds, P, tlo, thi, wg, !RP);
        } else if (!RP) {
            wa_exporter<NCW, D>(A, lds, P, tlo, thi, wg, elane, true);
        }
    } else if (t < NL + 128 + NCW * 64) {
        wa_bar<C>(lds, NROLE / 64);
        wa_bar<C>(lds, NROLE / 64);
        wa_bar<C>(lds, NROLE / 64);
        wa_bar<C>(lds, NROLE / 64);
        wa_loader<NCW, D>(A, lds, wg, (t - NL - 128) >> 6, tlo, thi);
    }
}

template <int MODE, int NCW, int D>
__global__ void __launch_bounds__((WaCfg<NCW, D>::Threads), (NCW == 2 ? 4 : 3))
k_ilu0_wa(WfArgs A)
{
    typedef WaCfg<NCW, D> C;
    constexpr int NL = C::NL;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int s_cnt[4], s_total;
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) {
        // Which tile: the next ticket -- or (experiment, flags bit 0; only when every workgroup of the launch is resident at once) the
        // next tile of THIS XCD's class (tile mod 8 = XCD).  The hardware starts the workgroups of one XCD in a row, so plain tickets
        // put a whole line of 16 neighbouring tiles -- which work at the same time -- behind one XCD's L2 and fabric port; by class
        // every band of the wavefront is spread over all eight.  Measured at 256^3: no difference (1.005 ms either way).
        int tile = -1;
        if (A.flags & 1) {
            const int x = (int)(__builtin_amdgcn_s_getreg(0x1814) & 7u);
            for (int i = 0; i < 8 && tile < 0; ++i) {
                const int xx = (x + i) & 7;
                const int c = atomicAdd(&A.ctrl[wa_xcd_word(xx)], 1);
                if (c * 8 + xx < (int)gridDim.x) tile = c * 8 + xx;
            }
        } else {
            tile = atomicAdd(&A.ctrl[0], 1);
        }
        s_ticket = (unsigned)tile;
    }
    __syncthreads();
    const int wg = (int)s_ticket;
    if (wg < 0) return;
    const int t = threadIdx.x;
    // all of LDS +0.0 once (the cells of zeros behind every wave's windows stay that way)
    for (int i = t; i < C::Lds / 8; i += C::Threads) reinterpret_cast<double *>(lds)[i] = 0.0;
    __syncthreads();
    constexpr int NROLE = NL + 128 + NCW * 64;
    if (t >= NROLE) {
        // the prefetcher: on its own from here
        wa_helper<MODE, NCW, D, true>(A, lds, wg);
        return;
    }
    wa_unit<MODE, NCW, D, false>(A, lds, s_cnt, s_total, wg, 0, 0);
    if (MODE != 2 || !(A.flags & 2) || !A.prog) return;
    // Replays: what the chains leave of the records, in units of kRpSteps steps of any tile that is far enough past them
    wa_bar<C>(lds, NROLE / 64);
    if (t == 0) wa_set<C>(lds, WA_CHAIN, 2);
    for (;;) {
        if (t < 64) wa_claim_unit<NCW, D>(A, lds, wg);
        wa_bar<C>(lds, NROLE / 64);
        const int um = wa_cnt<C>(lds, WA_UM), ua = wa_cnt<C>(lds, WA_UA), ub = wa_cnt<C>(lds, WA_UB);
        if (um < 0) break;
        wa_unit<MODE, NCW, D, true>(A, lds, s_cnt, s_total, um, ua, ub);
        wa_bar<C>(lds, NROLE / 64);
    }
}

static int wd_mode()
{
    static const int m = [] { const char *e = getenv("ILUPP_WD_MODE"); return e ? atoi(e) : 0; }();
    return m;
}
bool wa_on()
{
    static const bool on = getenv("ILUPP_NO_WA") == nullptr;
    return on;
}
// the factor kernel ilu0_numeric_wx launches, as a profiler names it (ilupp_hip_kernel_names)
const char *wx_factor_kernel_name()
{
    if (!wa_on()) return "k_ilu0_wx";
    if (wd_mode() == 1) return "k_ilu0_wa<1, 4, 4>";
    return getenv("ILUPP_REPLAY") != nullptr ? "k_ilu0_wa<2, 4, 4>" : "k_ilu0_wa<0, 4, 4>";
}

// what the factor kernel finds prepared, in ONE launch (three small launches in a row cost their dispatch gaps, and this chain -- not the
// proof beside it -- is what the analysis phase of a box grid lasts): the control words zero, the tiles' progress and claim words -1,
// the exchange buffer all-sentinel
__global__ void k_wa_prepare(int32_t *__restrict__ ctrl, int32_t *__restrict__ prog, const int nprog, unsigned long long *__restrict__ xch,
                             const long long nx)
{
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (i0 < 4) ctrl[i0] = 0;
    for (long long i = i0; i < nprog; i += stride) prog[i] = -1;
    for (long long i = i0; i < nx; i += stride) xch[i] = kSentinel;
}

int ilu0_numeric_wx(hipStream_t st, const DevMat &A, PackedSweep *pl, PackedSweep *pu, int32_t *d_ctrl, float *kernel_ms,
                    hipEvent_t e0, hipEvent_t e1)
{
    {
        static std::once_flag once[64];      // once per device
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_wx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWfLds));
            typedef WaCfg<4, 4> C44;
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_wa<0, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, C44::Lds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_wa<1, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, C44::Lds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_wa<2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, C44::Lds));
        });
    }
    WfArgs a;
    a.ltab = pl->ltab; a.ltabB = pu->ltab; a.uslot = pu->uslot; a.wtab = pl->wtab;
    const uintptr_t vp = reinterpret_cast<uintptr_t>(A.val);
    a.val = reinterpret_cast<const double *>(vp & ~(uintptr_t)15);
    a.val_shift = (int32_t)(vp & 15);
    a.val_bytes = (uint32_t)(A.nnz * 8 + a.val_shift);
    a.pkL = reinterpret_cast<unsigned char *>(pl->pk); a.pkU = reinterpret_cast<unsigned char *>(pu->pk);
    a.xe = pl->xe; a.xw = pl->xw; a.xch = pl->xch; a.xch_len = pl->xch_len; a.ctrl = d_ctrl;
    a.flags = 0;
    a.prog = nullptr;
    if (wa_on() && getenv("ILUPP_NO_PREFETCH") == nullptr) {
        // progress and claim words of the tiles, all -1 (one buffer per device, kept)
        static int32_t *buf[64];
        static int64_t cap[64];
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        const int d = dev & 63;
        if (cap[d] < 3 * (int64_t)pl->nwg) {
            if (buf[d]) ILUPP_HIP(pool_free(buf[d]));
            buf[d] = nullptr; cap[d] = 0;
            ILUPP_HIP(pool_malloc(&buf[d], sizeof(int32_t) * 3 * (size_t)pl->nwg));
            cap[d] = 3 * (int64_t)pl->nwg;
        }
        a.prog = buf[d];
    }
    {
        long long blocks = ((long long)pl->xch_len + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_wa_prepare, dim3((unsigned)blocks), dim3(256), 0, st, d_ctrl, a.prog, a.prog ? 3 * (int)pl->nwg : 0,
                           reinterpret_cast<unsigned long long *>(pl->xch), (long long)pl->xch_len);
    }
    {
        // does the chip hold every workgroup of the launch at once?  (Then the prefetchers stay behind their own tile's end, for the
        // tiles that still work; and, an experiment that changed nothing -- ILUPP_XCD_TICKETS=1 --, tiles are handed out by XCD.)
        static int ncu[64];
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        if (ncu[dev & 63] == 0) { int v = 0; ILUPP_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev)); ncu[dev & 63] = v > 0 ? v : -1; }
        if (ncu[dev & 63] > 0 && (int64_t)pl->nwg <= (int64_t)ncu[dev & 63]) {
            a.flags |= 2;
            if (getenv("ILUPP_XCD_TICKETS") != nullptr) {
                a.flags |= 1;
                ILUPP_HIP(hipMemsetAsync(d_ctrl + 9, 0, 6 * sizeof(int32_t), st));
            }
        }
    }
    if (pl->join_ev && pl->join_before) ILUPP_HIP(hipStreamWaitEvent(st, pl->join_ev, 0));       // (grid.hip's proof, on its side stream)
    ILUPP_HIP(hipEventRecord(e0, st));
    // compact L records when the L sweep that reads them will run (the vector-wave sweep: ILUPP_NO_COMPACT_L=1 for format 1)
    static const bool no_compact = getenv("ILUPP_NO_COMPACT_L") != nullptr;
    const bool compact = wa_on() && wd_mode() != 1 && !no_compact && wx_vec_on() && pl->vec_ok && pu->vec_ok;
    if (compact) a.flags |= 8;
    if (wa_on()) {
        typedef WaCfg<4, 4> C;
        // (MODE 2 -- the chains store a quarter of the records, replays by the workgroups whose tile has ended write the rest -- is an
        // experiment, ILUPP_REPLAY=1, where every workgroup is resident and the progress words exist: bit-identical, and slower at
        // 256^3 -- 1.07 ms against 0.93: a replay unit's set-up and its 30 KB per step cost the idle CUs more time than they have)
        static const bool nofin = getenv("ILUPP_REPLAY") == nullptr;
        if (wd_mode() == 1) hipLaunchKernelGGL((k_ilu0_wa<1, 4, 4>), dim3((unsigned)pl->nwg), dim3(C::Threads), C::Lds, st, a);
        else if ((a.flags & 2) && a.prog && !nofin) hipLaunchKernelGGL((k_ilu0_wa<2, 4, 4>), dim3((unsigned)pl->nwg), dim3(C::Threads), C::Lds, st, a);
        else hipLaunchKernelGGL((k_ilu0_wa<0, 4, 4>), dim3((unsigned)pl->nwg), dim3(C::Threads), C::Lds, st, a);
    } else {
        hipLaunchKernelGGL(k_ilu0_wx, dim3((unsigned)pl->nwg), dim3(kWfThreads), kWfLds, st, a);
    }
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[12];
    if (pl->join_ev && !pl->join_before) ILUPP_HIP(hipStreamWaitEvent(st, pl->join_ev, 0));
    ILUPP_HIP(d2h_async(st, ctrl, d_ctrl, sizeof(ctrl)));
    if (pl->arm && pl->arm_ev) {
        // (the first apply's control words and exchange buffers are made ready behind the read-back, while the host is on its way back)
        pl->fmt = compact ? 2 : 1; pu->fmt = 1;      // (what the kernel in flight writes)
        ILUPP_HIP(hipEventRecord(pl->arm_ev, st));
        pl->arm(pl->arm_ctx);
        ILUPP_HIP(event_sync(st, pl->arm_ev));
    } else {
        ILUPP_HIP(stream_sync(st));
    }
    pl->join_verdict = ctrl[8];
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    pl->fmt = compact ? 2 : 1; pu->fmt = 1;
    return ILUPP_OK;
}

#ifdef WX_STAMP
void wx_read_stamps(unsigned long long *out) { ILUPP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wx_stamp), sizeof(unsigned long long) * 16)); }
void wf_read_tl(unsigned long long *out) { ILUPP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wf_tl), sizeof(unsigned long long) * 4096 * 4)); }
void wf_read_wait(unsigned long long *out) { ILUPP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wf_wait), sizeof(unsigned long long) * 4096 * 16)); }
#endif

}  // namespace ilupp

#ifdef WX_STAMP
extern "C" int ilupp_hip_debug_wx_stamps(unsigned long long *out)
{
    try { ilupp::wx_read_stamps(out); } catch (...) { return -1; }
    return 0;
}
extern "C" int ilupp_hip_debug_wf_wait(unsigned long long *out)
{
    try { ilupp::wf_read_wait(out); } catch (...) { return -1; }
    return 0;
}
extern "C" int ilupp_hip_debug_wf_timeline(unsigned long long *out)
{
    try { ilupp::wf_read_tl(out); } catch (...) { return -1; }
    return 0;
}
#endif
