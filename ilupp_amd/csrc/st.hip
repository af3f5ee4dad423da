// ilupp_amd/csrc/st.hip -- static level-major ILU(0) and triangular sweeps for stencil-like matrices (gfx950).
//
// Same arithmetic as ilu0_lm.hip / sptrsv_lm.hip (reference ILU0.hpp:26-66: row-wise IKJ, eliminations in ascending k,
// separate multiply and subtract; sparse_implementation.h:4040-4087: sequential accumulation in stored order, division by
// the diagonal found by position), same placement (a lane owns a chain of consecutive rows, a workgroup a 16x16 patch of
// chains, row k of lane t is due at step k + skew(t)).  What is new is that NOTHING about the structure travels with the rows:
//
//   * the analysis proves, for EVERY row, that the lane's rows all look alike: the columns of row r are r + o for offsets o
//     out of a per-lane template of at most 3 offsets left and 3 right of the diagonal, each offset always produced by the
//     same lane (own previous row / a lane of the workgroup a fixed number of steps back / a lane of an earlier workgroup),
//     and every elimination meets the eliminated row on its diagonal only ("simple" rows: all 5-/7-point stencils).
//     Which template entries a row has (domain boundaries) is in the data: an entry that does not exist is kAbsent;
//   * for such rows the strictly-upper part of U is A's and l_ik = a_ik / u_kk with a_ik untouched: the ONLY recurrence of
//     the factorisation is the one of the pivots,  u_ii = a_ii - sum_k (a_ik / u_kk) * a_ki .  So the pass that brings A
//     into level-major order (k_st_rows) writes the off-diagonals of U itself and hands every row the transposed entries
//     a_ki it needs (each row scatters its upper entries into the records of the rows they meet), and the factor kernel
//     is left with what the L sweep does: one value per row handed from lane to lane;
//   * the consumer has no record decode, no tags and no polling inside a workgroup: the lane table sits in registers,
//     finished values go into ONE LDS array indexed by (step mod 8, lane) -- kept twice, 8 steps apart, so that every read
//     address is a lane constant plus an immediate -- and ONE s_barrier per step orders them (4 waves per workgroup,
//     nothing else resident: no loader waves, no importer wave);
//   * the streams are read by the consuming lane itself, 8 steps ahead, into a rotating register file (fully unrolled
//     loop), level-major and coalesced.  vmcnt retires loads and stores in issue order and hipcc only counts the
//     operations it is certain of, so EVERY memory operation of a step is unconditional (lanes or waves without a row
//     are redirected to a dump location): the waits then leave the whole read-ahead in flight;
//   * values of earlier workgroups (tile borders) are polled 4 steps ahead by the border lanes themselves (write-through
//     stores / cache-bypassing loads, the data is the flag); a value that is not there when its step comes is polled
//     again, which makes the tile fall back behind its producers by just the latency it needs.
//
// A step is bound by instruction issue (one wave per SIMD): everything that can be a lane constant is one.
// Everything the proof rejects runs on the record-decoding kernels (records_lm.hip) or the CSR kernels (any matrix).
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include <hipcub/hipcub.hpp>

#include "st_common.h"

namespace ilupp {

void FactorLM::release()
{
    if (pkA) (void)pool_free(pkA);
    if (xbase) (void)pool_free(xbase);
    if (xch) (void)pool_free(xch);
    if (xcount) (void)pool_free(xcount);
    pkA = nullptr; xbase = nullptr; xbase_len = 0; xch = nullptr; xcount = nullptr; built = false; values_packed = false; stat = false; direct = false; wxf = false; spec = false;
}

// the wave-exchange kernels and the skews they need (st_common.h); ILUPP_NO_WR=1: the schedules and kernels of round 3
static bool st_wx_on()
{
    static const bool on = getenv("ILUPP_NO_WR") == nullptr;
    return on;
}

__global__ void k_lm_ysrc(int32_t nslots, const int32_t *__restrict__ uslot, const int32_t *__restrict__ scount,
                          const int32_t *__restrict__ wtabL, const int32_t *__restrict__ skewL, int32_t *__restrict__ ysrc);   // sptrsv_lm.hip

// ---------------------------------------------------------------------------------------------
// analysis 1: lane templates.  TRI = +1: the entries left of the diagonal (forward schedule), -1: right (backward)
// ---------------------------------------------------------------------------------------------
template <int TRI>
__device__ __forceinline__ void
st_template_body(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t B, int32_t nb,
                 const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot, const int32_t *__restrict__ sfirst,
                 const int32_t *__restrict__ scount, int32_t *__restrict__ exported, int32_t *__restrict__ ltab,
                 int32_t *__restrict__ flags)
{
    const int wg = blockIdx.x, t = threadIdx.x;
    const int slot = wg * kThreads + t;
    const int cnt = scount[slot], first = sfirst[slot];
    int o0 = 0, o1 = 0, o2 = 0, s0 = 0, s1 = 0, s2 = 0, b0 = 0, b1 = 0, b2 = 0, k0 = 0, k1 = 0, k2 = 0;
    int nd = 0, bad = 0;
    for (int sample = 0; sample < 3 && cnt > 0; ++sample) {
        const int k = sample == 0 ? 0 : (sample == 1 ? cnt / 2 : cnt - 1);
        if ((sample == 1 && k == 0) || (sample == 2 && (k == 0 || k == cnt / 2))) continue;
        const int r = first + TRI * k;
        for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
            const int c = idx[q];
            if (TRI > 0 ? c >= r : c <= r) continue;
            const int o = c - r;
            if ((nd > 0 && o == o0) || (nd > 1 && o == o1) || (nd > 2 && o == o2)) continue;
            if (nd == 3) { bad = 1; continue; }
            const int b = block_of(c, B, nb, start);
            const int os = blk2slot[b];
            const int kloc = TRI > 0 ? c - start[b] : start[b + 1] - 1 - c;
            const int kp = kloc - k;
            int sw;
            if (os == slot) {
                sw = ST_OWN | (os << 2);
                if (kp != -1) bad = 1;                              // a dependency further back in the lane's own chain
            } else if ((os >> 8) == wg) {
                sw = ST_LOCAL | (os << 2);
            } else {
                sw = ST_GHOST | (os << 2);
                if ((os >> 8) >= wg) bad = 1;                       // producer not ahead of us in ticket order
                exported[os] = 1;
            }
            // insert, ascending offset (= ascending column = the reference's elimination / accumulation order)
            if (nd == 0 || (nd == 1 && o > o0) || (nd == 2 && o > o1)) {
                if (nd == 0) { o0 = o; s0 = sw; b0 = b; k0 = kp; }
                else if (nd == 1) { o1 = o; s1 = sw; b1 = b; k1 = kp; }
                else { o2 = o; s2 = sw; b2 = b; k2 = kp; }
            } else if (nd == 1 || (nd == 2 && o < o0)) {
                o2 = o1; s2 = s1; b2 = b1; k2 = k1;
                o1 = o0; s1 = s0; b1 = b0; k1 = k0;
                o0 = o; s0 = sw; b0 = b; k0 = kp;
            } else {                                                // nd == 2, o0 < o < o1
                o2 = o1; s2 = s1; b2 = b1; k2 = k1;
                o1 = o; s1 = sw; b1 = b; k1 = kp;
            }
            ++nd;
        }
    }
    if (nd == 3 && (s0 & 3) == ST_GHOST && (s1 & 3) == ST_GHOST && (s2 & 3) == ST_GHOST) bad = 1;   // the poll ring holds two
    int32_t *T = ltab + (size_t)slot * kStTab;
    T[ST_FIRST] = first; T[ST_CNT] = cnt; T[ST_SKEW] = 0; T[ST_ND] = nd;
    const int oo[3] = {o0, o1, o2}, ss[3] = {s0, s1, s2}, bb[3] = {b0, b1, b2}, kk[3] = {k0, k1, k2};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const bool in = j < nd;
        T[ST_OFF + j] = in ? oo[j] : 0;
        T[ST_SRC + j] = in ? ss[j] : 0;
        T[ST_KAP + j] = in ? kk[j] : 0;
        T[ST_DT + j] = 1;
        T[ST_SCAT + j] = -1;
        // the rows k of this lane whose entry at this offset lies where the template says (own chain: all but the first)
        int klo = 0, khi = 0;
        if (in) {
            if ((ss[j] & 3) == ST_OWN) { klo = 1; khi = cnt; }
            else if (TRI > 0) { klo = start[bb[j]] - first - oo[j]; khi = start[bb[j] + 1] - first - oo[j]; }
            else { klo = first + oo[j] - start[bb[j] + 1] + 1; khi = first + oo[j] - start[bb[j]] + 1; }
        }
        T[ST_KLO + j] = klo; T[ST_KHI + j] = khi;
    }
    T[ST_UP0] = 0;
    if (bad) atomicOr(&flags[0], 2);
}

template <int TRI>
__global__ void __launch_bounds__(kThreads)
k_st_template(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t B, int32_t nb,
              const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot, const int32_t *__restrict__ sfirst,
              const int32_t *__restrict__ scount, int32_t *__restrict__ exported, int32_t *__restrict__ ltab,
              int32_t *__restrict__ flags)
{
    st_template_body<TRI>(ptr, idx, B, nb, start, blk2slot, sfirst, scount, exported, ltab, flags);
}
// both schedules of a matrix with one launch (blockIdx.y: 0 forward, 1 backward; a launch of 256 small workgroups is 25 us of
// dependent loads whatever it does)
struct StTplArgs { int32_t B, nb; const int32_t *start, *blk2slot, *sfirst, *scount; int32_t *exported, *ltab, *flags; };
__global__ void __launch_bounds__(kThreads)
k_st_template_pair(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, StTplArgs f, StTplArgs b)
{
    if (blockIdx.y == 0) st_template_body<1>(ptr, idx, f.B, f.nb, f.start, f.blk2slot, f.sfirst, f.scount, f.exported, f.ltab, f.flags);
    else st_template_body<-1>(ptr, idx, b.B, b.nb, b.start, b.blk2slot, b.sfirst, b.scount, b.exported, b.ltab, b.flags);
}

// ---------------------------------------------------------------------------------------------
// analysis 2: skews (longest-path fixpoint over the workgroup's in-workgroup dependencies, exact: the templates hold
// for every row), steps back of every in-workgroup dependency, chunk range of each wave; FWD: the proof that each
// elimination meets the eliminated row on its diagonal only
// ---------------------------------------------------------------------------------------------
template <bool FWD>
__device__ __forceinline__ void
st_link_body(int32_t *__restrict__ ltab, const int32_t *__restrict__ ltab_u, const int32_t *__restrict__ uslot,
             int32_t *__restrict__ skew, int32_t *__restrict__ wtab, int32_t *__restrict__ flags, int *s, const bool wx, const bool bwd_sched)
{
    const int wg = blockIdx.x, t = threadIdx.x;
    const int slot = wg * kThreads + t;
    int32_t *T = ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], nd = T[ST_ND];
    int cb[3] = {0, 0, 0}, cd[3] = {0, 0, 0};
    bool loc[3] = {false, false, false};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int sw = T[ST_SRC + j];
        loc[j] = j < nd && (sw & 3) == ST_LOCAL;
        cb[j] = loc[j] ? ((sw >> 2) & 255) : t;
        // (wx: a value that does not stay in the registers of the consumer's wave is at least kWrLag steps old -- st_common.h)
        cd[j] = T[ST_KAP + j] + wr_edge_lag(t, cb[j], wx);
    }
    s[t] = 0;
    __syncthreads();
    for (int it = 0; it < 1024; ++it) {
        int v = s[t];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (loc[j]) { const int c = s[cb[j]] + cd[j]; v = c > v ? c : v; }
        v = v > kStMaxSkew ? kStMaxSkew : v;
        const int changed = v != s[t];
        __syncthreads();
        s[t] = v;
        if (!__syncthreads_or(changed)) break;
    }
    const int sk = s[t];
    int bad = sk >= kStMaxSkew ? 1 : 0;
    T[ST_SKEW] = sk;
    skew[slot] = sk;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int dt = 1;
        if (loc[j]) {
            dt = sk - s[cb[j]] - T[ST_KAP + j];
            if (dt < 1 || dt > kStH - 1) bad = 1;
        }
        T[ST_DT + j] = dt;
    }
    // may the wave-exchange kernels run this lane?  (flags[9]: no)
    if (wx) {
        if (!wx_lane_ok(T, t, bwd_sched)) atomicOr(&flags[9], 1);
    } else if (t == 0) {
        atomicOr(&flags[9], 2);
    }
    int lo = cnt > 0 ? sk : 0x7fffffff, hi = cnt > 0 ? sk + cnt : -0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
    if ((t & 63) == 0) {
        int32_t *w = wtab + (size_t)(wg * 4 + (t >> 6)) * 4;
        const int nch = hi > lo ? hi - lo : 0;
        w[0] = 0; w[1] = nch > 0 ? lo : 0; w[2] = nch; w[3] = 0;
    }
    if (FWD && cnt > 0) {
        const int su = uslot[slot];
        if (su < 0) {
            bad = 1;
        } else {
            const int32_t *TU = ltab_u + (size_t)su * kStTab;
            const int nu = TU[ST_ND];
            const int mu[3] = {TU[ST_OFF], TU[ST_OFF + 1], TU[ST_OFF + 2]};
            const int ml[3] = {T[ST_OFF], T[ST_OFF + 1], T[ST_OFF + 2]};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < nd) {
                    const int pu = uslot[T[ST_SRC + j] >> 2];
                    if (pu < 0) {
                        bad = 1;
                    } else {
                        const int32_t *TP = ltab_u + (size_t)pu * kStTab;
                        const int np = TP[ST_ND];
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            if (p < np) {
                                const int m = ml[j] + TP[ST_OFF + p];       // column of the pivot row's entry, relative to this row
                                if (m != 0) {
                                    // a match off the diagonal would change an L or U entry of this row: not a "simple" row
#pragma unroll
                                    for (int q = 0; q < 3; ++q) {
                                        if (q < nd && m == ml[q]) bad = 1;
                                        if (q < nu && m == mu[q]) bad = 1;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (bad) atomicOr(&flags[0], 4);
}

template <bool FWD>
__global__ void __launch_bounds__(kThreads)
k_st_link(int32_t *__restrict__ ltab, const int32_t *__restrict__ ltab_u, const int32_t *__restrict__ uslot,
          int32_t *__restrict__ skew, int32_t *__restrict__ wtab, int32_t *__restrict__ flags, int32_t wx)
{
    __shared__ int s[kThreads];
    st_link_body<FWD>(ltab, ltab_u, uslot, skew, wtab, flags, s, (wx & 1) != 0, (wx & 2) != 0);
}
// both schedules of an ILU(0) with one launch (blockIdx.y: 0 forward with the proof about the eliminations, 1 backward)
__global__ void __launch_bounds__(kThreads)
k_st_link_pair(int32_t *__restrict__ ltabF, int32_t *__restrict__ ltabB, const int32_t *__restrict__ uslot, int32_t *__restrict__ skewF,
               int32_t *__restrict__ skewB, int32_t *__restrict__ wtabF, int32_t *__restrict__ wtabB, int32_t *__restrict__ flagsF,
               int32_t *__restrict__ flagsB, int32_t wx)
{
    __shared__ int s[kThreads];
    if (blockIdx.y == 0) st_link_body<true>(ltabF, ltabB, uslot, skewF, wtabF, flagsF, s, wx != 0, false);
    else st_link_body<false>(ltabB, nullptr, nullptr, skewB, wtabB, flagsB, s, wx != 0, true);
}

// exclusive scan of the waves' chunk counts (one block); flags[1] = total, flags[2] = longest wave
__device__ __forceinline__ void st_scan_body(int32_t nwaves, int32_t *__restrict__ wtab, int32_t *__restrict__ flags, int *part, int *pmax)
{
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
__global__ void __launch_bounds__(kThreads)
k_st_scan(int32_t nwaves, int32_t *__restrict__ wtab, int32_t *__restrict__ flags)
{
    __shared__ int part[kThreads];
    __shared__ int pmax[kThreads];
    st_scan_body(nwaves, wtab, flags, part, pmax);
}
// both chunk scans of an ILU(0) with one launch (block 0: forward, block 1: backward)
__global__ void __launch_bounds__(kThreads)
k_st_scan_pair(int32_t nwaves, int32_t *__restrict__ wtabF, int32_t *__restrict__ flagsF, int32_t *__restrict__ wtabB, int32_t *__restrict__ flagsB)
{
    __shared__ int part[kThreads];
    __shared__ int pmax[kThreads];
    if (blockIdx.x == 0) st_scan_body(nwaves, wtabF, flagsF, part, pmax); else st_scan_body(nwaves, wtabB, flagsB, part, pmax);
}

// inverse of the slot map, and where the backward sweep finds its right-hand side (k_lm_ysrc, sptrsv_lm.hip) -- one launch
__global__ void k_st_inv_ysrc(int32_t nslots, const int32_t *__restrict__ uslot, int32_t *__restrict__ inv, const int32_t *__restrict__ scount,
                              const int32_t *__restrict__ wtabL, const int32_t *__restrict__ skewL, int32_t *__restrict__ ysrc)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nslots) return;
    const int su = uslot[f];
    if (su < 0) return;
    inv[su] = f;
    const int w = f >> 6;
    ysrc[su] = (wtabL[(size_t)w * 4] + (scount[f] - 1 + skewL[f] - wtabL[(size_t)w * 4 + 1])) * 64 + (f & 63);
}
// the arrays the analysis wants cleared, with one launch (each memset is a launch of its own: 5 us of a 450 us analysis apiece)
struct StClearList { int32_t *p[4]; int32_t v[4]; int32_t n[4]; };
__global__ void k_st_clear(StClearList L)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (L.p[q] && i < L.n[q]) L.p[q][i] = L.v[q];
}


// analysis 3 (forward lanes, after the chunk tables): where the lane's rows sit in the backward sweep's records, and
// where each of their upper entries a(r, r+o) goes: into the record of row r+o, as the transposed entry of that row's
// elimination with row r.  Both are "position of row 0 of the lane" + a fixed stride per row.
// ... and everything the rows pass needs about a lane in one 128-byte record (rtab):
//   {first, cnt, skew, nL} {oL0, oL1, oL2, nU} {oU0, oU1, oU2, up0} {kloL x3, -} {khiL x3, -} {kloU x3, -} {khiU x3, -} {scat x3, -}
__global__ void k_st_scat(int32_t nslots, int32_t *__restrict__ ltabF, const int32_t *__restrict__ ltabB,
                          const int32_t *__restrict__ uslot, const int32_t *__restrict__ inv,
                          const int32_t *__restrict__ wtabF, const int32_t *__restrict__ wtabU, int32_t *__restrict__ rtab,
                          int32_t *__restrict__ flags, const int32_t *__restrict__ Aptr, int32_t *__restrict__ dflags)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nslots) return;
    // (the direct-feed factor kernel's lane fields and lane-level checks: st_direct.hip)
    if (dflags) {
        sd_tab_lane(f, ltabF, ltabB, uslot, Aptr, dflags);
        // (dflags + 2 = flags[10]: a lane the wave-exchange factor kernel cannot run)
        if (!wf_lane_ok(ltabF + (size_t)f * kStTab, ltabB, uslot)) atomicOr(dflags + 2, 1);
    }
    int32_t *T = ltabF + (size_t)f * kStTab;
    int32_t *R = rtab + (size_t)f * 32;
    const int cnt = T[ST_CNT];
    R[0] = T[ST_FIRST]; R[1] = cnt; R[2] = T[ST_SKEW]; R[3] = T[ST_ND];
    if (cnt <= 0) return;
    const int su = uslot[f];
    if (su < 0) { atomicOr(&flags[0], 16); return; }
    const int32_t *TB = ltabB + (size_t)su * kStTab;
    const int wu = su >> 6;
    // 16-byte units; row k of the lane: - 128 k
    T[ST_UP0] = (wtabU[wu * 4] + (cnt - 1 + TB[ST_SKEW] - wtabU[wu * 4 + 1])) * 128 + (su & 63);
    const int nu = TB[ST_ND];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int sc = -1;
        if (p < nu) {
            const int cs = inv[TB[ST_SRC + p] >> 2];                 // forward slot of the chain that holds rows r + o
            if (cs < 0) {
                atomicOr(&flags[0], 16);
            } else {
                const int32_t *TC = ltabF + (size_t)cs * kStTab;
                const int o = TB[ST_OFF + p];
                int jc = -1;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < TC[ST_ND] && TC[ST_OFF + j] == -o) jc = j;
                if (jc >= 0) {
                    const int kc = T[ST_FIRST] + o - TC[ST_FIRST];               // index of row r + o in its chain, for row 0 of this lane
                    const int cw = cs >> 6;
                    const int chunk0 = wtabF[cw * 4] + (kc + TC[ST_SKEW] - wtabF[cw * 4 + 1]);
                    // 8-byte units: piece 2 + jc/2 of the chunk, half jc%2; row k of the lane: + 512 k
                    sc = (chunk0 * 256 + (2 + (jc >> 1)) * 64 + (cs & 63)) * 2 + (jc & 1);
                }
            }
        }
        T[ST_SCAT + p] = sc;
        {
            // the forward dependency p's producer lane, when it is one of the 64 lanes of the same wave: lane | (k' - k + 128) << 8
            const int sw = T[ST_SRC + p];
            const int ty = p < T[ST_ND] ? (sw & 3) : ST_NONE;
            const int os = sw >> 2;
            R[28 + p] = ((ty == ST_OWN || ty == ST_LOCAL) && (os >> 6) == (f >> 6) && T[ST_KAP + p] > -128 && T[ST_KAP + p] < 128)
                            ? ((os & 63) | ((T[ST_KAP + p] + 128) << 8)) : -1;
        }
        R[4 + p] = T[ST_OFF + p]; R[8 + p] = TB[ST_OFF + p];
        R[12 + p] = T[ST_KLO + p]; R[16 + p] = T[ST_KHI + p];
        R[20 + p] = TB[ST_KLO + p]; R[24 + p] = TB[ST_KHI + p];
    }
    R[7] = nu; R[11] = T[ST_UP0];
}

// ---------------------------------------------------------------------------------------------
// analysis 4: the proof, row by row, and the factor kernel's input.  Per (chunk, lane) 64 bytes
//   {a0,a1} {a2,a3} {t0,t1} {t2,a6}: a0..a2 the entries left of the diagonal by template position, a3 the diagonal,
//   t_j = a(k_j, r) the transposed entry of elimination j (fetched from row k_j), a6 the last upper entry;
// the off-diagonal part {a4,a5} of the row of U goes straight into the backward sweep's record.
// A block owns the 64 lanes of one wave x 8 consecutive rows of each; inside a wave 8 lanes x 8 rows, so a load
// instruction touches 8 contiguous segments of A and a store instruction 8 neighbouring places of 8 chunks.
// ---------------------------------------------------------------------------------------------
// workgroup id -> position in the work list such that each of the 8 XCDs (workgroup id modulo 8) gets a contiguous eighth
__device__ __forceinline__ int st_xcd_order(int b, int n)
{
    const int per = n >> 3, x = b & 7, q = b >> 3;
    return q < per ? x * per + q : b;                  // (the tail n % 8 stays where it is)
}

// One row of A -> the ten doubles of its records {a0,a1} {a2,a3} {t0,t1} {t2,a6} {a4,a5}.  The row's entries come from LDS (the
// lane's span), the transposed entries t_j = a(c_j, r) of the pivot rows c_j = r + oF[j] from LDS too when the pivot row is a
// row of this block, else from A.
struct StSpan { const double *v; const int *c; int len, room; };           // a row: values, columns, entries; readable places from c
__device__ __forceinline__ int st_row_find(const StSpan &s, int col)
{
    int at = -1;
#pragma unroll
    for (int i = 0; i < 7; ++i) { const int ci = s.c[min(i, s.room)]; at = (i < s.len && ci == col) ? i : at; }
    return at;
}

// A block owns 8 consecutive rows (k0 .. k0+7) of each of the 64 lanes of a wave.  The vector-memory pipe takes about one cycle
// per 64-byte piece an instruction touches, whatever it returns, and a row-per-thread read of A (56-byte rows, 16 bytes per
// instruction) touches 64 pieces per instruction: the kernel was bound by that pipe (1.1 ms on 256^3, 26 % of the wave cycles
// stalled at issue, 76 cycles per vector-memory instruction and CU).  So the 9 rows a lane contributes (one before the eight:
// the pivot row of the first) are read as ONE contiguous run, 8 threads x 16 bytes per instruction, into LDS, and the rows
// and most of the pivot rows are picked from there.
// (plain stores, not streaming ones: a wave holds its LDS and its slot until its stores are acknowledged, and the L2 acknowledges
// a write-back store sooner -- 1.13 -> 0.93 ms in an A/B on the same box; the sweeps' and the factor kernel's stores showed no such gain)
#define ST_ROWS_STORE(v, p) (*(p) = (v))
static constexpr int kStSpanMax = 64;            // entries of a lane's run (9 rows of at most 7)
__global__ void __launch_bounds__(512)
k_st_rows(const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval, int64_t nnz,
          const int32_t *__restrict__ rtab, const int32_t *__restrict__ wtab, v2d *__restrict__ pkA, v2d *__restrict__ pkU,
          int32_t *__restrict__ flags, int32_t groups)
{
    __shared__ double sv[64][kStSpanMax + 1];
    __shared__ int si[64][kStSpanMax + 1];
    __shared__ int rp[64][12];                   // row pointers of rows k0-1 .. k0+8 (the last: the end of row k0+7); -1: no such row
    __shared__ int slo[64], sfirst[64];
    // consecutive workgroups go to different XCDs, each with its own L2: hand every XCD a contiguous range of (wave, rows), so
    // that the cache lines two neighbouring runs share are fetched once
    const int bid = st_xcd_order(blockIdx.x, gridDim.x);
    const int w = bid / groups;
    const int k0 = (bid % groups) * 8;
    const int t = threadIdx.x;
    const int slot0 = (w >> 2) * kThreads + (w & 3) * 64;
    for (int q = t; q < 64 * 10; q += 512) {
        const int l = q / 10, i = q % 10;
        const v4i h = *reinterpret_cast<const v4i *>(rtab + (size_t)(slot0 + l) * 32);      // first, cnt, skew, nL
        const int kr = k0 - 1 + i;
        rp[l][i] = (kr >= 0 && kr <= h.y && k0 < h.y) ? Aptr[h.x + kr] : -1;
        if (i == 0) sfirst[l] = h.x;
    }
    // (the lane record of this thread's row: asked for here, used after the two barriers below -- one round trip less in the chain)
    const int l = t >> 3, sub = t & 7;
    const v4i *R = reinterpret_cast<const v4i *>(rtab + (size_t)(slot0 + l) * 32);
    const v4i t0 = R[0], t1 = R[1], t2 = R[2], klF = R[3], khF = R[4], klB = R[5], khB = R[6], src = R[7];
    __syncthreads();
    int bad = 0;
    {
        // the lane's run: from its first row here to the end of its last
        int lo = rp[l][0] >= 0 ? rp[l][0] : rp[l][1], hi = -1;
#pragma unroll
        for (int i = 1; i < 10; ++i) hi = rp[l][i] >= 0 ? rp[l][i] : hi;
        const int len = lo >= 0 ? hi - lo : 0;
        if (len > kStSpanMax) bad = 1;
        if (sub == 0) slo[l] = lo;
        if (!bad) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int e = (it * 8 + sub) * 2;
                if (e + 1 < len) { const D2s x = *reinterpret_cast<const D2s *>(Aval + lo + e); sv[l][e] = x.v[0]; sv[l][e + 1] = x.v[1]; }
                else if (e < len) sv[l][e] = Aval[lo + e];
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = (it * 8 + sub) * 4;
                if (e + 3 < len) {
                    const I4a x = *reinterpret_cast<const I4a *>(Aidx + lo + e);
#pragma unroll
                    for (int z = 0; z < 4; ++z) si[l][e + z] = x.v[z];
                } else {
#pragma unroll
                    for (int z = 0; z < 4; ++z) if (e + z < len) si[l][e + z] = Aidx[lo + e + z];
                }
            }
        }
    }
    __syncthreads();
    // the row of this thread: lane l, k = k0 + sub
    const int cnt = t0.y, k = k0 + sub;
    if (k < cnt && !bad) {
        const int r = t0.x + k;
        const int b = rp[l][sub + 1] - slo[l];
        const int len = rp[l][sub + 2] - rp[l][sub + 1];
        if (len > 7 || len < 1) bad = 1;
        const int ndF = t0.w, ndB = t1.w;
        const int oF[3] = {t1.x, t1.y, t1.z};
        const int oB[3] = {t2.x, t2.y, t2.z};
        const double absent = st_dbl(kAbsent);
        double a[7] = {absent, absent, absent, absent, absent, absent, absent};
        double tj[3] = {absent, absent, absent};
        int mask = 0;
        if (!bad) {
            // the row is sorted: cl entries left of the diagonal, the diagonal, the rest right of it; each side is matched against
            // its own side of the template only
            int cc[7]; double v[7];
#pragma unroll
            // (unconditional reads at clamped places, then a select: a read under a condition is a branch around it)
            for (int i = 0; i < 7; ++i) {
                const int e = min(b + i, kStSpanMax);
                const int ci = si[l][e]; const double vi = sv[l][e];
                cc[i] = i < len ? ci : 0x7fffffff; v[i] = i < len ? st_clean(vi) : 0.0;
            }
            int cl = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i) cl += cc[i] < r ? 1 : 0;
            if (cl > 3 || len - cl - 1 > 3 || len - cl - 1 < 0) bad = 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const bool in = i < cl;
                const int o = cc[i] - r;
                bool hit = false;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const bool h = in && j < ndF && o == oF[j];
                    if (h) { a[j] = v[i]; mask |= 1 << j; }
                    hit |= h;
                }
                if (in && !hit) bad = 1;                            // a column outside the lane's template
            }
            {
                const int cd = cl == 0 ? cc[0] : cl == 1 ? cc[1] : cl == 2 ? cc[2] : cc[3];
                const double vd = cl == 0 ? v[0] : cl == 1 ? v[1] : cl == 2 ? v[2] : v[3];
                if (cd == r) { a[3] = vd; mask |= 8; }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool in = cl + 1 + u < len;
                const int cu = cl == 0 ? cc[1 + u] : cl == 1 ? cc[2 + u] : cl == 2 ? cc[3 + u] : cc[4 + u];
                const double vu = cl == 0 ? v[1 + u] : cl == 1 ? v[2 + u] : cl == 2 ? v[3 + u] : v[4 + u];
                const int o = cu - r;
                bool hit = false;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const bool h = in && q < ndB && o == oB[q];
                    if (h) { a[4 + q] = vu; mask |= 16 << q; }
                    hit |= h;
                }
                if (in && !hit) bad = 1;
            }
            if (!(mask & 8)) bad = 1;
            // every entry is produced where the template says
            const int kb = cnt - 1 - k;                             // the row's index in the backward schedule
            const int kl[3] = {klF.x, klF.y, klF.z}, kh[3] = {khF.x, khF.y, khF.z};
            const int bl[3] = {klB.x, klB.y, klB.z}, bh[3] = {khB.x, khB.y, khB.z};
            const int sr[3] = {src.x, src.y, src.z};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if ((mask & (1 << j)) && (k < kl[j] || k >= kh[j])) bad = 1;
                if ((mask & (16 << j)) && (kb < bl[j] || kb >= bh[j])) bad = 1;
            }
            // t_j = a(c_j, r): from the pivot row c_j = r + oF[j] (gathered, not scattered by the row that owns it: a scattered
            // 8-byte store is a read-modify-write of a memory burst, and the records would have to be pre-filled)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (!bad && (mask & (1 << j))) {
                    const int c = r + oF[j];
                    bool done = false;
                    if (sr[j] >= 0) {
                        const int lp = sr[j] & 63, ip = k + ((sr[j] >> 8) - 128) - (k0 - 1);       // lane and row slot of the pivot row
                        if (ip >= 0 && ip <= 8 && rp[lp][ip] >= 0 && rp[lp][ip + 1] >= 0 && sfirst[lp] + (k0 - 1 + ip) == c) {
                            const int bp = rp[lp][ip] - slo[lp];
                            StSpan sp; sp.v = &sv[lp][bp]; sp.c = &si[lp][bp]; sp.len = min(rp[lp][ip + 1] - rp[lp][ip], 7); sp.room = kStSpanMax - bp;
                            const int at = st_row_find(sp, r);
                            if (at >= 0) tj[j] = st_clean(sp.v[at]);
                            done = true;
                        }
                    }
                    if (!done) {
                        const int b0 = Aptr[c];
                        const int bl_ = Aptr[c + 1] - b0;
                        const Row8 rc = load_row8(Aidx, b0, bl_ > 8 ? 8 : bl_, nnz);
                        int at_ = -1;
#pragma unroll
                        for (int i = 0; i < 8; ++i) at_ = rc.c[i] == r ? i : at_;
                        if (at_ >= 0) tj[j] = st_clean(Aval[b0 + at_]);
                    }
                }
            }
        }
        if (!bad) {
            const int c = k + t0.z - wtab[(size_t)w * 4 + 1];
            v2d *p = pkA + ((size_t)wtab[(size_t)w * 4] + c) * 256 + l;
            v2d x;
            x.x = a[0]; x.y = a[1]; ST_ROWS_STORE(x, p);
            x.x = a[2]; x.y = a[3]; ST_ROWS_STORE(x, p + 64);
            x.x = tj[0]; x.y = tj[1]; ST_ROWS_STORE(x, p + 128);
            x.x = tj[2]; x.y = a[6]; ST_ROWS_STORE(x, p + 192);
            x.x = a[4]; x.y = a[5]; ST_ROWS_STORE(x, pkU + ((size_t)wtab[(size_t)w * 4] + c) * 128 + l);
        }
    }
    if (bad) atomicOr(&flags[0], 8);
}

#ifndef ST_STAMP_WG
#define ST_STAMP_WG -1
#endif
#ifndef ST_STAMP_WV
#define ST_STAMP_WV 0
#endif
#ifdef ST_STAMP
// diagnostics build only: cycles of the first wave of the LAST workgroup, summed over its steps, per segment of a step
__device__ unsigned long long g_st_stamp[32];
__device__ unsigned long long g_st_tl[4096 * 4];     // per workgroup of the forward sweep: entry, first row done, last row done, exit (100 MHz)
#define ST_T(i) do { if (stamp_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_[i] += n_ - last_; last_ = __builtin_amdgcn_s_memtime(); } } while (0)
#define ST_T_DECL(cond) const bool stamp_on = (cond); unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long last_ = __builtin_amdgcn_s_memtime(); unsigned long long nst_ = 0
#define ST_T_END(off) do { if (stamp_on && ln == 0) { for (int i_ = 0; i_ < 8; ++i_) g_st_stamp[(off) + i_] = acc_[i_]; g_st_stamp[(off) + 8] = nst_; } } while (0)
#else
#define ST_T(i) do { } while (0)
#define ST_T_DECL(cond) do { } while (0)
#define ST_T_END(off) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// the sweeps.  DR = +1 forward (rows ascending; the diagonal of L is 1: no division), -1 backward.
// Vectors travel level-major too: a lane reading / writing its own natural-order element is one cache line per lane and
// instruction -- 64 lines per wave instruction -- and that alone was half of a forward step and a third of a backward one
// (single tile, 400 steps: 204 us with the natural-order right-hand side, 101 without; backward 195 -> 124 without the
// natural-order store).  So k_st_vec brings the right-hand side into the forward sweep's order first, the forward sweep
// stores the intermediate vector in the BACKWARD sweep's order (each lane knows where its rows sit there), the backward sweep
// reads and overwrites it in place, and k_st_vec takes the result back to natural order.  Natural-order stores are left
// for the lanes other workgroups read (write-through, the data is the flag).
// ---------------------------------------------------------------------------------------------
// Who does what in a workgroup of a sweep: waves 0-3 are the 256 lanes of the schedule, wave 4 is the COURIER.  Unknowns of
// earlier workgroups ("ghosts") used to be polled, tested and selected by the lanes that need them; that code was 150 of the 200
// instructions of a step in a wave that has such lanes (a wave64 instruction is 4 cycles: 0.2 us per step, and with the barrier
// the whole workgroup runs at the pace of that wave) -- a workgroup behind a border ran at half the speed of one without, whatever
// the poll distance or the kind of load.  Now the courier polls (kStPF / kStPS steps ahead, one (lane, dependency) pair per courier lane,
// at most 64 per workgroup -- the analysis checks), waits for values not there yet, and puts them into the hand-off array as the
// values of 64 more lanes "of this step"; to the lanes of the schedule a ghost is just another LDS read at a constant address.
// EX: some lane's unknowns are read by later workgroups (a template parameter, not a branch: hipcc's waitcnt pass only counts the
// memory operations it is sure were issued, so a store inside a branch makes every later wait of a wave that does execute it
// stricter than meant by one)
template <int DR, bool EX, bool TR>
__device__ __forceinline__ void st_sweep_wave(const StSArgs &A, unsigned char *xh, const int wg, const unsigned *va, const int tlo, const int thi)
{
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], sk = T[ST_SKEW];
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    const int xe = A.xe[slot];
    const bool exports = cnt > 0 && xe >= 0;
    const int xE = __builtin_amdgcn_readfirstlane(A.xw[wg * 4]);
    const int xoff = A.xw[wg * 4 + 3] - A.xw[wg * 4 + 1] * xE + xe;          // + step * xE: where this lane's value of a step goes
    const unsigned char *pr = reinterpret_cast<const unsigned char *>(A.pk) + (DR > 0 ? (size_t)(nchw > 0 ? base : 0) * 2048 : (size_t)0);
    // forward: the right-hand side from xlm, chunk by chunk, and the intermediate vector back into the same place (a store
    // in another order -- the backward sweep's -- would be 64 partial writes per instruction: lanes in descending address
    // order do not merge; it cost the forward sweep 130 us on 256^3); backward: from xlm lane by lane (a load does not care)
    const unsigned char *prr = reinterpret_cast<const unsigned char *>(A.xlm) + (DR > 0 ? (size_t)(nchw > 0 ? base : 0) * 512 : (size_t)0);
    unsigned char *py = reinterpret_cast<unsigned char *>(DR > 0 ? A.xlm : A.ylm);
    // (the backward sweep's records too are stored in the forward order: the factor kernel writes them)
    const int ysrc_ = DR > 0 ? 0 : A.ysrc[slot];
    const int ychunk0 = (ysrc_ >> 6) + sk;                               // chunk of step 0 (step s: - s)
    const unsigned ylane16 = (unsigned)(ysrc_ & 63) * 16u;
    const unsigned lo16 = (unsigned)ln * 16u, lo8 = (unsigned)ln * 8u;
    const int cmax = nchw > 0 ? nchw - 1 : 0;
    const int ydump = A.nchY + wg * 4 + wv;                             // spare chunk of this wave (a shared one would be a hot spot)
    v2d ra[kStRA][2];
    double rr[kStRA];

#define STS_LOAD(u, tp)                                                                                              \
    do {                                                                                                             \
        if (DR > 0) {                                                                                                \
            const int cw_ = st_med3((tp) - tminw, 0, cmax);                                                          \
            const unsigned char *q_ = pr + (size_t)cw_ * 2048;                                                       \
            ra[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + lo16));                         \
            ra[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + lo16 + 1024));                  \
            rr[u] = __builtin_nontemporal_load(reinterpret_cast<const double *>(prr + (size_t)cw_ * 512 + lo8));     \
        } else {                                                                                                     \
            const unsigned ch_ = (unsigned)st_med3(ychunk0 - (tp), 0, A.xlm_chunks - 1);                             \
            const unsigned char *q_ = pr + ch_ * 2048u + ylane16;                                                    \
            ra[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_));                                \
            ra[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + 1024));                         \
            rr[u] = __builtin_nontemporal_load(reinterpret_cast<const double *>(prr + ch_ * 512u + (ylane16 >> 1))); \
        }                                                                                                            \
    } while (0)

#pragma unroll
    for (int u = 0; u < kStRA; ++u) {
        STS_LOAD(u, tlo + u);
        // (the stores of a step, to the dump places: the waitcnt pass takes the minimum over the way into the loop and its back
        // edge, so the way in has to look like a pass of the loop -- see k_ilu0_st)
        ST_STREAM_STORE(0.0, reinterpret_cast<double *>(py + (size_t)ydump * 512 + lo8));
        if (EX) st_agent_f64(reinterpret_cast<double *>(py + (size_t)ydump * 512 + lo8), 0.0);
        asm volatile("" ::: "memory");
    }
    ST_T_DECL(wv == ST_STAMP_WV && wg == (ST_STAMP_WG < 0 ? (int)gridDim.x - 1 : ST_STAMP_WG));
#ifdef ST_STAMP
    if (DR > 0 && t == 0 && wg < 4096) g_st_tl[wg * 4] = __builtin_amdgcn_s_memrealtime();
#endif

    for (int tb = tlo; tb < thi; tb += kStRA) {
        const int kb = tb - sk;
#pragma unroll
        for (int u = 0; u < kStRA; ++u) {
            const int k = kb + u;
            const bool valid = (unsigned)k < (unsigned)cnt;
#ifdef ST_STAMP
            if (DR > 0 && t == 0 && wg < 4096 && k == 1) g_st_tl[wg * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            if (DR > 0 && t == 255 && wg < 4096 && k == cnt - 1) g_st_tl[wg * 4 + 2] = __builtin_amdgcn_s_memrealtime();
            ++nst_;
#endif
            ST_T(0);
            const v2d va_ = ra[u][0], vb = ra[u][1];
            const double v[3] = {va_.x, va_.y, vb.x};
            bool pj[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) pj[j] = valid && st_bits(v[j]) != kAbsent;
            // (the forward sweep does not divide: the diagonal is used HERE so that its register stays taken until this
            // step -- a dead quarter of a 16-byte load is a free register to the allocator, and whatever it puts there has
            // to wait for that load, a load of a later step)
            if (DR > 0 && !TR) pj[2] = pj[2] && st_bits(vb.y) != kAbsent;
            ST_BARRIER();
            ST_T(1);
            double xs[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) xs[j] = st_lds(xh, va[j] + (unsigned)(u % kStH) * (kStRow * 8));
#ifdef ST_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            ST_T(2);
            ST_T(3);
            // sequential accumulation in stored order; the division by a diagonal of 1.0 would return the dividend
            // (transposed factors: U^T forward with its diagonal; L^T backward in DESCENDING column order -- the reference walks
            // the columns of a row-stored L from the last to the first, sparse_implementation.h:4040-4087)
            double acc = rr[u];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int j = (TR && DR < 0) ? 2 - jj : jj;
                const double p = v[j] * xs[j];
                const double na = acc - p;
                acc = pj[j] ? na : acc;
            }
            double x = (DR > 0 && !TR) ? acc : acc / vb.y;
            if (x != x) x = st_dbl(kCanonNaN);
#ifdef ST_STAMP
            asm volatile("" :: "v"(x));
#endif
            ST_T(4);
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)(u % kStH) * (kStRow * 8)) = x;
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)(u % kStH + kStH) * (kStRow * 8)) = x;
            // the stream store happens on every step (lanes / waves without a row store to the wave's spare chunk); unknowns
            // that other workgroups read: also to the natural-order vector, write-through
            {
                const int cw = tb + u - tminw;
                unsigned char *o = py + (size_t)((unsigned)cw < (unsigned)nchw ? base + cw : ydump) * 512;
                ST_STREAM_STORE(x, reinterpret_cast<double *>(o + lo8));
            }
            if (EX) st_agent_f64((exports && valid) ? A.xch + (xoff + (tb + u) * xE) : reinterpret_cast<double *>(py + (size_t)ydump * 512 + lo8), x);
            ST_T(5);
            STS_LOAD(u, tb + u + kStRA);
            ST_T(6);
        }
    }
#undef STS_LOAD
    ST_T_END(DR > 0 ? 10 : 20);
#ifdef ST_STAMP
    if (DR > 0 && t == 0 && wg < 4096) g_st_tl[wg * 4 + 3] = __builtin_amdgcn_s_memrealtime();
#endif
}

// the courier of a sweep: lane p serves pair p.  src: the exchange the producers write through to (all-sentinel before the
// sweep); a pair's value for step s is element idx0 + s stride, wanted for klo <= s - sk < khi.
template <int RA, int NP>
__device__ __forceinline__ void st_courier(const unsigned long long *src, const unsigned long long *idle, unsigned char *xh, const StPair P,
                                                 const int tlo, const int thi, int32_t *err)
{
    const int ln = threadIdx.x & 63;
    const unsigned span = (unsigned)(P.khi - P.klo);
    unsigned long long gq[NP];
#define STC_ADDR(k_) ((unsigned)((k_) - P.klo) < span ? src + (P.idx0 + ((k_) + P.sk) * P.stride) : idle)
#pragma unroll
    for (int g = 0; g < NP; ++g) { gq[g] = ld_agent_u64(STC_ADDR(tlo + g - P.sk)); asm volatile("" ::: "memory"); }
    bool dead = false;
    for (int tb = tlo; tb < thi; tb += RA) {
#pragma unroll
        for (int u = 0; u < RA; ++u) {
            const int k = tb + u - P.sk;
            const bool need = (unsigned)(k - P.klo) < span;
            unsigned long long v = gq[u % NP];
            if (!dead) {
                unsigned spins = 0;
                while (__builtin_amdgcn_ballot_w64(need && v == kSentinel) != 0) {
                    if (need && v == kSentinel) v = ld_agent_u64(STC_ADDR(k));
                    __builtin_amdgcn_s_waitcnt(0x0F70);          // retired here, not at the join after the loop
                    __builtin_amdgcn_s_sleep(ST_CSLEEP);
                    if ((++spins & 255u) == 0) {
                        if (spins > kStSpinLimit) atomicExch(err, 1);
                        const int e = ld_agent_i32(err);
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        if (spins > kStSpinLimit || e != 0) { dead = true; break; }
                    }
                }
            }
            *reinterpret_cast<unsigned long long *>(xh + (unsigned)(kThreads + ln) * 8 + (unsigned)(u % kStH + kStH) * (kStRow * 8)) = v;
            gq[u % NP] = ld_agent_u64(STC_ADDR(k + NP));
            ST_BARRIER();
        }
    }
#undef STC_ADDR
    if (dead && ln == 0) atomicExch(err, 1);
}

template <int DR, bool TR>
__global__ void __launch_bounds__(kStWgThreads)
k_sptrsv_st(StSArgs A)
{
    __shared__ __attribute__((aligned(16))) unsigned char xh[2 * kStH * kStRow * 8];
    __shared__ StPair s_pairs[64];
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
        const int cnt = T[ST_CNT];
        bool isg[3]; int idx0[3], stride[3]; unsigned va[3];
        st_lane_sources(T, t, A.ltab, A.xe, A.xw, isg, idx0, stride, va);
        __syncthreads();                                              // (s_pairs zeroed)
        st_number_pairs(T, t, isg, idx0, stride, va, s_pairs, s_cnt, &s_total);
        __syncthreads();
        if (t == 0 && s_total > 64) atomicExch(A.err, 1);             // (the analysis does not let such a schedule through)
        const bool wave_exports = __any(cnt > 0 && A.xe[slot] >= 0);
        // (every wave passes the same number of barriers whichever variant it runs)
        if (wave_exports) st_sweep_wave<DR, true, TR>(A, xh, wg, va, tlo, thi); else st_sweep_wave<DR, false, TR>(A, xh, wg, va, tlo, thi);
    } else {
        __syncthreads();
        __syncthreads();                                              // (the one inside st_number_pairs)
        __syncthreads();
        const StPair P = s_pairs[t - kThreads];
        // (a poll nobody needs goes to a place of this workgroup's own: the same address for the whole chip would be a hot spot)
        const unsigned long long *idle = reinterpret_cast<const unsigned long long *>(A.ltab + (size_t)wg * kThreads * kStTab);
        st_courier<kStRA, kStPS>(reinterpret_cast<const unsigned long long *>(A.xch), idle, xh, P, tlo, thi, A.err);
    }
}

// ---------------------------------------------------------------------------------------------
// the factor kernel: pivots only
// ---------------------------------------------------------------------------------------------
struct StFArgs {
    const int32_t *ltab, *wtab;               // forward schedule
    const v2d *pkA;                           // 4 x 64 x 16 B per chunk: {a0,a1}{a2,a3}{t0,t1}{t2,a6}
    v2d *pkL, *pkU;                           // 2 x 64 x 16 B per chunk: {l0,l1}{l2,1} / {u1,u2}{u3,u0}, BOTH in the forward schedule's order
    int32_t nchL;                             // index of the first spare chunk (one per wave)
    const int32_t *xe, *xw;                   // the forward schedule's exchange between workgroups (PackedSweep::xe, xw, xch),
    double *xch;                              // all-sentinel before the kernel: pivots of exported lanes
    int32_t *ctrl;                            // [0] ticket, [1] error
};

// the 256 lanes of the schedule (the courier wave brings the pivots of earlier workgroups: see the sweeps)
template <bool EX>
__device__ __forceinline__ void st_factor_wave(const StFArgs &A, unsigned char *xh, const int wg, const unsigned *va, const int tlo, const int thi)
{
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], sk = T[ST_SKEW];
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    const int xe = A.xe[slot];
    const bool exports = cnt > 0 && xe >= 0;
    const int xE = __builtin_amdgcn_readfirstlane(A.xw[wg * 4]);
    const int xoff = A.xw[wg * 4 + 3] - A.xw[wg * 4 + 1] * xE + xe;          // + step * xE: where this lane's pivot of a step goes
    const unsigned char *pa = reinterpret_cast<const unsigned char *>(A.pkA) + (size_t)(nchw > 0 ? base : 0) * 4096;
    unsigned char *pl = reinterpret_cast<unsigned char *>(A.pkL);
    unsigned char *pu = reinterpret_cast<unsigned char *>(A.pkU);
    const unsigned lo16 = (unsigned)ln * 16u;
    const int cmax = nchw > 0 ? nchw - 1 : 0;
    // dump places: one spare chunk per wave behind the records (a place shared by all waves would be a hot spot)
    const int wglob = wg * 4 + wv;
    const unsigned udump = (unsigned)(A.nchL + wglob) * 2048u + lo16;
    const int ldump = A.nchL + wglob;

    v2d ra[kStH][4];

#define STF_LOAD(u, tp)                                                                                              \
    do {                                                                                                             \
        const int cw_ = st_med3((tp) - tminw, 0, cmax);                                                              \
        const unsigned char *q_ = pa + (size_t)cw_ * 4096;                                                           \
        ra[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + lo16));                             \
        ra[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + lo16 + 1024));                      \
        ra[u][2] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + lo16 + 2048));                      \
        ra[u][3] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + lo16 + 3072));                      \
    } while (0)

    // the way into the loop in step order (a compiler barrier after each) and with the stores a step of the loop has, to the
    // dump places: the waits of the loop are derived from the oldest position a register's load can have, hipcc takes the
    // smaller of the operation counts behind the load on the two ways into the loop body, and the scheduler is free to turn an
    // unordered prologue upside down
    const double absent = st_dbl(kAbsent);
#pragma unroll
    for (int u = 0; u < kStH; ++u) {
        STF_LOAD(u, tlo + u);
        v2d z; z.x = absent; z.y = absent;
        ST_STREAM_STORE(z, reinterpret_cast<v2d *>(pl + (size_t)ldump * 2048 + lo16));
        ST_STREAM_STORE(z, reinterpret_cast<v2d *>(pl + (size_t)ldump * 2048 + lo16 + 1024));
        ST_STREAM_STORE(z, reinterpret_cast<v2d *>(pu + udump));
        if (EX) st_agent_f64(reinterpret_cast<double *>(pu + udump), absent);
        asm volatile("" ::: "memory");
    }
    ST_T_DECL(wv == ST_STAMP_WV && wg == (ST_STAMP_WG < 0 ? (int)gridDim.x - 1 : ST_STAMP_WG));

    for (int tb = tlo; tb < thi; tb += kStH) {
        const int kb = tb - sk;
#pragma unroll
        for (int u = 0; u < kStH; ++u) {
            const int k = kb + u;
            const bool valid = (unsigned)k < (unsigned)cnt;
#ifdef ST_STAMP
            ++nst_;
#endif
            ST_T(0);
            const v2d r0 = ra[u][0], r1 = ra[u][1], r2 = ra[u][2], r3 = ra[u][3];
            const double av[3] = {r0.x, r0.y, r1.x}, at[3] = {r2.x, r2.y, r3.x};
            bool pj[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) pj[j] = valid && st_bits(av[j]) != kAbsent;
            ST_BARRIER();
            ST_T(1);
            double piv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) piv[j] = st_lds(xh, va[j] + (unsigned)u * (kStRow * 8));
#ifdef ST_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            ST_T(2);
            ST_T(3);
            // u_ii = a_ii - sum (a_ik / u_kk) a_ki, eliminations in ascending k (ILU0.hpp:47-62 for rows whose eliminations
            // meet them on the diagonal only)
            double w3 = r1.y;
            double l[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                l[j] = av[j] / piv[j];
                const double pr = l[j] * at[j];
                const double nw = w3 - pr;
                w3 = (pj[j] && st_bits(at[j]) != kAbsent) ? nw : w3;
            }
#ifdef ST_STAMP
            asm volatile("" :: "v"(w3));
#endif
            ST_T(4);
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)u * (kStRow * 8)) = w3;
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)(u + kStH) * (kStRow * 8)) = w3;
            // pivots that other workgroups read: write-through, to the exchange
            if (EX) st_agent_f64((exports && valid) ? A.xch + (xoff + (tb + u) * xE) : reinterpret_cast<double *>(pu + udump), w3);
            // the record stores happen on every step (lanes / waves without a row store to their dump place)
            const int cw = tb + u - tminw;
            {
                unsigned char *o = pl + (size_t)((unsigned)cw < (unsigned)nchw ? base + cw : ldump) * 2048;
                v2d la, lb;
                la.x = pj[0] ? l[0] : absent; la.y = pj[1] ? l[1] : absent;
                lb.x = pj[2] ? l[2] : absent; lb.y = 1.0;
                ST_STREAM_STORE(la, reinterpret_cast<v2d *>(o + lo16));
                ST_STREAM_STORE(lb, reinterpret_cast<v2d *>(o + lo16 + 1024));
            }
            {
                v2d ub; ub.x = r3.y; ub.y = w3;
                ST_STREAM_STORE(ub, reinterpret_cast<v2d *>(pu + (size_t)((unsigned)cw < (unsigned)nchw ? base + cw : ldump) * 2048 + 1024 + lo16));
            }
            ST_T(5);
            STF_LOAD(u, tb + u + kStH);
            ST_T(6);
        }
    }
#undef STF_LOAD
    ST_T_END(0);
}

template <bool EX>
__device__ __forceinline__ void st_chol_wave(const StFArgs &A, unsigned char *xh, const int wg, const unsigned *va, const int tlo, const int thi);

template <bool CHOL>
__device__ __forceinline__ void st_factor_body(const StFArgs &A)
{
    __shared__ __attribute__((aligned(16))) unsigned char xh[2 * kStH * kStRow * 8];      // pivots of finished rows: [16][320], slots s and s+8 alike
    __shared__ StPair s_pairs[64];
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
    tlo &= ~(kStH - 1);                                                 // step % 8 = position in the unrolled loop
    if (t < 64) { StPair z; z.idx0 = 0; z.stride = 0; z.sk = 0; z.klo = 0; z.khi = 0; s_pairs[t] = z; }
    if (t < kThreads) {
        const int slot = wg * kThreads + t;
        const int32_t *T = A.ltab + (size_t)slot * kStTab;
        bool isg[3]; int idx0[3], stride[3]; unsigned va[3];
        st_lane_sources(T, t, A.ltab, A.xe, A.xw, isg, idx0, stride, va);
        __syncthreads();                                              // (s_pairs zeroed)
        st_number_pairs(T, t, isg, idx0, stride, va, s_pairs, s_cnt, &s_total);
        __syncthreads();
        if (t == 0 && s_total > 64) atomicExch(&A.ctrl[1], 1);        // (the analysis does not let such a schedule through)
        const bool wave_exports = __any(T[ST_CNT] > 0 && A.xe[slot] >= 0);
        if (CHOL) { if (wave_exports) st_chol_wave<true>(A, xh, wg, va, tlo, thi); else st_chol_wave<false>(A, xh, wg, va, tlo, thi); }
        else      { if (wave_exports) st_factor_wave<true>(A, xh, wg, va, tlo, thi); else st_factor_wave<false>(A, xh, wg, va, tlo, thi); }
    } else {
        __syncthreads();
        __syncthreads();                                              // (the one inside st_number_pairs)
        __syncthreads();
        const StPair P = s_pairs[t - kThreads];
        const unsigned long long *idle = reinterpret_cast<const unsigned long long *>(A.ltab + (size_t)wg * kThreads * kStTab);
        st_courier<kStH, kStPF>(reinterpret_cast<const unsigned long long *>(A.xch), idle, xh, P, tlo, thi, &A.ctrl[1]);
    }
}

__global__ void __launch_bounds__(kStWgThreads)
k_ilu0_st(StFArgs A)
{
    st_factor_body<false>(A);
}

// ---------------------------------------------------------------------------------------------
// IChol(0) on the static form (IChol.hpp:33-59 for rows whose dot products with earlier rows are all empty: the "simple" rows of a
// 5-/7-point stencil's lower triangle, proven lane by lane in k_st_chol_check): l_ij = a_ij / l_jj, l_ii = sqrt(a_ii - sum_j l_ij^2),
// the sum in ascending column order starting from 0.0 as sparse_dot_product does (IChol.hpp:16-28).  Same machinery as the ILU(0)
// kernel -- lane tables, one barrier per step, hand-off array in LDS (here the finished diagonal l_ii), courier wave, exchange --
// on records {a0,a1}{a2,a_ii} that are overwritten in place by {l0,l1}{l2,l_ii} (a chunk is read eight steps before it is written).
// ---------------------------------------------------------------------------------------------
template <bool EX>
__device__ __forceinline__ void st_chol_wave(const StFArgs &A, unsigned char *xh, const int wg, const unsigned *va, const int tlo, const int thi)
{
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], sk = T[ST_SKEW];
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    const int xe = A.xe[slot];
    const bool exports = cnt > 0 && xe >= 0;
    const int xE = __builtin_amdgcn_readfirstlane(A.xw[wg * 4]);
    const int xoff = A.xw[wg * 4 + 3] - A.xw[wg * 4 + 1] * xE + xe;
    unsigned char *pk = reinterpret_cast<unsigned char *>(A.pkL);
    const unsigned char *pa = pk + (size_t)(nchw > 0 ? base : 0) * 2048;
    const unsigned lo16 = (unsigned)ln * 16u;
    const int cmax = nchw > 0 ? nchw - 1 : 0;
    const int ldump = A.nchL + wg * 4 + wv;                       // one spare chunk per wave behind the records
    double *xdump = reinterpret_cast<double *>(pk + (size_t)ldump * 2048 + lo16);

    v2d ra[kStH][2];
#define STC_LOAD(u, tp)                                                                                              \
    do {                                                                                                             \
        const int cw_ = st_med3((tp) - tminw, 0, cmax);                                                              \
        const unsigned char *q_ = pa + (size_t)cw_ * 2048;                                                           \
        ra[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + lo16));                             \
        ra[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(q_ + lo16 + 1024));                      \
    } while (0)
    const double absent = st_dbl(kAbsent);
    // (the way into the loop in step order with the stores a step has: see st_factor_wave)
#pragma unroll
    for (int u = 0; u < kStH; ++u) {
        STC_LOAD(u, tlo + u);
        v2d z; z.x = absent; z.y = absent;
        ST_STREAM_STORE(z, reinterpret_cast<v2d *>(pk + (size_t)ldump * 2048 + lo16));
        ST_STREAM_STORE(z, reinterpret_cast<v2d *>(pk + (size_t)ldump * 2048 + lo16 + 1024));
        if (EX) st_agent_f64(xdump, absent);
        asm volatile("" ::: "memory");
    }
    for (int tb = tlo; tb < thi; tb += kStH) {
        const int kb = tb - sk;
#pragma unroll
        for (int u = 0; u < kStH; ++u) {
            const int k = kb + u;
            const bool valid = (unsigned)k < (unsigned)cnt;
            const v2d r0 = ra[u][0], r1 = ra[u][1];
            const double av[3] = {r0.x, r0.y, r1.x};
            bool pj[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) pj[j] = valid && st_bits(av[j]) != kAbsent;
            ST_BARRIER();
            double piv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) piv[j] = st_lds(xh, va[j] + (unsigned)u * (kStRow * 8));
            double l[3];
            double z = 0.0;                                               // sparse_dot_product of the row with itself (IChol.hpp:16-28)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                l[j] = av[j] / piv[j];                                    // (A_ij - 0.0) / L_jj  (IChol.hpp:49-51)
                const double sq = l[j] * l[j];
                const double nz = z + sq;
                z = pj[j] ? nz : z;
            }
            const double d = sqrt(r1.y - z);                              // IChol.hpp:52-53
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)u * (kStRow * 8)) = d;
            *reinterpret_cast<double *>(xh + (unsigned)t * 8 + (unsigned)(u + kStH) * (kStRow * 8)) = d;
            if (EX) st_agent_f64((exports && valid) ? A.xch + (xoff + (tb + u) * xE) : xdump, d);
            const int cw = tb + u - tminw;
            {
                unsigned char *o = pk + (size_t)((unsigned)cw < (unsigned)nchw ? base + cw : ldump) * 2048;
                v2d la, lb;
                la.x = pj[0] ? l[0] : absent; la.y = pj[1] ? l[1] : absent;
                lb.x = pj[2] ? l[2] : absent; lb.y = d;
                ST_STREAM_STORE(la, reinterpret_cast<v2d *>(o + lo16));
                ST_STREAM_STORE(lb, reinterpret_cast<v2d *>(o + lo16 + 1024));
            }
            STC_LOAD(u, tb + u + kStH);
        }
    }
#undef STC_LOAD
}

__global__ void __launch_bounds__(kStWgThreads)
k_ichol0_st(StFArgs A)
{
    st_factor_body<true>(A);
}

// the proof: no row of a lane shares a column left of the diagonal with a row it divides by (then every dot product of
// IChol.hpp:47 with an earlier row is empty).  Offsets of the lane and of each of its dependencies' lanes, from the lane tables.
__global__ void __launch_bounds__(kThreads)
k_st_chol_check(const int32_t *__restrict__ ltab, int32_t *__restrict__ flags)
{
    const int slot = blockIdx.x * kThreads + threadIdx.x;
    const int32_t *T = ltab + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT], nd = T[ST_ND];
    if (cnt <= 0) return;
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nd) {
            const int32_t *TP = ltab + (size_t)(T[ST_SRC + j] >> 2) * kStTab;
            const int np = TP[ST_ND];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (p < np) {
                    const int m = T[ST_OFF + j] + TP[ST_OFF + p];        // a column of the dependency's row, relative to this row
#pragma unroll
                    for (int q = 0; q < 3; ++q) if (q < nd && m == T[ST_OFF + q]) bad = 1;
                }
            }
        }
    }
    if (bad) atomicOr(&flags[0], 4);
}

// lower rows (diagonal last) -> records {a0,a1}{a2,a_ii}, every entry matched against its lane's template and range of rows.  The CSR
// side is read as in k_st_unpack it is written: a lane's eight rows of a group of eight chunks are one contiguous run, fetched by eight
// threads per lane into LDS (256^3: 1.35 ms with a row per lane and instruction)
__global__ void __launch_bounds__(512)
k_st_pack_lower(const int32_t *__restrict__ Lptr, const int32_t *__restrict__ Lidx, const double *__restrict__ Lval,
                const int32_t *__restrict__ ltab, const int32_t *__restrict__ wtab, v2d *__restrict__ pk, int32_t *__restrict__ flags)
{
    __shared__ double s_val[64][33];
    __shared__ int s_idx[64][33];
    __shared__ int s_lo[64], s_hi[64];
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    if (threadIdx.x < 64) { s_lo[threadIdx.x] = 0x7fffffff; s_hi[threadIdx.x] = -1; }
    __syncthreads();
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const int32_t *T = ltab + (size_t)slot * kStTab;
    const int k = tmin + c - T[ST_SKEW];
    const int cnt = T[ST_CNT];
    const bool inchunk = c < nch;
    const bool live = inchunk && k >= 0 && k < cnt;
    const int r = live ? T[ST_FIRST] + k : 0;
    const int q0 = live ? Lptr[r] : 0, q1 = live ? Lptr[r + 1] : 0;
    const bool shape_ok = q1 - q0 >= 1 && q1 - q0 <= 4;
    if (live && shape_ok) { atomicMin(&s_lo[L], q0); atomicMax(&s_hi[L], q1); }
    __syncthreads();
    {
        const int L2 = threadIdx.x >> 3, j = threadIdx.x & 7;
        const int lo2 = s_lo[L2], len = s_hi[L2] - lo2;
        if (len > 0 && len <= 32) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * j + e < len) { s_val[L2][4 * j + e] = Lval[lo2 + 4 * j + e]; s_idx[L2][4 * j + e] = Lidx[lo2 + 4 * j + e]; }
        }
    }
    __syncthreads();
    if (!inchunk) return;
    const double absent = st_dbl(kAbsent);
    v2d *o = pk + ((size_t)base + c) * 128 + L;
    v2d x;
    if (!live) { x.x = absent; x.y = absent; o[0] = x; o[64] = x; return; }
    const int lo = s_lo[L];
    const bool staged = shape_ok && s_hi[L] - lo <= 32;
    double lv[3] = {absent, absent, absent};
    int bad = 0;
    const int dcol = !shape_ok ? -1 : (staged ? s_idx[L][q1 - 1 - lo] : Lidx[q1 - 1]);
    if (!shape_ok || dcol != r) bad = 1;
    for (int q = q0; q < q1 - 1 && !bad; ++q) {
        const int col = staged ? s_idx[L][q - lo] : Lidx[q];
        const int off = col - r;
        int hit = -1;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (j < T[ST_ND] && T[ST_OFF + j] == off) hit = j;
        const double vq = staged ? s_val[L][q - lo] : Lval[q];
        if (hit < 0 || k < T[ST_KLO + hit] || k >= T[ST_KHI + hit]) bad = 1;
        else { const double cv = st_clean(vq); if (hit == 0) lv[0] = cv; else if (hit == 1) lv[1] = cv; else lv[2] = cv; }
    }
    if (bad) { atomicOr(&flags[0], 8); return; }
    x.x = lv[0]; x.y = lv[1]; o[0] = x;
    x.x = lv[2]; x.y = st_clean(staged ? s_val[L][q1 - 1 - lo] : Lval[q1 - 1]); o[64] = x;
}

// natural order <-> level-major order of a sweep (64 per chunk), both sides coalesced through an LDS tile of 32 chunks x 64 lanes:
// the natural-order side is touched as one contiguous run of 32 elements per lane (8 threads x 32 B), the level-major side as whole
// chunks (a wave per chunk: 512 contiguous bytes).  Which element of a lane sits in a chunk depends on the lane's skew:
// k = chunk + tmin - skew; DR = +1: the lane's rows ascend with k (forward schedule), -1: they descend (backward schedule).
template <int DR, bool TO_LM>
__global__ void __launch_bounds__(512)
k_st_vec(const int32_t *__restrict__ ltab, const int32_t *__restrict__ wtab, double *__restrict__ nat, double *__restrict__ lm)
{
    __shared__ double tile[32][65];
    const int w = blockIdx.x;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    const int t = threadIdx.x;
    // natural-order role: lane l, elements j = sub*4 .. sub*4+3 of its run (chunk c0 + j)
    const int l = t >> 3, sub = t & 7;
    const v4i t0 = *reinterpret_cast<const v4i *>(ltab + (size_t)((w >> 2) * kThreads + (w & 3) * 64 + l) * kStTab);      // first, cnt, skew, nd
    // level-major role: chunk c0 + (t >> 6) + 8 i, lane t & 63
    const int ll = t & 63, cw = t >> 6;
    // a block takes every gridDim.y-th group of 32 chunks of its wave; the loads of the next group are in flight while this one
    // is stored (a block per group was 9 000 short-lived blocks, each a chain of table -> loads -> LDS -> stores: 2.7 TB/s)
    double v[4];
#define STV_LOAD(c0_)                                                                                   \
    do {                                                                                                \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
            if (TO_LM) {                                                                                \
                const int j = sub * 4 + q;                                                              \
                const int k = (c0_) + j + tmin - t0.z;                                                  \
                v[q] = (k >= 0 && k < t0.y) ? nat[t0.x + DR * k] : 0.0;                                 \
            } else {                                                                                    \
                const int j = cw + 8 * q;                                                               \
                v[q] = ((c0_) + j < nch) ? lm[((size_t)base + (c0_) + j) * 64 + ll] : 0.0;              \
            }                                                                                           \
        }                                                                                               \
    } while (0)
    int c0 = blockIdx.y * 32;
    if (c0 >= nch) return;
    STV_LOAD(c0);
    for (; c0 < nch; c0 += gridDim.y * 32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (TO_LM) tile[sub * 4 + q][l] = v[q]; else tile[cw + 8 * q][ll] = v[q];
        }
        __syncthreads();
        const int cn = c0 + gridDim.y * 32;
        double o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = TO_LM ? tile[cw + 8 * q][ll] : tile[sub * 4 + q][l];
        if (cn < nch) STV_LOAD(cn);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (TO_LM) {
                const int j = cw + 8 * q;
                if (c0 + j < nch) lm[((size_t)base + c0 + j) * 64 + ll] = o[q];
            } else {
                const int j = sub * 4 + q;
                const int k = c0 + j + tmin - t0.z;
                if (k >= 0 && k < t0.y && c0 + j < nch) nat[t0.x + DR * k] = o[q];
            }
        }
        __syncthreads();
    }
#undef STV_LOAD
}

// CSR values AND column indices of a factor from its records.  The eight rows a lane contributes to a group of eight chunks are
// consecutive rows, i.e. ONE contiguous run of at most 32 entries of val / idx: a thread per (chunk, lane) puts its row's entries into the
// lane's run in LDS (the records side: 64 lanes x 16 bytes per instruction), then eight threads per lane write the run out in
// 32-byte pieces -- an instruction covers 8 lanes' runs instead of 64 rows that lie 14 KB apart (256^3, IChol0's L: 1.33 ms before).
template <int KIND>
__global__ void __launch_bounds__(512)
k_st_unpack(const int32_t *__restrict__ ptr, int32_t *__restrict__ idx, double *__restrict__ val, const int32_t *__restrict__ wtab,
            const int32_t *__restrict__ ltab, const v2d *__restrict__ pk, const int32_t *__restrict__ ysrc)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr int DR = FWD ? 1 : -1;
    __shared__ double s_val[64][33];
    __shared__ int s_idx[64][33];
    __shared__ int s_lo[64], s_hi[64];
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    if (threadIdx.x < 64) { s_lo[threadIdx.x] = 0x7fffffff; s_hi[threadIdx.x] = -1; }
    __syncthreads();
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const int32_t *T = ltab + (size_t)slot * kStTab;
    const int k = tmin + c - T[ST_SKEW];
    const bool live = c < nch && k >= 0 && k < T[ST_CNT];
    const int r = live ? T[ST_FIRST] + DR * k : 0;
    const int q0 = live ? ptr[r] : 0, q1 = live ? ptr[r + 1] : 0;
    if (live) { atomicMin(&s_lo[L], q0); atomicMax(&s_hi[L], q1); }
    __syncthreads();
    const int lo = s_lo[L];
    const bool staged = s_hi[L] - lo <= 32;                              // (rows as long as the static form has them: always)
    if (live) {
        // (the backward sweep's records are stored in the forward schedule's order: ysrc = chunk * 64 + lane of the lane's row 0)
        const int pos = FWD ? 0 : ysrc[slot] - 64 * k;
        const v2d *p = FWD ? pk + ((size_t)base + c) * 128 + L : pk + (size_t)(pos >> 6) * 128 + (pos & 63);
        const v2d a = p[0], b = p[64];
        const double v[3] = {a.x, a.y, b.x};
        const int qd = FWD ? q1 - 1 : q0;                               // the diagonal: last (forward) / first (backward) of the row
        if (staged) { s_val[L][qd - lo] = b.y; s_idx[L][qd - lo] = r; } else { val[qd] = b.y; idx[qd] = r; }
        int q = FWD ? q0 : q0 + 1;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (st_bits(v[j]) != kAbsent && q < (FWD ? q1 - 1 : q1)) {
                if (staged) { s_val[L][q - lo] = v[j]; s_idx[L][q - lo] = r + T[ST_OFF + j]; } else { val[q] = v[j]; idx[q] = r + T[ST_OFF + j]; }
                ++q;
            }
    }
    __syncthreads();
    {
        const int L2 = threadIdx.x >> 3, j = threadIdx.x & 7;
        const int lo2 = s_lo[L2], len = s_hi[L2] - lo2;
        if (len > 0 && len <= 32) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * j + e < len) { val[lo2 + 4 * j + e] = s_val[L2][4 * j + e]; idx[lo2 + 4 * j + e] = s_idx[L2][4 * j + e]; }
        }
    }
}

// entries per row of a factor, from its records (the CSR row pointers of the static form are made when somebody asks for them)
template <int KIND>
__global__ void __launch_bounds__(512)
k_st_rowcounts(int32_t *__restrict__ cnt, const int32_t *__restrict__ wtab, const int32_t *__restrict__ ltab, const v2d *__restrict__ pk,
               const int32_t *__restrict__ ysrc)
{
    constexpr bool FWD = (KIND == SWEEP_FWD_LAST_ASC);
    constexpr int DR = FWD ? 1 : -1;
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    if (c >= nch) return;
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const int32_t *T = ltab + (size_t)slot * kStTab;
    const int k = tmin + c - T[ST_SKEW];
    if (k < 0 || k >= T[ST_CNT]) return;
    const int pos = FWD ? 0 : ysrc[slot] - 64 * k;
    const v2d *p = FWD ? pk + ((size_t)base + c) * 128 + L : pk + (size_t)(pos >> 6) * 128 + (pos & 63);
    const v2d a = p[0], b = p[64];
    cnt[T[ST_FIRST] + DR * k] = 1 + (st_bits(a.x) != kAbsent ? 1 : 0) + (st_bits(a.y) != kAbsent ? 1 : 0) + (st_bits(b.x) != kAbsent ? 1 : 0);
}

int st_make_csr(hipStream_t st, int32_t n, const PackedSweep &pl, const PackedSweep &pu, DevMat *L, DevMat *U)
{
    int32_t *cnt = nullptr;
    ILUPP_HIP(pool_malloc(&cnt, sizeof(int32_t) * (size_t)n));
    for (int d = 0; d < 2; ++d) {
        const PackedSweep &ps = d == 0 ? pl : pu;
        DevMat *M = d == 0 ? L : U;
        const dim3 grid((unsigned)(ps.nwg * 4), (unsigned)((ps.max_chunks + 7) / 8));
        if (d == 0)
            hipLaunchKernelGGL((k_st_rowcounts<SWEEP_FWD_LAST_ASC>), grid, dim3(512), 0, st, cnt, ps.wtab, ps.ltab,
                               reinterpret_cast<const v2d *>(ps.pk), static_cast<const int32_t *>(nullptr));
        else
            hipLaunchKernelGGL((k_st_rowcounts<SWEEP_BWD_FIRST_ASC>), grid, dim3(512), 0, st, cnt, ps.wtab, ps.ltab,
                               reinterpret_cast<const v2d *>(ps.pk), ps.ysrc);
        const int rc = csr_ptrs_from_counts(st, n, cnt, M);
        if (rc) { (void)pool_free(cnt); return rc; }
    }
    ILUPP_HIP(pool_free(cnt));
    return ILUPP_OK;
}

void st_unpack(hipStream_t st, const DevMat &M, const Schedule &sch, const PackedSweep &ps)
{
    (void)sch;
    const dim3 grid((unsigned)(ps.nwg * 4), (unsigned)((ps.max_chunks + 7) / 8));
    if ((SweepKind)ps.kind == SWEEP_FWD_LAST_ASC)
        hipLaunchKernelGGL((k_st_unpack<SWEEP_FWD_LAST_ASC>), grid, dim3(512), 0, st, M.ptr, M.idx, M.val, ps.wtab, ps.ltab,
                           reinterpret_cast<const v2d *>(ps.pk), static_cast<const int32_t *>(nullptr));
    else
        hipLaunchKernelGGL((k_st_unpack<SWEEP_BWD_FIRST_ASC>), grid, dim3(512), 0, st, M.ptr, M.idx, M.val, ps.wtab, ps.ltab,
                           reinterpret_cast<const v2d *>(ps.pk), ps.ysrc);
    ILUPP_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// the step-major exchange of a schedule: ordinals of the exported lanes of each workgroup, the workgroup's row length and steps
static constexpr int kStXAlign = 16;          // first step of a workgroup's exchange rows: its first step rounded down to this
__global__ void __launch_bounds__(kThreads)
k_st_xch_layout(const int32_t *__restrict__ exported, const int32_t *__restrict__ ltab, const int32_t *__restrict__ wtab,
                int32_t *__restrict__ xe, int32_t *__restrict__ xw, int32_t *__restrict__ xsz, int32_t *__restrict__ flags)
{
    __shared__ int s_cnt[4], s_pairs;
    const int wg = blockIdx.x, t = threadIdx.x, wv = t >> 6;
    const int slot = wg * kThreads + t;
    const int32_t *T = ltab + (size_t)slot * kStTab;
    const bool ex = T[ST_CNT] > 0 && exported[slot] != 0;
    // the courier wave of the kernels serves at most 64 (lane, dependency) pairs that come from earlier workgroups
    __shared__ int s_phase[16];
    if (t == 0) s_pairs = 0;
    if (t < 16) s_phase[t] = 0;
    __syncthreads();
    // (the vector wave of st_wave.hip serves at most 16 lanes per skew modulo 16: flags[9] & 8, as in k_st_xch_pair)
    if (T[ST_CNT] > 0 && atomicAdd(&s_phase[T[ST_SKEW] & 15], 1) >= 16) atomicOr(&flags[9], 8);
    int ng = 0;
    for (int j = 0; j < 3; ++j) ng += (j < T[ST_ND] && T[ST_CNT] > 0 && (T[ST_SRC + j] & 3) == ST_GHOST) ? 1 : 0;
    if (ng) atomicAdd(&s_pairs, ng);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(ex);
    if ((t & 63) == 0) s_cnt[wv] = __popcll(bal);
    __syncthreads();
    int before = 0;
    for (int q = 0; q < wv; ++q) before += s_cnt[q];
    xe[slot] = ex ? before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0)) : -1;
    if (t == 0) {
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const int E = (total + 15) & ~15;
        int tlo = 0x7fffffff, thi = -0x7fffffff;
        for (int q = 0; q < 4; ++q) {
            const int a = wtab[(size_t)(wg * 4 + q) * 4 + 1], b = wtab[(size_t)(wg * 4 + q) * 4 + 2];
            if (b > 0) { tlo = min(tlo, a); thi = max(thi, a + b); }
        }
        if (thi <= tlo) { tlo = 0; thi = 0; }
        tlo &= ~(kStXAlign - 1);
        xw[wg * 4 + 0] = E; xw[wg * 4 + 1] = tlo; xw[wg * 4 + 2] = thi - tlo; xw[wg * 4 + 3] = 0;
        xsz[wg] = E * (thi - tlo);
        if (s_pairs > 64) atomicOr(&flags[0], 32);
        if (total > 64) atomicOr(&flags[9], 4);          // (st_wave.hip's courier exports with one store instruction per step)
    }
}
// the exchange layouts of BOTH schedules of an ILU(0) and the offsets of their rows with one launch: block (w, d) lays out workgroup w
// of direction d (k_st_xch_layout's job); the last block of a direction to finish scans that direction's sizes (an atomic ticket;
// sizes through agent-scope stores and loads: the blocks sit on different XCDs) and writes the row offsets and tot[2 d] = the offset
// of the last workgroup, tot[2 d + 1] = its size.  (Before: a launch for the layout, two for the scan, one for the offsets, per direction
// -- 5 us each in an analysis of 450.)
struct StXchArgs { const int32_t *exported, *ltab, *wtab; int32_t *xe, *xw, *xsz, *flags; };
__global__ void __launch_bounds__(kThreads)
k_st_xch_pair(StXchArgs F, StXchArgs B, int32_t nwg, int32_t *__restrict__ tot)
{
    __shared__ int s_cnt[4], s_pairs, s_last;
    __shared__ int part[kThreads];
    const StXchArgs &X = blockIdx.y == 0 ? F : B;
    const int wg = blockIdx.x, t = threadIdx.x, wv = t >> 6;
    const int slot = wg * kThreads + t;
    const int32_t *T = X.ltab + (size_t)slot * kStTab;
    const bool ex = T[ST_CNT] > 0 && X.exported[slot] != 0;
    __shared__ int s_phase[16];
    if (t == 0) s_pairs = 0;
    if (t < 16) s_phase[t] = 0;
    __syncthreads();
    // (st_wave.hip's vector wave serves the lanes whose skew is congruent to the step modulo 16 with two instructions of 8 lanes: flags[9] & 8
    // when a workgroup has more than 16 such lanes)
    if (T[ST_CNT] > 0 && atomicAdd(&s_phase[T[ST_SKEW] & 15], 1) >= 16) atomicOr(&X.flags[9], 8);
    int ng = 0;
    for (int j = 0; j < 3; ++j) ng += (j < T[ST_ND] && T[ST_CNT] > 0 && (T[ST_SRC + j] & 3) == ST_GHOST) ? 1 : 0;
    if (ng) atomicAdd(&s_pairs, ng);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(ex);
    if ((t & 63) == 0) s_cnt[wv] = __popcll(bal);
    __syncthreads();
    int before = 0;
    for (int q = 0; q < wv; ++q) before += s_cnt[q];
    X.xe[slot] = ex ? before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0)) : -1;
    if (t == 0) {
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const int E = (total + 15) & ~15;
        int tlo = 0x7fffffff, thi = -0x7fffffff;
        for (int q = 0; q < 4; ++q) {
            const int a = X.wtab[(size_t)(wg * 4 + q) * 4 + 1], b = X.wtab[(size_t)(wg * 4 + q) * 4 + 2];
            if (b > 0) { tlo = min(tlo, a); thi = max(thi, a + b); }
        }
        if (thi <= tlo) { tlo = 0; thi = 0; }
        tlo &= ~(kStXAlign - 1);
        X.xw[wg * 4 + 0] = E; X.xw[wg * 4 + 1] = tlo; X.xw[wg * 4 + 2] = thi - tlo;
        st_agent_i32(&X.xsz[wg], E * (thi - tlo));
        if (s_pairs > 64) atomicOr(&X.flags[0], 32);
        if (total > 64) atomicOr(&X.flags[9], 4);          // (st_wave.hip's courier exports with one store instruction per step)
        __threadfence();
        s_last = atomicAdd(&X.flags[11], 1) == nwg - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // exclusive scan of this direction's sizes (one block)
    const int per = (nwg + kThreads - 1) / kThreads;
    int sum = 0;
    for (int i = t * per; i < (t + 1) * per && i < nwg; ++i) sum += ld_agent_i32(&X.xsz[i]);
    part[t] = sum;
    __syncthreads();
    if (t == 0) { int run = 0; for (int i = 0; i < kThreads; ++i) { const int c = part[i]; part[i] = run; run += c; } }
    __syncthreads();
    int run = part[t];
    for (int i = t * per; i < (t + 1) * per && i < nwg; ++i) {
        const int c = ld_agent_i32(&X.xsz[i]);
        X.xw[i * 4 + 3] = run;
        if (i == nwg - 1) { tot[2 * blockIdx.y] = run; tot[2 * blockIdx.y + 1] = c; }
        run += c;
    }
}
__global__ void k_st_xch_rows(int32_t nwg, const int32_t *__restrict__ xoff, int32_t *__restrict__ xw)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwg) xw[w * 4 + 3] = xoff[w];
}

static void st_structure(hipStream_t st, const Schedule &sch, PackedSweep *ps, int kind)
{
    ps->nwg = sch.nslots / kThreads;
    ps->kind = kind;
    ILUPP_HIP(pool_malloc(&ps->ltab, sizeof(int32_t) * kStTab * (size_t)sch.nslots));
    ILUPP_HIP(pool_malloc(&ps->skew, sizeof(int32_t) * (size_t)sch.nslots));
    ILUPP_HIP(pool_malloc(&ps->wtab, sizeof(int32_t) * 16 * (size_t)ps->nwg));
    ILUPP_HIP(pool_malloc(&ps->flags, 64));
    ILUPP_HIP(pool_malloc(&ps->dump, sizeof(double) * (size_t)sch.nslots));
    ILUPP_HIP(hipMemsetAsync(ps->flags, 0, 64, st));
}

static void st_pack_values(hipStream_t st, const DevMat &A, PackedSweep *pl, PackedSweep *pu, FactorLM *f)
{
    const int groups = (int)((pl->max_chunks + 7) / 8);                                     // every lane has at most max_chunks rows
    hipLaunchKernelGGL(k_st_rows, dim3((unsigned)(pl->nwg * 4 * groups)), dim3(512), 0, st, A.ptr, A.idx, A.val, (int64_t)A.nnz,
                       f->xbase + pl->nwg * kThreads, pl->wtab, reinterpret_cast<v2d *>(f->pkA), reinterpret_cast<v2d *>(pu->pk),
                       pl->flags, groups);
}

// The whole static analysis of an ILU(0): true when the factor kernel and both sweeps can run from lane tables
// (pl, pu, f then complete, the values of A packed); false leaves the three objects released.
bool st_analyse_ilu0(hipStream_t st, const DevMat &A, const Schedule &fwd, const Schedule &bwd, PackedSweep *pl,
                     PackedSweep *pu, FactorLM *f, SideJoin *join, const GridDims *grid)
{
    pl->release(); pu->release(); f->release();
    static const bool off = getenv("ILUPP_NO_PACKED") != nullptr ||
                            getenv("ILUPP_CLASSIC_ANALYSIS") != nullptr || getenv("ILUPP_NO_STATIC") != nullptr;
    static const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    if (off || A.nnz > 7 * (int64_t)A.n || A.nnz < 16 || fwd.nslots < kThreads || fwd.nslots != bwd.nslots || !A.val) return false;
    const int nwg = fwd.nslots / kThreads;
    const int nslots = fwd.nslots;
    st_structure(st, fwd, pl, (int)SWEEP_FWD_LAST_ASC);
    st_structure(st, bwd, pu, (int)SWEEP_BWD_FIRST_ASC);
    bool linked_by_grid = false;
    if (grid) {
        // a box grid (grid.hip): slot tables, lane templates and the link of the two schedules follow from its dimensions
        pu->built = true;
        ILUPP_HIP(pool_malloc(&pu->uslot, sizeof(int32_t) * (size_t)fwd.nslots));
        linked_by_grid = grid_lane_tables(st, *grid, fwd, bwd, pl->ltab, pu->ltab, pl->flags, pu->flags, pu->uslot, pl->skew, pu->skew,
                                          pl->wtab, pu->wtab, st_wx_on());
    } else {
        StTplArgs tf, tb;
        tf.B = fwd.B; tf.nb = fwd.nb; tf.start = fwd.start; tf.blk2slot = fwd.blk2slot; tf.sfirst = fwd.sfirst; tf.scount = fwd.scount;
        tf.exported = fwd.exported; tf.ltab = pl->ltab; tf.flags = pl->flags;
        tb.B = bwd.B; tb.nb = bwd.nb; tb.start = bwd.start; tb.blk2slot = bwd.blk2slot; tb.sfirst = bwd.sfirst; tb.scount = bwd.scount;
        tb.exported = bwd.exported; tb.ltab = pu->ltab; tb.flags = pu->flags;
        hipLaunchKernelGGL(k_st_template_pair, dim3((unsigned)nwg, 2), dim3(kThreads), 0, st, A.ptr, A.idx, tf, tb);
        pu->built = true;
        lm_link_factor(st, fwd, bwd, pu);               // forward slot -> backward slot of the same chain (flags[3] when there is none)
    }
    if (!linked_by_grid)
        hipLaunchKernelGGL(k_st_link_pair, dim3((unsigned)nwg, 2), dim3(kThreads), 0, st, pl->ltab, pu->ltab, pu->uslot, pl->skew, pu->skew,
                           pl->wtab, pu->wtab, pl->flags, pu->flags, st_wx_on() ? 1 : 0);
    // (a box grid in 16 x 16 patches: one closed-form launch below instead of the scan, the clears, the slot maps, k_st_scat and the
    // exchange layout -- grid.hip: k_grid_scat; ILUPP_GRID_SCAT=0: the general kernels on the grid's lane tables)
    static const bool grid_scat_on = []() { const char *e = getenv("ILUPP_GRID_SCAT"); return !(e && atoi(e) == 0); }();
    const bool scat_by_grid = linked_by_grid && grid_scat_on && getenv("ILUPP_SD_VERIFY") == nullptr;
    if (!scat_by_grid) hipLaunchKernelGGL(k_st_scan_pair, dim3(2), dim3(kThreads), 0, st, nwg * 4, pl->wtab, pl->flags, pu->wtab, pu->flags);
    int32_t *inv = nullptr;
    ILUPP_HIP(pool_malloc(&inv, sizeof(int32_t) * (size_t)nslots));
    ILUPP_HIP(pool_malloc(&pu->ysrc, sizeof(int32_t) * (size_t)nslots));
    const unsigned gb = (unsigned)((nslots + 255) / 256);
    // the rows pass' lane records (behind nslots unused ints); with them the lane fields and lane-level checks of the factor kernel
    // that reads A's values where they lie (st_direct.hip; its verdict: pl->flags[8])
    const bool try_direct = st_direct_prepare(st, A, fwd, pl->flags + 8);
    if (!scat_by_grid) {
        // (what has to be cleared: one launch)
        StClearList cl;
        cl.p[0] = inv; cl.v[0] = -1; cl.n[0] = nslots;
        cl.p[1] = pu->ysrc; cl.v[1] = 0; cl.n[1] = nslots;
        cl.p[2] = nullptr; cl.v[2] = 0; cl.n[2] = 0;
        cl.p[3] = nullptr; cl.v[3] = 0; cl.n[3] = 0;
        hipLaunchKernelGGL(k_st_clear, dim3(gb), dim3(256), 0, st, cl);
        hipLaunchKernelGGL(k_st_inv_ysrc, dim3(gb), dim3(256), 0, st, nslots, pu->uslot, inv, fwd.scount, pl->wtab, pl->skew, pu->ysrc);
    }
    ILUPP_HIP(pool_malloc(&f->xbase, sizeof(int32_t) * (size_t)nslots * 33));
    f->xbase_len = (int64_t)nslots * 33;
    if (!scat_by_grid) {
        hipLaunchKernelGGL(k_st_scat, dim3(gb), dim3(256), 0, st, nslots, pl->ltab, pu->ltab, pu->uslot, inv, pl->wtab, pu->wtab,
                           f->xbase + nslots, pl->flags, A.ptr, try_direct ? pl->flags + 8 : static_cast<int32_t *>(nullptr));
        if (try_direct) st_direct_verify(st, A, fwd, pl, pu, pl->flags + 8);
    }
    // the exchange layouts of both schedules; their sizes come back with the flags (one wait for all)
    int32_t xtot[2][2];
    int32_t *xsz = nullptr;
    int32_t hl[12], hu[12];
    {
        ILUPP_HIP(pool_malloc(&xsz, sizeof(int32_t) * ((size_t)nwg * 2 + 4)));
        PackedSweep *pp[2] = {pl, pu};
        const Schedule *ss[2] = {&fwd, &bwd};
        StXchArgs xa[2];
        for (int d = 0; d < 2; ++d) {
            ILUPP_HIP(pool_malloc(&pp[d]->xe, sizeof(int32_t) * (size_t)nslots));
            ILUPP_HIP(pool_malloc(&pp[d]->xw, sizeof(int32_t) * (size_t)nwg * 4));
            xa[d].exported = ss[d]->exported; xa[d].ltab = pp[d]->ltab; xa[d].wtab = pp[d]->wtab;
            xa[d].xe = pp[d]->xe; xa[d].xw = pp[d]->xw; xa[d].xsz = xsz + (size_t)d * nwg; xa[d].flags = pp[d]->flags;
        }
        int32_t *tot = xsz + (size_t)nwg * 2;
        if (scat_by_grid)
            grid_scat_tables(st, *grid, fwd, pl->ltab, pu->ltab, pl->wtab, pu->wtab, pu->ysrc, f->xbase + nslots, pl->xe, pu->xe, pl->xw, pu->xw,
                             pl->flags, pu->flags, tot);
        else
            hipLaunchKernelGGL(k_st_xch_pair, dim3((unsigned)nwg, 2), dim3(kThreads), 0, st, xa[0], xa[1], (int32_t)nwg, tot);
        D2HItem items[4];
        items[0] = {f->chk_xtot, tot, 4 * sizeof(int32_t)};
        items[1] = {f->chk_hl, pl->flags, sizeof(f->chk_hl)};
        items[2] = {f->chk_hu, pu->flags, sizeof(f->chk_hu)};
        int ni = 3;
        if (join) {
            // (a verdict from a side stream comes home with this read-back)
            ILUPP_HIP(hipStreamWaitEvent(st, join->ev, 0));
            items[ni++] = {join->host, join->dev, sizeof(int32_t)};
            join->done = true;
        }
        ILUPP_HIP(d2h_async_many(st, items, ni));
    }
    // A box grid in 16 x 16 patches: the sizes are known from the dimensions (grid.hip: grid_predict_sizes), so nothing is waited for
    // here -- the factor kernel follows the lane-table kernels at once, and what the device found is compared with the prediction when
    // the construction's last read-back has arrived (f->spec; api.hip).  ILUPP_NO_SPEC=1: wait as for any other matrix.
    {
        static const bool nospec = getenv("ILUPP_NO_SPEC") != nullptr;
        int64_t pn = 0, px = 0; int32_t pm = 0, pxl = 0;
        f->spec = grid && !nospec && !join && st_wx_on() && try_direct && getenv("ILUPP_NO_WXF") == nullptr &&
                  grid_predict_sizes(*grid, fwd.tile_ty, fwd.tile_tz, &pn, &pm, &px, &pxl);
        if (f->spec) {
            for (int i = 0; i < 12; ++i) hl[i] = hu[i] = 0;
            hl[1] = hu[1] = (int32_t)pn; hl[2] = hu[2] = pm;
            xtot[0][0] = xtot[1][0] = (int32_t)px; xtot[0][1] = xtot[1][1] = pxl;
            f->pred[0] = (int32_t)pn; f->pred[1] = pm; f->pred[2] = (int32_t)px; f->pred[3] = pxl;
        } else {
            ILUPP_HIP(stream_sync(st));
            for (int i = 0; i < 12; ++i) { hl[i] = f->chk_hl[i]; hu[i] = f->chk_hu[i]; }
            for (int i = 0; i < 4; ++i) (&xtot[0][0])[i] = f->chk_xtot[i];
        }
    }
    ILUPP_HIP(pool_free(inv)); ILUPP_HIP(pool_free(xsz));
    // row slots of the chunks against rows: lines of 8 rows in a 16 x 16 patch (30 steps of skew) are 4.75 slots per row, and
    // still 400 times faster than what the other generations make of 90 000 such lines
    const int64_t lim = 6 * (int64_t)A.n + 64 * 4 * (int64_t)nwg;
    if (hl[0] || hu[0] || hu[3] || hl[1] <= 0 || hu[1] <= 0 || (int64_t)hl[1] * 64 > lim || (int64_t)hu[1] * 64 > lim ||
        hl[1] + 4 * nwg >= kStMaxChunks || hu[1] + 4 * nwg >= kStMaxChunks ||
        // (the exchange layout is indexed with 32 bits: at most 256 exported lanes x the steps of a workgroup, each)
        (int64_t)nwg * kThreads * ((int64_t)(hl[2] > hu[2] ? hl[2] : hu[2]) + 2 * kStXAlign) > 0x7fffffffLL) {
        if (dbg) fprintf(stderr, "[ilupp] static analysis: structure rejected (flags %d %d link %d, %d %d chunks)\n", hl[0], hu[0], hu[3], hl[1], hu[1]);
        pl->release(); pu->release(); f->release();
        return false;
    }
    pl->nchunks = hl[1]; pl->max_chunks = hl[2];
    pu->nchunks = hu[1]; pu->max_chunks = hu[2];
    // one spare chunk per wave each: where waves / lanes without a row at a step store
    ILUPP_HIP(pool_malloc(&pl->pk, (size_t)(pl->nchunks + 4 * nwg) * 2048));
    ILUPP_HIP(pool_malloc(&pu->pk, (size_t)(pl->nchunks + 4 * nwg) * 2048));      // (the forward schedule's order: see k_sptrsv_st)
    pl->built = true;
    // vectors travel level-major: the right-hand side and the intermediate vector in the L sweep's order (pl->ybuf, in place), the
    // result in the U sweep's (pu->xlm)
    ILUPP_HIP(pool_malloc(&pl->ybuf, sizeof(double) * 64 * (size_t)(pl->nchunks + 4 * nwg)));
    ILUPP_HIP(pool_malloc(&pu->xlm, sizeof(double) * 64 * (size_t)(pu->nchunks + 4 * nwg)));
    pu->y_chunks = pl->nchunks + 4 * nwg;
    pl->xch_len = (int64_t)xtot[0][0] + xtot[0][1] + 64;
    pu->xch_len = (int64_t)xtot[1][0] + xtot[1][1] + 64;
    ILUPP_HIP(pool_malloc(&pl->xch, sizeof(double) * (size_t)pl->xch_len));
    ILUPP_HIP(pool_malloc(&pu->xch, sizeof(double) * (size_t)pu->xch_len));
    f->direct = try_direct && hl[8] == 0;
    f->wxf = f->direct && (hl[9] & ~8) == 0 && (hu[9] & ~8) == 0 && hl[10] == 0 && st_wx_on() && getenv("ILUPP_NO_WXF") == nullptr;
    if (!f->direct) {
        // lanes not uniform enough for the direct feed: factor records made by the rows pass (it proves what they rely on, row by row)
        if (dbg && try_direct) fprintf(stderr, "[ilupp] static analysis: lanes not uniform (flags %d): factor records\n", hl[8]);
        ILUPP_HIP(pool_malloc(&f->pkA, (size_t)pl->nchunks * 4096));
        st_pack_values(st, A, pl, pu, f);
        int32_t gl[4];
        ILUPP_HIP(d2h_async(st, gl, pl->flags, sizeof(gl)));
        ILUPP_HIP(stream_sync(st));
        if (dbg) fprintf(stderr, "[ilupp] static analysis: row flags %d, %d+%d chunks\n", gl[0], hl[1], hu[1]);
        if (gl[0]) { pl->release(); pu->release(); f->release(); return false; }
    } else if (dbg) {
        fprintf(stderr, "[ilupp] static analysis: %d+%d chunks, direct feed\n", hl[1], hu[1]);
    }
    pl->valid = pu->valid = true;
    pl->stat = pu->stat = true;
    pl->wx = (hl[9] & ~8) == 0; pu->wx = (hu[9] & ~8) == 0;
    pl->vec_ok = pl->wx && (hl[9] & 8) == 0; pu->vec_ok = pu->wx && (hu[9] & 8) == 0;
    if (dbg) fprintf(stderr, "[ilupp] static analysis: wave-exchange kernels: forward %s, backward %s\n", pl->wx ? "yes" : "no", pu->wx ? "yes" : "no");
    pu->linked = true;
    f->built = true;
    f->stat = true;
    f->values_packed = !f->direct;
    return true;
}

int ilu0_numeric_st(hipStream_t st, const DevMat &A, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu, FactorLM *f,
                    int32_t *d_ctrl, float *kernel_ms, hipEvent_t e0, hipEvent_t e1)
{
    (void)fwd;
    pl->fmt = pu->fmt = 0;                               // (the factor kernels below write records by template position)
    if (f->direct && f->wxf && (uint64_t)(pl->nchunks + 4 * (int64_t)pl->nwg) * 2048u < 0xfff00000ull)
        return ilu0_numeric_wx(st, A, pl, pu, d_ctrl, kernel_ms, e0, e1);
    if (f->direct) {
        const int rc = ilu0_numeric_sd(st, A, pl, pu, d_ctrl, kernel_ms, e0, e1);
        // the sweeps of st_wave.hip read class-aligned records
        if (rc == ILUPP_OK && pl->wx && pu->wx && st_wx_on() && (uint64_t)(pl->nchunks + 4 * (int64_t)pl->nwg) * 2048u < 0xfff00000ull)
            wx_convert_records(st, pl, pu, 1);
        return rc;
    }
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    fill_u64(st, reinterpret_cast<unsigned long long *>(pl->xch), pl->xch_len, kSentinel);
    const bool repack = !f->values_packed;
    if (repack) {
        // new values on the same pattern: the rows pass once more (it re-proves what it relies on)
        ILUPP_HIP(hipMemsetAsync(pl->flags, 0, 4, st));
        st_pack_values(st, A, pl, pu, f);
    }
    f->values_packed = false;
    StFArgs a;
    a.ltab = pl->ltab; a.wtab = pl->wtab;
    a.pkA = reinterpret_cast<const v2d *>(f->pkA);
    a.pkL = reinterpret_cast<v2d *>(pl->pk); a.pkU = reinterpret_cast<v2d *>(pu->pk);
    a.nchL = (int32_t)pl->nchunks;
    a.xe = pl->xe; a.xw = pl->xw; a.xch = pl->xch; a.ctrl = d_ctrl;
    ILUPP_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_ilu0_st, dim3((unsigned)pl->nwg), dim3(kStWgThreads), 0, st, a);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4], fl[4] = {0, 0, 0, 0};
    ILUPP_HIP(d2h_async(st, ctrl, d_ctrl, 16));
    if (repack) ILUPP_HIP(d2h_async(st, fl, pl->flags, 16));
    ILUPP_HIP(stream_sync(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    if (fl[0] != 0) { set_error("ILU0: the matrix handed to the numeric re-factorisation does not have the analysed pattern"); return ILUPP_ERR_INVALID; }
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

static void st_solo_attr_T()
{
    static std::once_flag once[64];      // once per device
    int dev = 0;
    ILUPP_HIP(hipGetDevice(&dev));
    std::call_once(once[dev & 63], [] {
        ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_st<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
        ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_st<-1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
    });
}

static inline int st_vec_groups(int max_chunks) { const int g = (max_chunks + 31) / 32; return g < 3 ? (g < 1 ? 1 : g) : 3; }

void st_vec_to_lm(hipStream_t st, const PackedSweep &ps, const double *nat, double *lm)
{
    hipLaunchKernelGGL((k_st_vec<1, true>), dim3((unsigned)(ps.nwg * 4), (unsigned)st_vec_groups(ps.max_chunks)), dim3(512), 0, st, ps.ltab, ps.wtab,
                       const_cast<double *>(nat), lm);
}
void st_vec_from_lm(hipStream_t st, const PackedSweep &ps, double *nat)
{
    hipLaunchKernelGGL((k_st_vec<-1, false>), dim3((unsigned)(ps.nwg * 4), (unsigned)st_vec_groups(ps.max_chunks)), dim3(512), 0, st, ps.ltab, ps.wtab,
                       nat, ps.xlm);
}

// One sweep of an apply.  Forward: `rhs` (natural order) -> the intermediate vector in `ypk_out` (the forward sweep's ybuf, in the
// forward sweep's level-major order); backward: `ypk_in` (the same buffer) -> the result in `out` (natural order).
int sptrsv_st(hipStream_t st, const PackedSweep &ps, const Schedule &sch, int32_t n, const double *rhs, double *out,
              int32_t *d_ticket, int32_t *d_err, double *ypk_out, const double *ypk_in, const int32_t *ysrc)
{
    (void)sch;
    if (ps.fmt >= 1) return sptrsv_wx(st, ps, n, rhs, out, d_ticket, d_err, ypk_out, ypk_in, ysrc);
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
    const dim3 grid((unsigned)ps.nwg);
    // one workgroup per CU: with 40 KB of LDS three would fit, and the dispatcher does put several on one CU while others stay
    // empty; the steps of two schedules on the same four SIMDs take turns.  The unused dynamic LDS makes the workgroup too big
    // for a neighbour.
    {
        static std::once_flag once[64];      // once per device
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_st<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
            ILUPP_HIP(hipFuncSetAttribute((const void *)k_sptrsv_st<-1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStSoloLds));
        });
    }
    if (ps.pair) st_solo_attr_T();
    if (fwd) {
        hipLaunchKernelGGL((k_st_vec<1, true>), dim3((unsigned)(ps.nwg * 4), (unsigned)st_vec_groups(ps.max_chunks)), dim3(512), 0, st, ps.ltab, ps.wtab,
                           const_cast<double *>(rhs), lml);
        // (a pair of stored factors: the forward factor has a diagonal of its own)
        if (ps.pair) hipLaunchKernelGGL((k_sptrsv_st<1, true>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
        else hipLaunchKernelGGL((k_sptrsv_st<1, false>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
    } else {
        if (ps.pair && ps.desc) hipLaunchKernelGGL((k_sptrsv_st<-1, true>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
        else hipLaunchKernelGGL((k_sptrsv_st<-1, false>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
        hipLaunchKernelGGL((k_st_vec<-1, false>), dim3((unsigned)(ps.nwg * 4), (unsigned)st_vec_groups(ps.max_chunks)), dim3(512), 0, st, ps.ltab, ps.wtab,
                           out, ps.xlm);
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

// ---------------------------------------------------------------------------------------------
// static sweeps for a pair of STORED factors (IChol0, ICholT without fill, ILU(0) of a matrix whose factorisation took another
// generation): the same lane templates, skews, chunk tables and exchange as for ILU(0), proven on the factors' own patterns; the
// records are the factors' values.  One thread per (forward lane, row): the row of the forward factor (diagonal last) and the row
// of the backward factor (diagonal first), each entry matched against the lane's template and its range of rows.
// ---------------------------------------------------------------------------------------------
// the runs of eight consecutive rows of 64 lanes (at most 32 entries each) of a CSR matrix into LDS, eight threads per lane (the pass of
// k_st_pack_lower): q0, q1 = this thread's row; use = the row counts for its lane's run; false for a lane whose run does not fit
struct StRuns { double val[64][33]; int idx[64][33]; int lo[64], hi[64]; };
__device__ __forceinline__ void st_stage_runs(StRuns &S, const int32_t *__restrict__ idx, const double *__restrict__ val, const int L,
                                              const bool use, const int q0, const int q1)
{
    if (threadIdx.x < 64) { S.lo[threadIdx.x] = 0x7fffffff; S.hi[threadIdx.x] = -1; }
    __syncthreads();
    if (use) { atomicMin(&S.lo[L], q0); atomicMax(&S.hi[L], q1); }
    __syncthreads();
    const int L2 = threadIdx.x >> 3, j = threadIdx.x & 7;
    const int lo2 = S.lo[L2], len = S.hi[L2] - lo2;
    if (len > 0 && len <= 32) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * j + e < len) { S.val[L2][4 * j + e] = val[lo2 + 4 * j + e]; S.idx[L2][4 * j + e] = idx[lo2 + 4 * j + e]; }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(512)
k_st_pack_pair(const int32_t *__restrict__ Lptr, const int32_t *__restrict__ Lidx, const double *__restrict__ Lval,
               const int32_t *__restrict__ Uptr, const int32_t *__restrict__ Uidx, const double *__restrict__ Uval,
               const int32_t *__restrict__ ltabF, const int32_t *__restrict__ ltabB, const int32_t *__restrict__ uslot,
               const int32_t *__restrict__ wtab, v2d *__restrict__ pkL, v2d *__restrict__ pkU, int32_t *__restrict__ flags,
               const int fmt1, const int descB)
{
    __shared__ StRuns SL, SU;
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const int32_t *T = ltabF + (size_t)slot * kStTab;
    const int k = tmin + c - T[ST_SKEW];
    const int cnt = T[ST_CNT];
    const bool live = c < nch && k >= 0 && k < cnt;
    const int su = live ? uslot[slot] : 0;
    const int r = live ? T[ST_FIRST] + k : 0;
    const int q0 = live ? Lptr[r] : 0, q1 = live ? Lptr[r + 1] : 0;
    const int p0 = live ? Uptr[r] : 0, p1 = live ? Uptr[r + 1] : 0;
    const bool okL = q1 - q0 >= 1 && q1 - q0 <= 4, okU = p1 - p0 >= 1 && p1 - p0 <= 4;
    // (the rows of a lane's eight steps are consecutive rows of both factors: one run each)
    st_stage_runs(SL, Lidx, Lval, L, live && okL, q0, q1);
    st_stage_runs(SU, Uidx, Uval, L, live && okU, p0, p1);
    if (!live) {
        // (class-aligned records, st_wave.hip's format 1: the zero record wherever a lane has no row at a step of its wave)
        if (fmt1 && c < nch) {
            v2d z0, z1; z0.x = 0.0; z0.y = 0.0; z1.x = 0.0; z1.y = 1.0;
            v2d *pz = pkL + ((size_t)base + c) * 128 + L; pz[0] = z0; pz[64] = z1;
            pz = pkU + ((size_t)base + c) * 128 + L; pz[0] = z0; pz[64] = z1;
        }
        return;
    }
    if (su < 0) { atomicOr(&flags[0], 64); return; }
    const int32_t *TB = ltabB + (size_t)su * kStTab;
    const double absent = st_dbl(kAbsent);
    double lv[3] = {absent, absent, absent}, uv[3] = {absent, absent, absent};
    int bad = 0;
    const int loL = SL.lo[L], loU = SU.lo[L];
    const bool stL = okL && SL.hi[L] - loL <= 32, stU = okU && SU.hi[L] - loU <= 32;
#define LI(q) (stL ? SL.idx[L][(q) - loL] : Lidx[q])
#define LV(q) (stL ? SL.val[L][(q) - loL] : Lval[q])
#define UI(q) (stU ? SU.idx[L][(q) - loU] : Uidx[q])
#define UV(q) (stU ? SU.val[L][(q) - loU] : Uval[q])
    if (!okL || LI(q1 - 1) != r) bad = 1;
    for (int q = q0; q < q1 - 1 && !bad; ++q) {
        const int o = LI(q) - r;
        int hit = -1;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (j < T[ST_ND] && T[ST_OFF + j] == o) hit = j;
        if (hit < 0 || k < T[ST_KLO + hit] || k >= T[ST_KHI + hit]) bad = 1;
        else { const double cv = st_clean(LV(q)); if (hit == 0) lv[0] = cv; else if (hit == 1) lv[1] = cv; else lv[2] = cv; }
    }
    const int kb = cnt - 1 - k;
    if (!okU || UI(p0) != r) bad = 1;
    for (int q = p0 + 1; q < p1 && !bad; ++q) {
        const int o = UI(q) - r;
        int hit = -1;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (j < TB[ST_ND] && TB[ST_OFF + j] == o) hit = j;
        if (hit < 0 || kb < TB[ST_KLO + hit] || kb >= TB[ST_KHI + hit]) bad = 1;
        else { const double cv = st_clean(UV(q)); if (hit == 0) uv[0] = cv; else if (hit == 1) uv[1] = cv; else uv[2] = cv; }
    }
    if (bad) { atomicOr(&flags[0], 8); return; }
    if (fmt1) {
        // straight into the class-aligned form (what k_wx_convert<true> makes of the records by template position: a missing
        // coefficient is +0.0, the coefficients sit where their dependency class puts them)
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            double *v = side == 0 ? lv : uv;
            int cls[3]; bool ring[3];
            (void)wr_classify(side == 0 ? T : TB, side == 0 ? (slot & 255) : (su & 255), side == 1, cls, ring);
            double o[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (cls[j] != WR_NONE && st_bits(v[j]) != kAbsent) {
                    const int sl = wr_slot_of(cls[j], side == 1 && !descB);
                    if (sl == 0) o[0] = v[j]; else if (sl == 1) o[1] = v[j]; else o[2] = v[j];
                }
            v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
        }
    }
    v2d x;
    v2d *pl_ = pkL + ((size_t)base + c) * 128 + L;
    x.x = lv[0]; x.y = lv[1]; pl_[0] = x;
    x.x = lv[2]; x.y = st_clean(LV(q1 - 1)); pl_[64] = x;
    v2d *pu_ = pkU + ((size_t)base + c) * 128 + L;
    x.x = uv[0]; x.y = uv[1]; pu_[0] = x;
    x.x = uv[2]; x.y = st_clean(UV(p0)); pu_[64] = x;
#undef LI
#undef LV
#undef UI
#undef UV
}

bool st_analyse_pair(hipStream_t st, int32_t n, const DevMat &Lrow, const DevMat &Urow, const Schedule &fwd, const Schedule &bwd,
                     PackedSweep *pl, PackedSweep *pu, bool bwd_desc)
{
    pl->release(); pu->release();
    static const bool off = getenv("ILUPP_NO_PACKED") != nullptr || getenv("ILUPP_NO_STATIC") != nullptr;
    static const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    if (off || Lrow.nnz > 4 * (int64_t)n || Urow.nnz > 4 * (int64_t)n || n < 16 || fwd.nslots < kThreads || fwd.nslots != bwd.nslots ||
        !Lrow.val || !Urow.val || !fwd.exported || !bwd.exported)
        return false;
    const int nwg = fwd.nslots / kThreads;
    const int nslots = fwd.nslots;
    st_structure(st, fwd, pl, (int)SWEEP_FWD_LAST_ASC);
    st_structure(st, bwd, pu, (int)SWEEP_BWD_FIRST_ASC);
    hipLaunchKernelGGL((k_st_template<1>), dim3((unsigned)nwg), dim3(kThreads), 0, st, Lrow.ptr, Lrow.idx, fwd.B, fwd.nb, fwd.start,
                       fwd.blk2slot, fwd.sfirst, fwd.scount, fwd.exported, pl->ltab, pl->flags);
    hipLaunchKernelGGL((k_st_template<-1>), dim3((unsigned)nwg), dim3(kThreads), 0, st, Urow.ptr, Urow.idx, bwd.B, bwd.nb, bwd.start,
                       bwd.blk2slot, bwd.sfirst, bwd.scount, bwd.exported, pu->ltab, pu->flags);
    pu->built = true;
    lm_link_factor(st, fwd, bwd, pu);
    hipLaunchKernelGGL((k_st_link<false>), dim3((unsigned)nwg), dim3(kThreads), 0, st, pl->ltab, static_cast<const int32_t *>(nullptr),
                       static_cast<const int32_t *>(nullptr), pl->skew, pl->wtab, pl->flags, st_wx_on() ? 1 : 0);
    hipLaunchKernelGGL((k_st_link<false>), dim3((unsigned)nwg), dim3(kThreads), 0, st, pu->ltab, static_cast<const int32_t *>(nullptr),
                       static_cast<const int32_t *>(nullptr), pu->skew, pu->wtab, pu->flags, st_wx_on() ? 3 : 0);
    hipLaunchKernelGGL(k_st_scan, dim3(1), dim3(kThreads), 0, st, nwg * 4, pl->wtab, pl->flags);
    hipLaunchKernelGGL(k_st_scan, dim3(1), dim3(kThreads), 0, st, nwg * 4, pu->wtab, pu->flags);
    int32_t xtot[2][2];
    int32_t *xsz = nullptr;
    void *tmp2 = nullptr;
    {
        ILUPP_HIP(pool_malloc(&xsz, sizeof(int32_t) * (size_t)nwg * 4));
        PackedSweep *pp[2] = {pl, pu};
        const Schedule *ss[2] = {&fwd, &bwd};
        size_t tb2 = 0;
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, xsz, xsz + nwg, nwg, st));
        ILUPP_HIP(pool_malloc(&tmp2, tb2));
        for (int d = 0; d < 2; ++d) {
            int32_t *sz = xsz + 2 * d * nwg, *offp = sz + nwg;
            ILUPP_HIP(pool_malloc(&pp[d]->xe, sizeof(int32_t) * (size_t)nslots));
            ILUPP_HIP(pool_malloc(&pp[d]->xw, sizeof(int32_t) * (size_t)nwg * 4));
            hipLaunchKernelGGL(k_st_xch_layout, dim3((unsigned)nwg), dim3(kThreads), 0, st, ss[d]->exported, pp[d]->ltab, pp[d]->wtab,
                               pp[d]->xe, pp[d]->xw, sz, pp[d]->flags);
            ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp2, tb2, sz, offp, nwg, st));
            hipLaunchKernelGGL(k_st_xch_rows, dim3((unsigned)((nwg + 255) / 256)), dim3(256), 0, st, nwg, offp, pp[d]->xw);
            ILUPP_HIP(d2h_async(st, &xtot[d][0], offp + (nwg - 1), sizeof(int32_t)));
            ILUPP_HIP(d2h_async(st, &xtot[d][1], sz + (nwg - 1), sizeof(int32_t)));
        }
    }
    int32_t hl[12], hu[12];
    ILUPP_HIP(d2h_async(st, hl, pl->flags, sizeof(hl)));
    ILUPP_HIP(d2h_async(st, hu, pu->flags, sizeof(hu)));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(xsz)); ILUPP_HIP(pool_free(tmp2));
    const int64_t lim = 6 * (int64_t)n + 64 * 4 * (int64_t)nwg;
    if (hl[0] || hu[0] || hu[3] || hl[1] <= 0 || hu[1] <= 0 || (int64_t)hl[1] * 64 > lim || (int64_t)hu[1] * 64 > lim ||
        hl[1] + 4 * nwg >= kStMaxChunks || hu[1] + 4 * nwg >= kStMaxChunks ||
        (int64_t)nwg * kThreads * ((int64_t)(hl[2] > hu[2] ? hl[2] : hu[2]) + 2 * kStXAlign) > 0x7fffffffLL) {
        if (dbg) fprintf(stderr, "[ilupp] static sweeps of a factor pair: structure rejected (flags %d %d link %d, %d %d chunks)\n", hl[0], hu[0], hu[3], hl[1], hu[1]);
        pl->release(); pu->release();
        return false;
    }
    pl->nchunks = hl[1]; pl->max_chunks = hl[2];
    pu->nchunks = hu[1]; pu->max_chunks = hu[2];
    ILUPP_HIP(pool_malloc(&pl->pk, (size_t)(pl->nchunks + 4 * nwg) * 2048));
    ILUPP_HIP(pool_malloc(&pu->pk, (size_t)(pl->nchunks + 4 * nwg) * 2048));      // (both in the forward schedule's order)
    pl->built = true;
    // (a pair whose lanes fit the wave-exchange classes gets its records class-aligned at once: no second pass over 2 GB)
    const bool fmt1 = (hl[9] & ~8) == 0 && (hu[9] & ~8) == 0 && st_wx_on() && getenv("ILUPP_PACK_FMT0") == nullptr;
    hipLaunchKernelGGL(k_st_pack_pair, dim3((unsigned)(nwg * 4), (unsigned)((pl->max_chunks + 7) / 8)), dim3(512), 0, st, Lrow.ptr, Lrow.idx,
                       Lrow.val, Urow.ptr, Urow.idx, Urow.val, pl->ltab, pu->ltab, pu->uslot, pl->wtab, reinterpret_cast<v2d *>(pl->pk),
                       reinterpret_cast<v2d *>(pu->pk), pl->flags, fmt1 ? 1 : 0, bwd_desc ? 1 : 0);
    ILUPP_HIP(pool_malloc(&pl->ybuf, sizeof(double) * 64 * (size_t)(pl->nchunks + 4 * nwg)));
    ILUPP_HIP(pool_malloc(&pu->xlm, sizeof(double) * 64 * (size_t)(pu->nchunks + 4 * nwg)));
    pu->y_chunks = pl->nchunks + 4 * nwg;
    pl->xch_len = (int64_t)xtot[0][0] + xtot[0][1] + 64;
    pu->xch_len = (int64_t)xtot[1][0] + xtot[1][1] + 64;
    ILUPP_HIP(pool_malloc(&pl->xch, sizeof(double) * (size_t)pl->xch_len));
    ILUPP_HIP(pool_malloc(&pu->xch, sizeof(double) * (size_t)pu->xch_len));
    ILUPP_HIP(pool_malloc(&pu->ysrc, sizeof(int32_t) * (size_t)nslots));
    ILUPP_HIP(hipMemsetAsync(pu->ysrc, 0, sizeof(int32_t) * (size_t)nslots, st));
    hipLaunchKernelGGL(k_lm_ysrc, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st, nslots, pu->uslot, fwd.scount, pl->wtab, pl->skew, pu->ysrc);
    int32_t gl[4];
    ILUPP_HIP(d2h_async(st, gl, pl->flags, sizeof(gl)));
    ILUPP_HIP(stream_sync(st));
    if (dbg) fprintf(stderr, "[ilupp] static sweeps of a factor pair: row flags %d, %d+%d chunks\n", gl[0], hl[1], hu[1]);
    if (gl[0]) { pl->release(); pu->release(); return false; }
    pl->valid = pu->valid = true;
    pl->stat = pu->stat = true;
    pl->wx = (hl[9] & ~8) == 0; pu->wx = (hu[9] & ~8) == 0;
    pl->vec_ok = pl->wx && (hl[9] & 8) == 0; pu->vec_ok = pu->wx && (hu[9] & 8) == 0;
    pl->pair = pu->pair = true;
    pu->desc = bwd_desc;
    pu->linked = true;
    // a pair whose lanes fit the wave-exchange classes: class-aligned records, round 4's sweep kernels (neighbours in registers) with round
    // 5's vector wave; a backward sweep that accumulates in descending order (IChol0: T4) has its own instantiation (env ILUPP_NO_WR:
    // round 2's sweeps for every pair)
    if (fmt1) pl->fmt = pu->fmt = 1;
    if (pl->wx && pu->wx && st_wx_on()) wx_convert_records(st, pl, pu, 1);
    return true;
}

// ---------------------------------------------------------------------------------------------
// IChol(0), numeric phase on the static form.  L: on entry the lower triangle of A (rows, diagonal last), on exit the factor
// (values in place, pattern unchanged).  false = this matrix is not one for the static form (nothing has been touched).
// ---------------------------------------------------------------------------------------------
bool ichol0_numeric_st(hipStream_t st, DevMat *L, const Schedule &fwd, int32_t *d_ctrl, float *kernel_ms, int *rc_out)
{
    static const bool off = getenv("ILUPP_NO_PACKED") != nullptr || getenv("ILUPP_NO_STATIC") != nullptr;
    static const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    const int32_t n = L->n;
    *rc_out = ILUPP_OK;
    if (off || L->nnz > 4 * (int64_t)n || n < 16 || fwd.nslots < kThreads || !L->val || !fwd.exported) return false;
    const int nwg = fwd.nslots / kThreads;
    const int nslots = fwd.nslots;
    PackedSweep ps;
    st_structure(st, fwd, &ps, (int)SWEEP_FWD_LAST_ASC);
    hipLaunchKernelGGL((k_st_template<1>), dim3((unsigned)nwg), dim3(kThreads), 0, st, L->ptr, L->idx, fwd.B, fwd.nb, fwd.start,
                       fwd.blk2slot, fwd.sfirst, fwd.scount, fwd.exported, ps.ltab, ps.flags);
    hipLaunchKernelGGL((k_st_link<false>), dim3((unsigned)nwg), dim3(kThreads), 0, st, ps.ltab, static_cast<const int32_t *>(nullptr),
                       static_cast<const int32_t *>(nullptr), ps.skew, ps.wtab, ps.flags, 0);
    hipLaunchKernelGGL(k_st_chol_check, dim3((unsigned)nwg), dim3(kThreads), 0, st, ps.ltab, ps.flags);
    hipLaunchKernelGGL(k_st_scan, dim3(1), dim3(kThreads), 0, st, nwg * 4, ps.wtab, ps.flags);
    int32_t *xsz = nullptr;
    void *tmp2 = nullptr;
    size_t tb2 = 0;
    int32_t xtot[2] = {0, 0};
    ILUPP_HIP(pool_malloc(&xsz, sizeof(int32_t) * (size_t)nwg * 2));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, xsz, xsz + nwg, nwg, st));
    ILUPP_HIP(pool_malloc(&tmp2, tb2));
    ILUPP_HIP(pool_malloc(&ps.xe, sizeof(int32_t) * (size_t)nslots));
    ILUPP_HIP(pool_malloc(&ps.xw, sizeof(int32_t) * (size_t)nwg * 4));
    hipLaunchKernelGGL(k_st_xch_layout, dim3((unsigned)nwg), dim3(kThreads), 0, st, fwd.exported, ps.ltab, ps.wtab, ps.xe, ps.xw, xsz, ps.flags);
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp2, tb2, xsz, xsz + nwg, nwg, st));
    hipLaunchKernelGGL(k_st_xch_rows, dim3((unsigned)((nwg + 255) / 256)), dim3(256), 0, st, nwg, xsz + nwg, ps.xw);
    ILUPP_HIP(d2h_async(st, &xtot[0], xsz + nwg + (nwg - 1), sizeof(int32_t)));
    ILUPP_HIP(d2h_async(st, &xtot[1], xsz + (nwg - 1), sizeof(int32_t)));
    int32_t hl[4];
    ILUPP_HIP(d2h_async(st, hl, ps.flags, sizeof(hl)));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(xsz)); ILUPP_HIP(pool_free(tmp2));
    const int64_t lim = 6 * (int64_t)n + 64 * 4 * (int64_t)nwg;
    if (hl[0] || hl[1] <= 0 || (int64_t)hl[1] * 64 > lim || hl[1] + 4 * nwg >= kStMaxChunks ||
        (int64_t)nwg * kThreads * ((int64_t)hl[2] + 2 * kStXAlign) > 0x7fffffffLL) {
        if (dbg) fprintf(stderr, "[ilupp] IChol0 static form: structure rejected (flags %d, %d chunks)\n", hl[0], hl[1]);
        ps.release();
        return false;
    }
    ps.nchunks = hl[1]; ps.max_chunks = hl[2];
    ILUPP_HIP(pool_malloc(&ps.pk, (size_t)(ps.nchunks + 4 * nwg) * 2048));
    ps.built = true;
    hipLaunchKernelGGL(k_st_pack_lower, dim3((unsigned)(nwg * 4), (unsigned)((ps.max_chunks + 7) / 8)), dim3(512), 0, st, L->ptr, L->idx, L->val,
                       ps.ltab, ps.wtab, reinterpret_cast<v2d *>(ps.pk), ps.flags);
    ps.xch_len = (int64_t)xtot[0] + xtot[1] + 64;
    ILUPP_HIP(pool_malloc(&ps.xch, sizeof(double) * (size_t)ps.xch_len));
    fill_u64(st, reinterpret_cast<unsigned long long *>(ps.xch), ps.xch_len, kSentinel);
    int32_t gl[4];
    ILUPP_HIP(d2h_async(st, gl, ps.flags, sizeof(gl)));
    ILUPP_HIP(stream_sync(st));
    if (gl[0]) {
        if (dbg) fprintf(stderr, "[ilupp] IChol0 static form: row flags %d\n", gl[0]);
        ps.release();
        return false;
    }
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    StFArgs a;
    a.ltab = ps.ltab; a.wtab = ps.wtab;
    a.pkA = reinterpret_cast<const v2d *>(ps.pk);
    a.pkL = reinterpret_cast<v2d *>(ps.pk); a.pkU = nullptr;
    a.nchL = (int32_t)ps.nchunks;
    a.xe = ps.xe; a.xw = ps.xw; a.xch = ps.xch; a.ctrl = d_ctrl;
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    ILUPP_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_ichol0_st, dim3((unsigned)nwg), dim3(kStWgThreads), 0, st, a);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    ps.valid = true; ps.stat = true;
    st_unpack(st, *L, fwd, ps);
    int32_t ctrl[4] = {0, 0, 0, 0};
    ILUPP_HIP(d2h_async(st, ctrl, d_ctrl, 16));
    ILUPP_HIP(stream_sync(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    ILUPP_HIP(hipEventDestroy(e0));
    ILUPP_HIP(hipEventDestroy(e1));
    if (dbg) fprintf(stderr, "[ilupp] IChol0 static form: %d chunks, kernel %.3f ms, status %d\n", hl[1], kernel_ms ? *kernel_ms : 0.f, ctrl[1]);
    ps.release();
    if (ctrl[1] != 0) *rc_out = ILUPP_ERR_TIMEOUT;
    return true;
}

// ---------------------------------------------------------------------------------------------
// transposed apply on the static form: records of U^T (forward schedule) and L^T (backward schedule) from those of L and U.
// Row r of U^T has u(k, r) for the rows k = r + oF[j] of r's lower template, row r of L^T has l(k, r) for k = r + oB[q]; the
// entry sits in row k's record at the template position whose offset is the opposite one.  Every row is found through the
// forward schedule (block -> slot -> chunk); both outputs are stored in the forward order (as pkL / pkU are).
// count[0], count[1]: entries placed (the host compares them with the factors' entry counts: a pattern whose transposed
// rows do not fit the lanes' templates keeps the generic transposed solves).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
k_st_transpose(const int32_t *__restrict__ rtab, const int32_t *__restrict__ wtab, int32_t n, int32_t B, int32_t nb,
               const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot,
               const v2d *__restrict__ pkL, const v2d *__restrict__ pkU, v2d *__restrict__ pkUT, v2d *__restrict__ pkLT,
               unsigned long long *__restrict__ count)
{
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const v4i *R = reinterpret_cast<const v4i *>(rtab + (size_t)slot * 32);
    const v4i t0 = R[0], t1 = R[1], t2 = R[2];                           // first, cnt, skew, nL | oL x3, nU | oU x3, -
    const int k = tmin + c - t0.z;
    const bool live = c < nch && k >= 0 && k < t0.y;                     // (no early exit: the wave sums its counts below)
    const int r = t0.x + k;
    const double absent = st_dbl(kAbsent);
    const v2d *own = pkU + ((size_t)base + (live ? c : 0)) * 128 + L;
    double ut[3] = {absent, absent, absent}, lt[3] = {absent, absent, absent};
    const int oF[3] = {t1.x, t1.y, t1.z}, oB[3] = {t2.x, t2.y, t2.z};
    int nu = 0, nl = 0;
    // an entry only counts where the lane's template names the right producer for it (the rows k of R[3..6]); one that does
    // not is not placed, the totals then differ from the factors' and the host keeps the generic transposed solves
    const v4i klF = R[3], khF = R[4], klB = R[5], khB = R[6];
    const int klo[2][3] = {{klF.x, klF.y, klF.z}, {klB.x, klB.y, klB.z}}, khi[2][3] = {{khF.x, khF.y, khF.z}, {khB.x, khB.y, khB.z}};
    const int kside[2] = {k, t0.y - 1 - k};
#pragma unroll
    for (int side = 0; side < 2; ++side) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int o = side == 0 ? oF[j] : oB[j];
            if (!live || j >= (side == 0 ? t0.w : t1.w)) continue;
            const int kr = r + o;
            if (kr < 0 || kr >= n) continue;
            const int b = block_of(kr, B, nb, start);
            const int ps = blk2slot[b];
            const v4i *P = reinterpret_cast<const v4i *>(rtab + (size_t)ps * 32);
            const v4i p0 = P[0], p1 = P[1], p2 = P[2];
            const int kk = kr - p0.x;
            if (kk < 0 || kk >= p0.y) continue;
            const int pw = ps >> 6;
            const size_t at = ((size_t)wtab[(size_t)pw * 4] + (kk + p0.z - wtab[(size_t)pw * 4 + 1])) * 128 + (ps & 63);
            // the producer row's template position with the opposite offset: its upper side for U^T, its lower side for L^T
            const int po[3] = {side == 0 ? p2.x : p1.x, side == 0 ? p2.y : p1.y, side == 0 ? p2.z : p1.z};
            const int pn = side == 0 ? p1.w : p0.w;
            const v2d *src = (side == 0 ? pkU : pkL) + at;
            const v2d a = src[0], bq = src[64];
            const double pv[3] = {a.x, a.y, bq.x};
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < pn && po[q] == -o && st_bits(pv[q]) != kAbsent && kside[side] >= klo[side][j] && kside[side] < khi[side][j]) {
                    if (side == 0) { ut[j] = pv[q]; ++nu; } else { lt[j] = pv[q]; ++nl; }
                }
        }
    }
    if (live) {
    v2d x;
    v2d *pu_ = pkUT + ((size_t)base + c) * 128 + L;
    x.x = ut[0]; x.y = ut[1]; ST_STREAM_STORE(x, pu_);
    x.x = ut[2]; x.y = own[64].y; ST_STREAM_STORE(x, pu_ + 64);            // diagonal of U
    v2d *pl_ = pkLT + ((size_t)base + c) * 128 + L;
    x.x = lt[0]; x.y = lt[1]; ST_STREAM_STORE(x, pl_);
    x.x = lt[2]; x.y = 1.0; ST_STREAM_STORE(x, pl_ + 64);
    }
    // (per-wave partial sums: one atomic per wave and side)
    for (int off = 32; off > 0; off >>= 1) { nu += __shfl_xor(nu, off); nl += __shfl_xor(nl, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&count[0], (unsigned long long)nu); atomicAdd(&count[1], (unsigned long long)nl); }
}

void st_drop_transposed(PackedSweep *pl, PackedSweep *pu)
{
    if (pl->pkT) { (void)pool_free(pl->pkT); pl->pkT = nullptr; }
    if (pu->pkT) { (void)pool_free(pu->pkT); pu->pkT = nullptr; }
    pl->fmtT = pu->fmtT = 0;
}

bool st_build_transposed(hipStream_t st, const Schedule &fwd, int32_t n, const FactorLM &f, PackedSweep *pl, PackedSweep *pu,
                         int64_t offdiagL, int64_t offdiagU)
{
    const int32_t *rtab = f.xbase + (size_t)pl->nwg * kThreads;
    if (pl->pkT && pu->pkT) return true;
    st_drop_transposed(pl, pu);
    const size_t bytes = (size_t)(pl->nchunks + 4 * pl->nwg) * 2048;
    unsigned long long *cnt = nullptr;
    ILUPP_HIP(pool_malloc(&pl->pkT, bytes));
    ILUPP_HIP(pool_malloc(&pu->pkT, bytes));
    ILUPP_HIP(pool_malloc(&cnt, 16));
    ILUPP_HIP(hipMemsetAsync(cnt, 0, 16, st));
    const dim3 grid((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));
    hipLaunchKernelGGL(k_st_transpose, grid, dim3(512), 0, st, rtab, pl->wtab, n, fwd.B, fwd.nb,
                       fwd.start, fwd.blk2slot, reinterpret_cast<const v2d *>(pl->pk), reinterpret_cast<const v2d *>(pu->pk),
                       reinterpret_cast<v2d *>(pl->pkT), reinterpret_cast<v2d *>(pu->pkT), cnt);
    unsigned long long h[2] = {0, 0};
    ILUPP_HIP(d2h_async(st, h, cnt, 16));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(cnt));
    if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] static transposed records: %lld of %lld entries of U, %lld of %lld of L\n", (long long)h[0],
                                       (long long)offdiagU, (long long)h[1], (long long)offdiagL);
    if ((int64_t)h[0] != offdiagU || (int64_t)h[1] != offdiagL) { st_drop_transposed(pl, pu); return false; }
    return true;
}

// a sweep of the transposed apply: forward with U^T (pl's schedule and pkT), backward with L^T (pu's)
int sptrsv_st_T(hipStream_t st, const PackedSweep &ps, int32_t n, const double *rhs, double *out, int32_t *d_ticket, int32_t *d_err,
                double *lml, const int32_t *ysrc)
{
    const bool fwd = ps.kind == (int)SWEEP_FWD_LAST_ASC;
    if (!lml || !ps.pkT || (!fwd && (!ysrc || !ps.xlm))) { set_error("static transposed sweep without its records"); return ILUPP_ERR_INVALID; }
    if (ps.fmtT == 1) {
        // class-aligned transposed records: the wave-exchange sweeps (U^T: forward with a diagonal of its own; L^T: backward, unit
        // diagonal, descending accumulation) -- the object's own schedule, tables and exchange, the transposed records for `pk`
        PackedSweep q = ps;
        q.pk = ps.pkT; q.fmt = 1; q.pair = true; q.desc = !fwd; q.xch_armed = false;
        return sptrsv_wx(st, q, n, rhs, out, d_ticket, d_err, fwd ? lml : nullptr, fwd ? nullptr : lml, ysrc);
    }
    StSArgs a;
    a.pk = reinterpret_cast<const v2d *>(ps.pkT); a.ltab = ps.ltab; a.wtab = ps.wtab; a.n = n;
    a.nchY = (int32_t)ps.nchunks;
    a.xlm = lml; a.ylm = fwd ? nullptr : ps.xlm;
    a.ysrc = ysrc; a.xlm_chunks = (int32_t)(fwd ? ps.nchunks : ps.y_chunks);
    a.xe = ps.xe; a.xw = ps.xw; a.xch = ps.xch; a.ticket = d_ticket; a.err = d_err;
    fill_u64(st, reinterpret_cast<unsigned long long *>(ps.xch), ps.xch_len, kSentinel);
    const dim3 grid((unsigned)ps.nwg);
    st_solo_attr_T();
    if (fwd) {
        hipLaunchKernelGGL((k_st_vec<1, true>), dim3((unsigned)(ps.nwg * 4), (unsigned)st_vec_groups(ps.max_chunks)), dim3(512), 0, st, ps.ltab, ps.wtab,
                           const_cast<double *>(rhs), lml);
        hipLaunchKernelGGL((k_sptrsv_st<1, true>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
    } else {
        hipLaunchKernelGGL((k_sptrsv_st<-1, true>), grid, dim3(kStWgThreads), kStSoloLds, st, a);
        hipLaunchKernelGGL((k_st_vec<-1, false>), dim3((unsigned)(ps.nwg * 4), (unsigned)st_vec_groups(ps.max_chunks)), dim3(512), 0, st, ps.ltab, ps.wtab,
                           out, ps.xlm);
    }
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

#ifdef ST_STAMP
void st_read_stamps(unsigned long long *out) { ILUPP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_st_stamp), sizeof(unsigned long long) * 32)); }
void st_read_tl(unsigned long long *out) { ILUPP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_st_tl), sizeof(unsigned long long) * 4096 * 4)); }
#endif

}  // namespace ilupp

#ifdef ST_STAMP
extern "C" int ilupp_hip_debug_stamps(unsigned long long *out)
{
    try { ilupp::st_read_stamps(out); } catch (...) { return -1; }
    return 0;
}
extern "C" int ilupp_hip_debug_timeline(unsigned long long *out)
{
    try { ilupp::st_read_tl(out); } catch (...) { return -1; }
    return 0;
}
#endif
