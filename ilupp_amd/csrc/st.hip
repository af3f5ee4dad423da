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
//     and every elimination matches the eliminated row on its diagonal only ("simple" rows: all 5-/7-point stencils; then the
//     strictly-upper part of U is A's and the only recurrence is the one of the pivots).  What varies from row to row is
//     which template entries exist (domain boundaries): one mask byte per row, stored with the row's values of A;
//   * the consumer therefore has no record decode, no tags and no polling inside a workgroup: the lane table sits in
//     registers, finished unknowns / U rows go into LDS arrays indexed by (step mod 8, lane), and ONE s_barrier per step
//     orders them (4 waves per workgroup, nothing else resident: no loader waves, no importer wave);
//   * the streams are read by the consuming lane itself, kStD steps ahead, into a rotating register file (fully unrolled
//     loop), level-major and coalesced: factor kernel 64 B in (a0..a6, mask) and 2 x 32 B out per row, sweeps 32 B + rhs in;
//   * values of earlier workgroups (tile borders) are polled kStD steps ahead by the border lanes themselves
//     (write-through stores / cache-bypassing loads, the data is the flag); a value that is not there when its step
//     comes is polled again, which makes the tile fall back behind its producers by just the latency it needs.
//
// Everything the proof rejects runs on the record-decoding kernels (records_lm.hip) or the CSR kernels (any matrix).
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ilupp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

static constexpr int kStH = 8;                 // steps of hand-off history kept in LDS
#ifndef ST_D
#define ST_D 8
#endif
static constexpr int kStD = ST_D;              // steps the streams are read ahead (rotating registers)
#ifndef ST_P
#define ST_P 4
#endif
static constexpr int kStP = ST_P;              // steps ahead the values of earlier workgroups are polled (at most 2 such dependencies per lane)
static_assert(kStD % kStP == 0, "the poll ring is indexed by the unrolled step");
static constexpr int kStMaxSkew = 30000;
static constexpr unsigned kStSpinLimit = 1u << 21;

struct __attribute__((aligned(8))) D2s { double v[2]; };

// ---------------------------------------------------------------------------------------------
// analysis 1: lane templates.  TRI = +1: the entries left of the diagonal (forward schedule), -1: right (backward)
// ---------------------------------------------------------------------------------------------
template <int TRI>
__global__ void __launch_bounds__(kThreads)
k_st_template(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t B, int32_t nb,
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
    int32_t *T = ltab + (size_t)slot * kStTab;
    T[ST_FIRST] = first; T[ST_CNT] = cnt; T[ST_SKEW] = 0; T[ST_ND] = nd;
    T[ST_OFF] = o0; T[ST_OFF + 1] = o1; T[ST_OFF + 2] = o2;
    T[ST_SRC] = nd > 0 ? s0 : 0; T[ST_SRC + 1] = nd > 1 ? s1 : 0; T[ST_SRC + 2] = nd > 2 ? s2 : 0;
    T[ST_BLK] = b0; T[ST_BLK + 1] = b1; T[ST_BLK + 2] = b2;
    T[ST_KAP] = k0; T[ST_KAP + 1] = k1; T[ST_KAP + 2] = k2;
    T[ST_MPOS] = 0x3f; T[ST_DT] = 1; T[ST_DT + 1] = 1; T[ST_DT + 2] = 1;
    if (nd == 3 && (s0 & 3) == ST_GHOST && (s1 & 3) == ST_GHOST && (s2 & 3) == ST_GHOST) bad = 1;   // the poll ring holds two
    if (bad) atomicOr(&flags[0], 2);
}

// ---------------------------------------------------------------------------------------------
// analysis 2: skews (longest-path fixpoint over the workgroup's in-workgroup dependencies, now exact: the templates
// hold for every row), steps back of every in-workgroup dependency, chunk range of each wave; FWD: where each
// elimination's pivot row holds the entry that meets this row's diagonal, and the proof that it meets nothing else
// ---------------------------------------------------------------------------------------------
template <bool FWD>
__global__ void __launch_bounds__(kThreads)
k_st_link(int32_t *__restrict__ ltab, const int32_t *__restrict__ ltab_u, const int32_t *__restrict__ uslot,
          int32_t *__restrict__ skew, int32_t *__restrict__ wtab, int32_t *__restrict__ flags)
{
    __shared__ int s[kThreads];
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
        cd[j] = T[ST_KAP + j] + 1;
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
    int lo = cnt > 0 ? sk : 0x7fffffff, hi = cnt > 0 ? sk + cnt : -0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
    if ((t & 63) == 0) {
        int32_t *w = wtab + (size_t)(wg * 4 + (t >> 6)) * 4;
        const int nch = hi > lo ? hi - lo : 0;
        w[0] = 0; w[1] = nch > 0 ? lo : 0; w[2] = nch; w[3] = 0;
    }
    if (FWD && cnt > 0) {
        const int su = uslot[slot];
        int mpw = 0;
        if (su < 0) {
            bad = 1;
        } else {
            const int32_t *TU = ltab_u + (size_t)su * kStTab;
            const int nu = TU[ST_ND];
            const int mu[3] = {TU[ST_OFF], TU[ST_OFF + 1], TU[ST_OFF + 2]};
            const int ml[3] = {T[ST_OFF], T[ST_OFF + 1], T[ST_OFF + 2]};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int mp = 3;
                if (j < nd) {
                    const int sw = T[ST_SRC + j];
                    const int ps = sw >> 2;
                    const int pu = uslot[ps];
                    if (pu < 0) {
                        bad = 1;
                    } else {
                        const int32_t *TP = ltab_u + (size_t)pu * kStTab;
                        const int np = TP[ST_ND];
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            if (p < np) {
                                const int m = ml[j] + TP[ST_OFF + p];       // column of the pivot row's entry, relative to this row
                                if (m == 0) {
                                    mp = p;
                                } else {
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
                mpw |= mp << (2 * j);
            }
        }
        T[ST_MPOS] = mpw;
    }
    if (bad) atomicOr(&flags[0], 4);
}

// exclusive scan of the waves' chunk counts (one block); flags[1] = total, flags[2] = longest wave
__global__ void __launch_bounds__(kThreads)
k_st_scan(int32_t nwaves, int32_t *__restrict__ wtab, int32_t *__restrict__ flags)
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
// analysis 3: the proof, row by row, and the factor kernel's input: per (chunk, lane) 64 bytes
//   {a0,a1} {a2,a3} {a4,a5} {a6, mask}: the row of A by template position (a0..a2 left of the diagonal, a3 the
//   diagonal, a4..a6 right of it), mask bit j = position present (bit 3 = the row exists)
// A block owns the 64 lanes of one wave x 8 consecutive rows of each; inside a wave 8 lanes x 8 rows, so a load
// instruction touches 8 contiguous segments of A and a store instruction 8 neighbouring places of 8 chunks.
// VALUES: 0 = pattern only (masks), 1 = pattern and values, 2 = values only (masks exist: numeric re-factorisation)
// ---------------------------------------------------------------------------------------------
template <int VALUES>
__global__ void __launch_bounds__(512)
k_st_rows(const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval, int64_t nnz,
          const int32_t *__restrict__ ltabF, const int32_t *__restrict__ ltabB, const int32_t *__restrict__ uslot,
          const int32_t *__restrict__ wtab, const int32_t *__restrict__ startF, const int32_t *__restrict__ startB,
          v4i *__restrict__ pkA, int32_t *__restrict__ flags)
{
    const int w = blockIdx.x;
    const int L = (threadIdx.x >> 6) * 8 + (threadIdx.x & 7);
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const int32_t *T = ltabF + (size_t)slot * kStTab;
    const int cnt = T[ST_CNT];
    const int k = blockIdx.y * 8 + ((threadIdx.x >> 3) & 7);
    if (k >= cnt) return;
    const int r = T[ST_FIRST] + k;
    const int c = k + T[ST_SKEW] - wtab[(size_t)w * 4 + 1];
    v4i *p = pkA + ((size_t)wtab[(size_t)w * 4] + c) * 256 + L;
    const int a0 = Aptr[r];
    double v[8];
    if (VALUES != 0) {
        if ((int64_t)a0 + 8 <= nnz) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const D2s x = *reinterpret_cast<const D2s *>(Aval + a0 + 2 * i); v[2 * i] = x.v[0]; v[2 * i + 1] = x.v[1]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (int64_t)a0 + i < nnz ? Aval[a0 + i] : 0.0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.0;
    }
    double a[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int mask = 0;
    if (VALUES == 2) {
        mask = reinterpret_cast<const int *>(p + 192)[2];
        int pos = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (mask & (1 << j)) {
                a[j] = pos == 0 ? v[0] : pos == 1 ? v[1] : pos == 2 ? v[2] : pos == 3 ? v[3] : pos == 4 ? v[4] : pos == 5 ? v[5] : v[6];
                ++pos;
            }
        }
    } else {
        const int len = Aptr[r + 1] - a0;
        const int su = uslot[slot];
        int bad = (len > 7 || len < 1 || su < 0) ? 1 : 0;
        const int32_t *TB = ltabB + (size_t)(su < 0 ? 0 : su) * kStTab;
        const int ndF = T[ST_ND], ndB = TB[ST_ND];
        const int oF[3] = {T[ST_OFF], T[ST_OFF + 1], T[ST_OFF + 2]};
        const int oB[3] = {TB[ST_OFF], TB[ST_OFF + 1], TB[ST_OFF + 2]};
        const Row8 own = load_row8(Aidx, a0, len > 8 ? 8 : len, nnz);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const bool in = i < len;
            const int o = in ? own.c[i] - r : 0x40000000;
            bool hit = false;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool h = in && o < 0 && j < ndF && o == oF[j];
                if (h) { a[j] = v[i]; mask |= 1 << j; }
                hit |= h;
            }
            {
                const bool h = in && o == 0;
                if (h) { a[3] = v[i]; mask |= 8; }
                hit |= h;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bool h = in && o > 0 && q < ndB && o == oB[q];
                if (h) { a[4 + q] = v[i]; mask |= 16 << q; }
                hit |= h;
            }
            if (in && !hit) bad = 1;                            // a column outside the lane's template
        }
        if (!(mask & 8)) bad = 1;
        // every entry is produced where the template says: own chain, or the recorded block of the other lane
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (mask & (1 << j)) {
                const int col = r + oF[j];
                if ((T[ST_SRC + j] & 3) == ST_OWN) {
                    if (k == 0) bad = 1;
                } else {
                    const int b = T[ST_BLK + j];
                    if (col < startF[b] || col >= startF[b + 1]) bad = 1;
                }
            }
            if (mask & (16 << j)) {
                const int col = r + oB[j];
                if ((TB[ST_SRC + j] & 3) == ST_OWN) {
                    if (k == cnt - 1) bad = 1;
                } else {
                    const int b = TB[ST_BLK + j];
                    if (col < startB[b] || col >= startB[b + 1]) bad = 1;
                }
            }
        }
        if (bad) atomicOr(&flags[0], 8);
    }
    v2d x;
    if (VALUES != 0) {
        x.x = a[0]; x.y = a[1]; __builtin_nontemporal_store(x, reinterpret_cast<v2d *>(p));
        x.x = a[2]; x.y = a[3]; __builtin_nontemporal_store(x, reinterpret_cast<v2d *>(p) + 64);
        x.x = a[4]; x.y = a[5]; __builtin_nontemporal_store(x, reinterpret_cast<v2d *>(p) + 128);
    }
    v4i last; last.x = __double2loint(a[6]); last.y = __double2hiint(a[6]); last.z = mask; last.w = 0;
    __builtin_nontemporal_store(last, p + 192);
}

// ---------------------------------------------------------------------------------------------
// the barrier of a step: this wave's LDS writes of the previous step have landed, then everybody's have
// (NOT __syncthreads(): that would also drain the global loads in flight, i.e. the read-ahead)
// ---------------------------------------------------------------------------------------------
#define ST_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ unsigned long long st_bits(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double st_dbl(unsigned long long b) { return __longlong_as_double((long long)b); }
// a value about to be published must not look like one of the two markers
__device__ __forceinline__ double st_clean(double x)
{
    const unsigned long long b = st_bits(x);
    return (b == kSentinel || b == kAbsent) ? st_dbl(kCanonNaN) : x;
}

// ---------------------------------------------------------------------------------------------
// the factor kernel
// ---------------------------------------------------------------------------------------------
struct StFArgs {
    const int32_t *ltab, *wtab;               // forward schedule
    const v2d *pkA;                           // 4 x 64 x 16 B per chunk
    v2d *pkL, *pkU;                           // 2 x 64 x 16 B per chunk: {l0,l1}{l2,1} / {u1,u2}{u3,u0}
    const int32_t *wtabU, *skewU, *uslot;
    const int32_t *xbase;                     // per slot: first exchange row, -1 = nobody outside the workgroup reads it
    const long long *xcount;                  // doubles of xch in use
    double *xch;
    int32_t *ctrl;                            // [0] ticket, [1] error
};
static constexpr size_t kStFLds = (size_t)4 * kStH * kThreads * sizeof(double);

__global__ void __launch_bounds__(kThreads)
k_ilu0_st(StFArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *UH = reinterpret_cast<double *>(smem);               // [4][kStH][256]: u0 (pivot), u1, u2, u3 of finished rows
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = (unsigned)atomicAdd(&A.ctrl[0], 1);
    __syncthreads();
    const int wg = (int)s_ticket;
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int first = T[ST_FIRST], cnt = T[ST_CNT], sk = T[ST_SKEW], nd = T[ST_ND];
    (void)first;
    const long xrows = (long)(A.xcount[0] / 4);
    int ty[3], ix[3], dt[3], mp[3], gi[3];
    long gxr[2] = {0, 0};
    int gmo[2] = {1, 1};
    int ng = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int sw = T[ST_SRC + j];
        ty[j] = (j < nd && cnt > 0) ? (sw & 3) : ST_NONE;
        ix[j] = ty[j] == ST_LOCAL ? ((sw >> 2) & 255) : t;
        dt[j] = ty[j] == ST_LOCAL ? T[ST_DT + j] : 1;
        const int m = (T[ST_MPOS] >> (2 * j)) & 3;
        mp[j] = m;
        gi[j] = 0;
        if (ty[j] == ST_GHOST) {
            const long g = (long)A.xbase[sw >> 2] + T[ST_KAP + j];
            const int mo_ = 1 + (m < 3 ? m : 0);
            if (ng == 0) { gxr[0] = g; gmo[0] = mo_; } else { gxr[1] = g; gmo[1] = mo_; }
            gi[j] = ng < 1 ? 0 : 1;
            ++ng;
        }
    }
    const bool wave_ghost = __any(ng > 0);
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    int tlo = 0x7fffffff, thi = -0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t *w4 = A.wtab + (size_t)(wg * 4 + q) * 4;
        const int a = w4[1], b = w4[2];
        if (b > 0) { tlo = min(tlo, a); thi = max(thi, a + b); }
    }
    tlo = __builtin_amdgcn_readfirstlane(tlo); thi = __builtin_amdgcn_readfirstlane(thi);
    if (thi <= tlo) return;
    const int nsteps = ((thi - tlo + kStD - 1) / kStD) * kStD;
    // this lane's rows in the backward sweep's records: one chunk (128 sixteen-byte units) apart, descending
    long up0 = 0;
    if (cnt > 0) {
        const int su = A.uslot[slot];
        const int wu = su >> 6;
        up0 = ((long)A.wtabU[wu * 4] + (cnt - 1 + A.skewU[su] - A.wtabU[wu * 4 + 1])) * 128 + (su & 63);
    }
    const int xb = cnt > 0 ? A.xbase[slot] : -1;
    const bool exports = xb >= 0;
    const unsigned long long *xchb = reinterpret_cast<const unsigned long long *>(A.xch);
    const v2d *pa = A.pkA + (size_t)(nchw > 0 ? base : 0) * 256 + ln;
    v2d *lout = A.pkL + (size_t)base * 128 + ln;
    const int cmax = nchw > 0 ? nchw - 1 : 0;

    v2d ra[kStD][4];
    unsigned long long gq[kStP][2][2];            // [ring][dependency][pivot, matching entry]
    const int mo[3] = {1 + (mp[0] < 3 ? mp[0] : 0), 1 + (mp[1] < 3 ? mp[1] : 0), 1 + (mp[2] < 3 ? mp[2] : 0)};

#define STF_LOAD(u, tp)                                                                                              \
    do {                                                                                                             \
        int cw_ = (tp) - tminw; cw_ = cw_ < 0 ? 0 : (cw_ > cmax ? cmax : cw_);                                      \
        const v2d *q_ = pa + (size_t)cw_ * 256;                                                                      \
        ra[u][0] = __builtin_nontemporal_load(q_); ra[u][1] = __builtin_nontemporal_load(q_ + 64);                   \
        ra[u][2] = __builtin_nontemporal_load(q_ + 128); ra[u][3] = __builtin_nontemporal_load(q_ + 192);            \
    } while (0)
#define STF_POLL(g, tp)                                                                                              \
    do {                                                                                                             \
        if (wave_ghost) {                                                                                            \
            int kp_ = (tp) - sk; kp_ = kp_ < 0 ? 0 : (kp_ >= cnt ? cnt - 1 : kp_);                                   \
            _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                                                       \
                if (e_ < ng) {                                                                                       \
                    long row_ = gxr[e_] + kp_; row_ = row_ < 0 ? 0 : (row_ >= xrows ? xrows - 1 : row_);             \
                    gq[g][e_][0] = ld_agent_u64(xchb + row_ * 4);                                                    \
                    gq[g][e_][1] = ld_agent_u64(xchb + row_ * 4 + gmo[e_]);                                          \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)

#pragma unroll
    for (int g = 0; g < kStP; ++g) {
        gq[g][0][0] = kSentinel; gq[g][0][1] = kSentinel; gq[g][1][0] = kSentinel; gq[g][1][1] = kSentinel;
        STF_POLL(g, tlo + g);
        asm volatile("" ::: "memory");
    }
    // in step order (a compiler barrier after each): the waits of the loop are derived from the oldest position a
    // register's load can have, and the scheduler is free to turn an unordered prologue upside down
#pragma unroll
    for (int u = 0; u < kStD; ++u) { STF_LOAD(u, tlo + u); asm volatile("" ::: "memory"); }
    double pu0 = 0.0, pu1 = 0.0, pu2 = 0.0, pu3 = 0.0;          // U row of the lane's previous row
    bool dead = false;

    for (int tb = 0; tb < nsteps; tb += kStD) {
#pragma unroll
        for (int u = 0; u < kStD; ++u) {
            const int tau = tlo + tb + u;
            const int k = tau - sk;
            const bool valid = (unsigned)k < (unsigned)cnt;
            const v2d r0 = ra[u][0], r1 = ra[u][1], r2 = ra[u][2], r3 = ra[u][3];
            // (both halves of the mask word are used on purpose: a dead quarter of a 16-byte load is a free register to
            // the allocator, and a temporary placed there has to wait for that load -- a load of a LATER step)
            const int mask = valid ? ((int)(unsigned)st_bits(r3.y) | (int)(unsigned)(st_bits(r3.y) >> 32)) : 0;
            ST_BARRIER();
            // in-workgroup hand-off: pivot and the one matching entry of each eliminated row, kStH steps of history
            double piv[3], um[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int at = ((tau - dt[j]) & (kStH - 1)) * kThreads + ix[j];
                piv[j] = UH[at];
                um[j] = UH[mo[j] * kStH * kThreads + at];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (ty[j] == ST_OWN) { piv[j] = pu0; um[j] = mo[j] == 1 ? pu1 : (mo[j] == 2 ? pu2 : pu3); }
                if (ty[j] == ST_GHOST) {
                    piv[j] = st_dbl(gi[j] == 0 ? gq[u % kStP][0][0] : gq[u % kStP][1][0]);
                    um[j] = st_dbl(gi[j] == 0 ? gq[u % kStP][0][1] : gq[u % kStP][1][1]);
                }
            }
            if (wave_ghost && !dead) {
                // rows of earlier workgroups that had not arrived when they were asked for: ask again
                unsigned spins = 0;
                for (;;) {
                    bool miss = false;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (ty[j] == ST_GHOST && (mask & (1 << j)) && (st_bits(piv[j]) == kSentinel || st_bits(um[j]) == kSentinel)) miss = true;
                    if (!__any(miss)) break;
                    if (miss) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            if (ty[j] == ST_GHOST && (mask & (1 << j))) {
                                long row = (gi[j] == 0 ? gxr[0] : gxr[1]) + k; row = row < 0 ? 0 : (row >= xrows ? xrows - 1 : row);
                                piv[j] = st_dbl(ld_agent_u64(xchb + row * 4));
                                um[j] = st_dbl(ld_agent_u64(xchb + row * 4 + mo[j]));
                            }
                        }
                    }
                    // retired HERE: a load pending at the join below would make hipcc wait for vmcnt(0) -- the whole
                    // read-ahead -- on every step, also on those that never come through this loop
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        if (spins > kStSpinLimit) atomicExch(&A.ctrl[1], 1);
                        const int e = ld_agent_i32(&A.ctrl[1]);
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        if (spins > kStSpinLimit || e != 0) { dead = true; break; }
                    }
                }
            }
            double w0 = r0.x, w1 = r0.y, w2 = r1.x, w3 = r1.y;
            const double w4 = r2.x, w5 = r2.y, w6 = r3.x;
            // eliminations in ascending column; each meets this row on the diagonal only (ILU0.hpp:8-23, :47-62)
            if (mask & 1) { const double l = w0 / piv[0]; if (mp[0] < 3 && st_bits(um[0]) != kAbsent) { const double pr = l * um[0]; w3 = w3 - pr; } w0 = l; }
            if (mask & 2) { const double l = w1 / piv[1]; if (mp[1] < 3 && st_bits(um[1]) != kAbsent) { const double pr = l * um[1]; w3 = w3 - pr; } w1 = l; }
            if (mask & 4) { const double l = w2 / piv[2]; if (mp[2] < 3 && st_bits(um[2]) != kAbsent) { const double pr = l * um[2]; w3 = w3 - pr; } w2 = l; }
            const double absent = st_dbl(kAbsent);
            const double u0 = st_clean(w3);
            const double u1 = (mask & 16) ? st_clean(w4) : absent;
            const double u2 = (mask & 32) ? st_clean(w5) : absent;
            const double u3 = (mask & 64) ? st_clean(w6) : absent;
            {
                const int at = (tau & (kStH - 1)) * kThreads + t;
                UH[at] = u0; UH[kStH * kThreads + at] = u1; UH[2 * kStH * kThreads + at] = u2; UH[3 * kStH * kThreads + at] = u3;
            }
            if (exports && valid) {
                double *xr = A.xch + ((size_t)xb + k) * 4;
                st_agent_f64(xr, u0); st_agent_f64(xr + 1, u1); st_agent_f64(xr + 2, u2); st_agent_f64(xr + 3, u3);
            }
            const int cw = tau - tminw;
#ifndef EXP_ST_NOSTORE
            if ((unsigned)cw < (unsigned)nchw) {
                v2d la, lb;
                la.x = (mask & 1) ? st_clean(w0) : absent; la.y = (mask & 2) ? st_clean(w1) : absent;
                lb.x = (mask & 4) ? st_clean(w2) : absent; lb.y = 1.0;
                v2d *o = lout + (size_t)cw * 128;
                __builtin_nontemporal_store(la, o); __builtin_nontemporal_store(lb, o + 64);
            }
            if (valid) {
                v2d ua, ub;
                ua.x = u1; ua.y = u2; ub.x = u3; ub.y = u0;
                v2d *o = A.pkU + (up0 - 128 * (long)k);
                __builtin_nontemporal_store(ua, o); __builtin_nontemporal_store(ub, o + 64);
            }
#endif
            if (valid) { pu0 = u0; pu1 = u1; pu2 = u2; pu3 = u3; }
            STF_LOAD(u, tau + kStD);
            STF_POLL(u % kStP, tau + kStP);
        }
    }
#undef STF_LOAD
#undef STF_POLL
    if (dead && ln == 0) atomicExch(&A.ctrl[1], 1);
}

// ---------------------------------------------------------------------------------------------
// the sweeps.  DR = +1 forward (rows ascending), -1 backward
// ---------------------------------------------------------------------------------------------
struct StSArgs {
    const v2d *pk;                            // 2 x 64 x 16 B per chunk: {v0,v1}{v2,vdiag}, dependencies in accumulation order
    const int32_t *ltab, *wtab;
    int32_t n;
    const double *rhs;
    double *out;
    const int32_t *exported;
    double *ypk_out;                          // the unknowns level-major, 64 per chunk (then only exported lanes write `out`)
    const double *ypk_in;                     // right-hand side from such a vector written by the opposite sweep
    const int32_t *ysrc;
    int32_t *ticket, *err;
};

template <int DR>
__global__ void __launch_bounds__(kThreads)
k_sptrsv_st(StSArgs A)
{
    __shared__ double XH[kStH * kThreads];
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = (unsigned)atomicAdd(A.ticket, 1);
    __syncthreads();
    const int wg = (int)s_ticket;
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const int slot = wg * kThreads + t;
    const int32_t *T = A.ltab + (size_t)slot * kStTab;
    const int first = T[ST_FIRST], cnt = T[ST_CNT], sk = T[ST_SKEW], nd = T[ST_ND];
    int ty[3], ix[3], dt[3], goff[3], gi[3];
    int go2[2] = {0, 0};
    int ng = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int sw = T[ST_SRC + j];
        ty[j] = (j < nd && cnt > 0) ? (sw & 3) : ST_NONE;
        ix[j] = ty[j] == ST_LOCAL ? ((sw >> 2) & 255) : t;
        dt[j] = ty[j] == ST_LOCAL ? T[ST_DT + j] : 1;
        goff[j] = T[ST_OFF + j];
        gi[j] = 0;
        if (ty[j] == ST_GHOST) {
            if (ng == 0) go2[0] = goff[j]; else go2[1] = goff[j];
            gi[j] = ng < 1 ? 0 : 1;
            ++ng;
        }
    }
    const bool wave_ghost = __any(ng > 0);
    const int32_t *wt = A.wtab + (size_t)(wg * 4 + wv) * 4;
    const int base = __builtin_amdgcn_readfirstlane(wt[0]), tminw = __builtin_amdgcn_readfirstlane(wt[1]),
              nchw = __builtin_amdgcn_readfirstlane(wt[2]);
    int tlo = 0x7fffffff, thi = -0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t *w4 = A.wtab + (size_t)(wg * 4 + q) * 4;
        const int a = w4[1], b = w4[2];
        if (b > 0) { tlo = min(tlo, a); thi = max(thi, a + b); }
    }
    tlo = __builtin_amdgcn_readfirstlane(tlo); thi = __builtin_amdgcn_readfirstlane(thi);
    if (thi <= tlo) return;
    const int nsteps = ((thi - tlo + kStD - 1) / kStD) * kStD;
    const bool exports = cnt > 0 && A.exported[slot] != 0;
    const long ysrc0 = (A.ypk_in && cnt > 0) ? (long)A.ysrc[slot] : 0;
    const unsigned long long *outb = reinterpret_cast<const unsigned long long *>(A.out);
    const v2d *pr = A.pk + (size_t)(nchw > 0 ? base : 0) * 128 + ln;
    double *yo = A.ypk_out ? A.ypk_out + (size_t)base * 64 + ln : nullptr;
    const int cmax = nchw > 0 ? nchw - 1 : 0;
    const int n = A.n;

    v2d ra[kStD][2];
    double rr[kStD];
    unsigned long long gq[kStP][2];

#define STS_LOAD(u, tp)                                                                                              \
    do {                                                                                                             \
        int cw_ = (tp) - tminw; cw_ = cw_ < 0 ? 0 : (cw_ > cmax ? cmax : cw_);                                      \
        const v2d *q_ = pr + (size_t)cw_ * 128;                                                                      \
        ra[u][0] = __builtin_nontemporal_load(q_); ra[u][1] = __builtin_nontemporal_load(q_ + 64);                   \
        const int kq_ = (tp) - sk;                                                                                   \
        const bool in_ = (unsigned)kq_ < (unsigned)cnt;                                                              \
        int row_ = first + DR * (in_ ? kq_ : 0); row_ = row_ < 0 ? 0 : (row_ >= n ? n - 1 : row_);                   \
        rr[u] = A.ypk_in ? A.ypk_in[in_ ? (size_t)(ysrc0 - 64 * (long)kq_) : 0] : A.rhs[row_];                       \
    } while (0)
#define STS_POLL(g, tp)                                                                                              \
    do {                                                                                                             \
        if (wave_ghost) {                                                                                            \
            const int kq_ = (tp) - sk;                                                                               \
            const bool in_ = (unsigned)kq_ < (unsigned)cnt;                                                          \
            const int row_ = first + DR * (in_ ? kq_ : 0);                                                           \
            _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                                                       \
                if (e_ < ng) {                                                                                       \
                    int c_ = row_ + go2[e_]; c_ = c_ < 0 ? 0 : (c_ >= n ? n - 1 : c_);                               \
                    gq[g][e_] = ld_agent_u64(outb + c_);                                                             \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)

#pragma unroll
    for (int g = 0; g < kStP; ++g) {
        gq[g][0] = kSentinel; gq[g][1] = kSentinel;
        STS_POLL(g, tlo + g);
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int u = 0; u < kStD; ++u) { STS_LOAD(u, tlo + u); asm volatile("" ::: "memory"); }
    double prev = 0.0;
    bool dead = false;

    for (int tb = 0; tb < nsteps; tb += kStD) {
#pragma unroll
        for (int u = 0; u < kStD; ++u) {
            const int tau = tlo + tb + u;
            const int k = tau - sk;
            const bool valid = (unsigned)k < (unsigned)cnt;
            const int r = first + DR * k;
            const v2d va = ra[u][0], vb = ra[u][1];
            const double v[3] = {va.x, va.y, vb.x};
            bool pres[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) pres[j] = valid && ty[j] != ST_NONE && st_bits(v[j]) != kAbsent;
            ST_BARRIER();
            double xs[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) xs[j] = XH[((tau - dt[j]) & (kStH - 1)) * kThreads + ix[j]];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (ty[j] == ST_OWN) xs[j] = prev;
                if (ty[j] == ST_GHOST) xs[j] = st_dbl(gi[j] == 0 ? gq[u % kStP][0] : gq[u % kStP][1]);
            }
            if (wave_ghost && !dead) {
                unsigned spins = 0;
                for (;;) {
                    bool miss = false;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (ty[j] == ST_GHOST && pres[j] && st_bits(xs[j]) == kSentinel) miss = true;
                    if (!__any(miss)) break;
                    if (miss) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            if (ty[j] == ST_GHOST && pres[j]) {
                                int c = r + goff[j]; c = c < 0 ? 0 : (c >= n ? n - 1 : c);
                                xs[j] = st_dbl(ld_agent_u64(outb + c));
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);      // retired here, not at the join (see k_ilu0_st)
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        if (spins > kStSpinLimit) atomicExch(A.err, 1);
                        const int e = ld_agent_i32(A.err);
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        if (spins > kStSpinLimit || e != 0) { dead = true; break; }
                    }
                }
            }
            // sequential accumulation in stored order, division by the diagonal even when it is 1 (a quotient by 1.0 is
            // the dividend, bit for bit: the L sweep skips the divider)
            double acc = rr[u];
            if (pres[0]) { const double p = v[0] * xs[0]; acc = acc - p; }
            if (pres[1]) { const double p = v[1] * xs[1]; acc = acc - p; }
            if (pres[2]) { const double p = v[2] * xs[2]; acc = acc - p; }
            double x = acc;
            if (__any(valid && vb.y != 1.0)) x = acc / vb.y;
            if (x != x) x = st_dbl(kCanonNaN);
            XH[(tau & (kStH - 1)) * kThreads + t] = x;
            const int cw = tau - tminw;
            if (yo) {
                if ((unsigned)cw < (unsigned)nchw) __builtin_nontemporal_store(x, yo + (size_t)cw * 64);
                if (exports && valid) st_agent_f64(A.out + r, x);
            } else if (valid) {
                if (exports) st_agent_f64(A.out + r, x); else A.out[r] = x;
            }
            if (valid) prev = x;
            STS_LOAD(u, tau + kStD);
            STS_POLL(u % kStP, tau + kStP);
        }
    }
#undef STS_LOAD
#undef STS_POLL
    if (dead && ln == 0) atomicExch(A.err, 1);
}

// ---------------------------------------------------------------------------------------------
// CSR values on demand
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(512)
k_st_unpack(const int32_t *__restrict__ ptr, double *__restrict__ val, const int32_t *__restrict__ wtab,
            const int32_t *__restrict__ ltab, const v2d *__restrict__ pk)
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
    const int r = T[ST_FIRST] + DR * k;
    const int q0 = ptr[r], q1 = ptr[r + 1];
    const v2d *p = pk + ((size_t)base + c) * 128 + L;
    const v2d a = p[0], b = p[64];
    const double v[3] = {a.x, a.y, b.x};
    val[FWD ? q1 - 1 : q0] = b.y;
    int q = FWD ? q0 : q0 + 1;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (st_bits(v[j]) != kAbsent && q < (FWD ? q1 - 1 : q1)) val[q++] = v[j];
}

void st_unpack(hipStream_t st, const DevMat &M, const Schedule &sch, const PackedSweep &ps)
{
    (void)sch;
    const dim3 grid((unsigned)(ps.nwg * 4), (unsigned)((ps.max_chunks + 7) / 8));
    if ((SweepKind)ps.kind == SWEEP_FWD_LAST_ASC)
        hipLaunchKernelGGL((k_st_unpack<SWEEP_FWD_LAST_ASC>), grid, dim3(512), 0, st, M.ptr, M.val, ps.wtab, ps.ltab,
                           reinterpret_cast<const v2d *>(ps.pk));
    else
        hipLaunchKernelGGL((k_st_unpack<SWEEP_BWD_FIRST_ASC>), grid, dim3(512), 0, st, M.ptr, M.val, ps.wtab, ps.ltab,
                           reinterpret_cast<const v2d *>(ps.pk));
    ILUPP_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
__global__ void k_st_xrows(int32_t nslots, const int32_t *__restrict__ exported, const int32_t *__restrict__ scount,
                           int32_t *__restrict__ rows)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nslots) rows[s] = exported[s] ? scount[s] : 0;
}
__global__ void k_st_xbase(int32_t nslots, const int32_t *__restrict__ exported, const int32_t *__restrict__ scount,
                           int32_t *__restrict__ xbase, long long *__restrict__ xcount)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int b = xbase[s];
    if (s == nslots - 1) *xcount = ((long long)b + (exported[s] ? scount[s] : 0)) * 4;
    if (!exported[s]) xbase[s] = -1;
}
__global__ void k_st_fill(unsigned long long *__restrict__ p, const long long *__restrict__ count, unsigned long long v)
{
    const long long n = *count;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

static void st_structure(hipStream_t st, const Schedule &sch, PackedSweep *ps, int kind)
{
    ps->nwg = sch.nslots / kThreads;
    ps->kind = kind;
    ILUPP_HIP(pool_malloc(&ps->ltab, sizeof(int32_t) * kStTab * (size_t)sch.nslots));
    ILUPP_HIP(pool_malloc(&ps->skew, sizeof(int32_t) * (size_t)sch.nslots));
    ILUPP_HIP(pool_malloc(&ps->wtab, sizeof(int32_t) * 16 * (size_t)ps->nwg));
    ILUPP_HIP(pool_malloc(&ps->flags, 64));
    ILUPP_HIP(hipMemsetAsync(ps->flags, 0, 64, st));
}

// The whole static analysis of an ILU(0): true when the factor kernel and both sweeps can run from lane tables
// (pl, pu, f then complete, the values of A packed); false leaves the three objects released.
bool st_analyse_ilu0(hipStream_t st, const DevMat &A, const Schedule &fwd, const Schedule &bwd, PackedSweep *pl,
                     PackedSweep *pu, FactorLM *f)
{
    pl->release(); pu->release(); f->release();
    static const bool off = getenv("ILUPP_NO_PACKED") != nullptr || getenv("ILUPP_NO_PACKED_FACTOR") != nullptr ||
                            getenv("ILUPP_CLASSIC_ANALYSIS") != nullptr || getenv("ILUPP_NO_STATIC") != nullptr;
    static const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    if (off || A.nnz > 7 * (int64_t)A.n || A.nnz < 16 || fwd.nslots < kThreads || fwd.nslots != bwd.nslots || !A.val) return false;
    const int nwg = fwd.nslots / kThreads;
    st_structure(st, fwd, pl, (int)SWEEP_FWD_LAST_ASC);
    st_structure(st, bwd, pu, (int)SWEEP_BWD_FIRST_ASC);
    hipLaunchKernelGGL((k_st_template<1>), dim3((unsigned)nwg), dim3(kThreads), 0, st, A.ptr, A.idx, fwd.B, fwd.nb, fwd.start,
                       fwd.blk2slot, fwd.sfirst, fwd.scount, fwd.exported, pl->ltab, pl->flags);
    hipLaunchKernelGGL((k_st_template<-1>), dim3((unsigned)nwg), dim3(kThreads), 0, st, A.ptr, A.idx, bwd.B, bwd.nb, bwd.start,
                       bwd.blk2slot, bwd.sfirst, bwd.scount, bwd.exported, pu->ltab, pu->flags);
    pu->built = true;
    lm_link_factor(st, fwd, bwd, pu);                   // forward slot -> backward slot of the same chain (flags[3] when there is none)
    hipLaunchKernelGGL((k_st_link<true>), dim3((unsigned)nwg), dim3(kThreads), 0, st, pl->ltab, pu->ltab, pu->uslot, pl->skew,
                       pl->wtab, pl->flags);
    hipLaunchKernelGGL((k_st_link<false>), dim3((unsigned)nwg), dim3(kThreads), 0, st, pu->ltab, static_cast<const int32_t *>(nullptr),
                       static_cast<const int32_t *>(nullptr), pu->skew, pu->wtab, pu->flags);
    hipLaunchKernelGGL(k_st_scan, dim3(1), dim3(kThreads), 0, st, nwg * 4, pl->wtab, pl->flags);
    hipLaunchKernelGGL(k_st_scan, dim3(1), dim3(kThreads), 0, st, nwg * 4, pu->wtab, pu->flags);
    int32_t hl[4], hu[4];
    ILUPP_HIP(d2h_async(st, hl, pl->flags, sizeof(hl)));
    ILUPP_HIP(d2h_async(st, hu, pu->flags, sizeof(hu)));
    ILUPP_HIP(stream_sync(st));
    const int64_t lim = 2 * (int64_t)A.n + 64 * 4 * (int64_t)nwg;
    if (hl[0] || hu[0] || hu[3] || hl[1] <= 0 || hu[1] <= 0 || (int64_t)hl[1] * 64 > lim || (int64_t)hu[1] * 64 > lim) {
        if (dbg) fprintf(stderr, "[ilupp] static analysis: structure rejected (flags %d %d link %d, %d %d chunks)\n", hl[0], hu[0], hu[3], hl[1], hu[1]);
        pl->release(); pu->release();
        return false;
    }
    pl->nchunks = hl[1]; pl->max_chunks = hl[2];
    pu->nchunks = hu[1]; pu->max_chunks = hu[2];
    ILUPP_HIP(pool_malloc(&pl->pk, (size_t)pl->nchunks * 2048));
    ILUPP_HIP(pool_malloc(&pu->pk, (size_t)pu->nchunks * 2048));
    ILUPP_HIP(pool_malloc(&f->pkA, (size_t)pl->nchunks * 4096));
    pl->built = true;
    {
        const dim3 grid((unsigned)(nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));        // every lane has at most max_chunks rows
        hipLaunchKernelGGL((k_st_rows<1>), grid, dim3(512), 0, st, A.ptr, A.idx, A.val, (int64_t)A.nnz, pl->ltab, pu->ltab, pu->uslot,
                           pl->wtab, fwd.start, bwd.start, reinterpret_cast<v4i *>(f->pkA), pl->flags);
    }
    lm_link_y(st, fwd, pl, pu);
    // exchange rows of the forward lanes that other workgroups read
    const int nslots = fwd.nslots;
    ILUPP_HIP(pool_malloc(&f->xbase, sizeof(int32_t) * (size_t)nslots));
    ILUPP_HIP(pool_malloc(&f->xcount, 64));
    int32_t *rows = nullptr;
    ILUPP_HIP(pool_malloc(&rows, sizeof(int32_t) * (size_t)nslots));
    const unsigned gb = (unsigned)((nslots + 255) / 256);
    hipLaunchKernelGGL(k_st_xrows, dim3(gb), dim3(256), 0, st, nslots, fwd.exported, fwd.scount, rows);
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, rows, f->xbase, nslots, st));
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&tmp, tb));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, rows, f->xbase, nslots, st));
    hipLaunchKernelGGL(k_st_xbase, dim3(gb), dim3(256), 0, st, nslots, fwd.exported, fwd.scount, f->xbase, f->xcount);
    ILUPP_HIP(pool_malloc(&f->xch, sizeof(double) * 4 * (size_t)A.n + 64));
    int32_t gl[4];
    ILUPP_HIP(d2h_async(st, gl, pl->flags, sizeof(gl)));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(rows));
    ILUPP_HIP(pool_free(tmp));
    if (dbg) fprintf(stderr, "[ilupp] static analysis: row flags %d, %d+%d chunks\n", gl[0], hl[1], hu[1]);
    if (gl[0]) { pl->release(); pu->release(); f->release(); return false; }
    pl->valid = pu->valid = true;
    pl->stat = pu->stat = true;
    pu->linked = true;
    f->built = true;
    f->stat = true;
    f->values_packed = true;
    return true;
}

int ilu0_numeric_st(hipStream_t st, const DevMat &A, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu, FactorLM *f,
                    int32_t *d_ctrl, float *kernel_ms, hipEvent_t e0, hipEvent_t e1)
{
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    hipLaunchKernelGGL(k_st_fill, dim3(512), dim3(256), 0, st, reinterpret_cast<unsigned long long *>(f->xch),
                       reinterpret_cast<const long long *>(f->xcount), kSentinel);
    if (!f->values_packed) {
        const dim3 grid((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));
        hipLaunchKernelGGL((k_st_rows<2>), grid, dim3(512), 0, st, A.ptr, A.idx, A.val, (int64_t)A.nnz, pl->ltab, pu->ltab, pu->uslot,
                           pl->wtab, fwd.start, fwd.start, reinterpret_cast<v4i *>(f->pkA), pl->flags);
    }
    f->values_packed = false;
    StFArgs a;
    a.ltab = pl->ltab; a.wtab = pl->wtab;
    a.pkA = reinterpret_cast<const v2d *>(f->pkA);
    a.pkL = reinterpret_cast<v2d *>(pl->pk); a.pkU = reinterpret_cast<v2d *>(pu->pk);
    a.wtabU = pu->wtab; a.skewU = pu->skew; a.uslot = pu->uslot;
    a.xbase = f->xbase; a.xcount = f->xcount; a.xch = f->xch; a.ctrl = d_ctrl;
    ILUPP_HIP(hipEventRecord(e0, st));
    static std::once_flag once[16];
    int dev = 0;
    ILUPP_HIP(hipGetDevice(&dev));
    std::call_once(once[dev & 15], [] {
        ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_st, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStFLds));
    });
    hipLaunchKernelGGL(k_ilu0_st, dim3((unsigned)pl->nwg), dim3(kThreads), kStFLds, st, a);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4];
    ILUPP_HIP(d2h_async(st, ctrl, d_ctrl, 16));
    ILUPP_HIP(stream_sync(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

int sptrsv_st(hipStream_t st, const PackedSweep &ps, const Schedule &sch, int32_t n, const double *rhs, double *out,
              int32_t *d_ticket, int32_t *d_err, double *ypk_out, const double *ypk_in, const int32_t *ysrc)
{
    StSArgs a;
    a.pk = reinterpret_cast<const v2d *>(ps.pk); a.ltab = ps.ltab; a.wtab = ps.wtab; a.n = n; a.rhs = rhs; a.out = out;
    a.exported = sch.exported; a.ypk_out = ypk_out; a.ypk_in = ypk_in; a.ysrc = ysrc; a.ticket = d_ticket; a.err = d_err;
    if (ps.kind == (int)SWEEP_FWD_LAST_ASC)
        hipLaunchKernelGGL((k_sptrsv_st<1>), dim3((unsigned)ps.nwg), dim3(kThreads), 0, st, a);
    else
        hipLaunchKernelGGL((k_sptrsv_st<-1>), dim3((unsigned)ps.nwg), dim3(kThreads), 0, st, a);
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

}  // namespace ilupp
