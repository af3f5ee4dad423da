// ilupp_amd/csrc/sptrsv_lvl.hip -- triangular sweeps with one lane per row, rows in LEVEL order (gfx950 only).
//
// For factors whose rows are too long for the level-major records (ILUT / ILUC factors, ICholT with fill, ILU(0) of 9- and
// 27-point stencils).  k_sptrsv_rows (sptrsv.hip) walks the rows in natural order: a workgroup's ticket gives it 1024
// consecutive rows, and a row can only run once the rows it reads have; on a mesh that means only the rows of the few grid
// lines inside the window of resident tickets are ever runnable (9-point 4096 x 4096: 128 of 4096 lines; one apply 293 ms,
// more than the reference needs on one core).  Here the rows are renumbered once per factor:
//   level(r) = 0 for a row without off-diagonal entries, else 1 + max level(c) over its off-diagonal columns c
//              (k_lvl_levels: the same dataflow walk as a sweep, integers instead of unknowns),
//   position = rank of (level, r) (stable radix sort), column indices rewritten to positions,
//   every row stored as [off-diagonal entries in the order the reference applies them ..., diagonal]
// so that one kernel serves the three sweep kinds, every row only reads rows of earlier tickets (or earlier lanes of its
// own workgroup: LDS), and the rows that become runnable together sit next to each other.  The arithmetic per row is the
// reference's (sequential accumulation in stored order, one division by the diagonal: matrix_sparse::triangular_solve,
// sparse.hpp:4040-4075), so the result has the same bits as k_sptrsv_rows'.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace ilupp {

#ifndef LV_BLOCK
#define LV_BLOCK 64
#endif
static constexpr int kLvBlock = LV_BLOCK;       // lanes of a workgroup of the level pass
static constexpr int kLvRing = 16;              // levels a lane of the level pass keeps in LDS for its neighbours (a power of two)
static constexpr unsigned kLvSpinLimit = 1u << 22;

void LevelSweep::release()
{
    for (void *q : {(void *)ptr, (void *)idx, (void *)perm, (void *)val, (void *)xp}) if (q) (void)pool_free(q);
    ptr = idx = perm = nullptr; val = xp = nullptr;
    valid = false; tried = false; nlevels = 0;
}

// ---------------------------------------------------------------------------------------------
// levels: natural order, one lane per row, data-is-flag on lev[] (-1 = not yet)
// ---------------------------------------------------------------------------------------------
// MODE 0: lower factor (diagonal last), 1: upper factor (diagonal first, rows walked backwards), 2: the part left of the diagonal
// of a general matrix with sorted rows (= the dependencies of its ILU(0) rows).
// A lane walks one BLOCK of the sweep's schedule (schedule.hip: consecutive rows that form a chain, on a mesh a grid line), the
// level of its previous row in a register.  With one row per lane the rows that can run are those of the mesh lines inside the
// window of resident lanes (2048 x 2048, 9-point: 256 of 2048 lines, 25 ms; 4096 x 4096: 190 ms); with a lane per line every line
// is resident, and the 64 lanes of a wave are 64 neighbouring lines, each a step or two behind the one before.  (Sixteen
// consecutive rows per lane instead: the lanes of a wave are then pieces of ONE line and run one after the other, 1.5 s.)
// A row costs its lane one round trip (the levels it reads) and not three: the row pointers are read three rows ahead and the
// column indices of the next row while the current one waits.
template <int MODE, int W>
__global__ void __launch_bounds__(kLvBlock)
k_lvl_levels(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t nb, const int32_t *__restrict__ start,
             int32_t B, int32_t *lev, int32_t *ticket, int32_t *err)
{
    constexpr bool FWD = MODE != 1;
    constexpr int dir = FWD ? 1 : -1;
    constexpr int D = kLvRing;
    // the last D levels of every lane, {row, level} in slot row % D: what the neighbouring lines read (a link through memory costs
    // ~3 us, and on a mesh every row hangs on the line before it: 9 us per row, 56 ms for 2048 x 2048, without this)
    extern __shared__ unsigned long long lv_ring[];
    __shared__ unsigned wg_ticket;
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(ticket, 1);
#pragma unroll
    for (int d = 0; d < D; ++d) lv_ring[d * kLvBlock + threadIdx.x] = FWD ? 0xffffffff00000000ull : 0x7fffffff00000000ull;
    __syncthreads();
    volatile unsigned long long *ring = lv_ring;
    const int64_t tb = (int64_t)wg_ticket * kLvBlock;
    const int64_t q = tb + threadIdx.x;
    bool active = q < nb;
    int r = 0, cnt = 0, r0 = 0;
    if (active) {
        const int b = FWD ? (int)q : (int)(nb - 1 - q);
        const int s0 = start[b], s1 = start[b + 1];
        r0 = FWD ? s0 : s1 - 1;
        cnt = s1 - s0;
        active = cnt > 0;
    }
    r = r0;
    // boundary k of the lane: row k (the k-th it walks) lies between boundary k and boundary k + 1
#define LV_BOUND(k) ptr[FWD ? r0 + ((k) < cnt ? (k) : cnt) : r0 + 1 - ((k) < cnt ? (k) : cnt)]
#define LV_RANGE(ea, eb, j0, j1)                                                                                   \
    do {                                                                                                           \
        const int lo_ = FWD ? (ea) : (eb), hi_ = FWD ? (eb) : (ea);                                                \
        j0 = lo_; j1 = lo_;                                                                                        \
        if (hi_ > lo_) { j0 = MODE == 1 ? lo_ + 1 : lo_; j1 = MODE == 0 ? hi_ - 1 : hi_; }                         \
    } while (0)
    // lane of this workgroup that walks row c (or -1), and the rows [rlo, rhi) of that lane's block.  Looked up when a row becomes the
    // current one (its columns were fetched a row earlier), and only when the column has left the block its window position pointed
    // into for the previous row: on a mesh that is once per line, not a division and two dependent loads per entry and row.
#define LV_OWNER(c, out, rlo, rhi)                                                                                 \
    do {                                                                                                           \
        if (!((c) >= (rlo) && (c) < (rhi))) {                                                                      \
            const int bb_ = block_of((c), B, nb, start);                                                            \
            rlo = start[bb_]; rhi = start[bb_ + 1];                                                                \
            const int64_t ql_ = (FWD ? (int64_t)bb_ : (int64_t)(nb - 1 - bb_)) - tb;                               \
            out = (ql_ >= 0 && ql_ < kLvBlock) ? (int)ql_ : -1;                                                    \
        }                                                                                                          \
    } while (0)
    int e0 = 0, e1 = 0, e2 = 0, e3 = 0, k = 0;
    int j = 0, jend = 0, mx = 0, rprev = -1, mprev = 0;
    int wc[W], nw[W], wo[W], clo[W], chi[W];
#pragma unroll
    for (int u = 0; u < W; ++u) { wc[u] = 0; nw[u] = 0; wo[u] = -1; clo[u] = 0; chi[u] = 0; }
    if (active) {
        e0 = LV_BOUND(0); e1 = LV_BOUND(1); e2 = LV_BOUND(2); e3 = LV_BOUND(3);
        LV_RANGE(e0, e1, j, jend);
        int n0, n1;
        LV_RANGE(e1, e2, n0, n1);
#pragma unroll
        for (int u = 0; u < W; ++u) { wc[u] = j + u < jend ? idx[j + u] : 0; nw[u] = n0 + u < n1 ? idx[n0 + u] : 0; }
#pragma unroll
        for (int u = 0; u < W; ++u)
            if (j + u < jend && !(MODE == 2 && wc[u] >= r)) LV_OWNER(wc[u], wo[u], clo[u], chi[u]);
    }
    int wbase = j;                                  // entry the window starts at
    unsigned spins = 0;
    for (;;) {
        if (__ballot(active) == 0ull) break;
        bool progressed = false;
        if (active) {
            if (j - wbase >= W && j < jend) {       // a row with more entries than the window: the next piece (not read ahead)
                wbase = j;
#pragma unroll
                for (int u = 0; u < W; ++u) wc[u] = j + u < jend ? idx[j + u] : 0;
#pragma unroll
                for (int u = 0; u < W; ++u)
                    if (j + u < jend && !(MODE == 2 && wc[u] >= r)) LV_OWNER(wc[u], wo[u], clo[u], chi[u]);
                progressed = true;
            }
            const int off = j - wbase;
            int wl[W];
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const bool in = u >= off && wbase + u < jend;
                const bool beyond = MODE == 2 && wc[u] >= r;         // (sorted rows: the part left of the diagonal ends here)
                int v = -1;
                if (in && !beyond) {
                    if (wc[u] == rprev) {
                        v = mprev;
                    } else if (wo[u] >= 0) {
                        const unsigned long long e = ring[(wc[u] & (D - 1)) * kLvBlock + wo[u]];
                        const int tag = (int)(e >> 32);
                        if (tag == wc[u]) v = (int)(unsigned)e;
                        else if (FWD ? tag > wc[u] : tag < wc[u]) v = ld_agent_i32(lev + wc[u]);    // the lane is more than D rows past it
                    } else {
                        v = ld_agent_i32(lev + wc[u]);
                    }
                }
                wl[u] = !in ? -1 : (beyond ? -2 : v);
            }
            bool stop = false, rowdone = false;
#pragma unroll
            for (int u = 0; u < W; ++u) {
                if (!stop && u >= off && wbase + u < jend) {
                    if (wl[u] == -2) { rowdone = true; stop = true; }
                    else if (wl[u] >= 0) { mx = wl[u] + 1 > mx ? wl[u] + 1 : mx; ++j; progressed = true; }
                    else stop = true;
                }
            }
            if (j == jend || rowdone) {
                st_agent_i32(lev + r, mx);
                ring[(r & (D - 1)) * kLvBlock + threadIdx.x] = ((unsigned long long)(unsigned)r << 32) | (unsigned long long)(unsigned)mx;
                rprev = r; mprev = mx;
                ++k;
                active = k < cnt;
                r += dir;
                // the next row becomes the current one; the row after it is fetched (its pointers are already here)
                e0 = e1; e1 = e2; e2 = e3;
                LV_RANGE(e0, e1, j, jend);
                wbase = j; mx = 0;
                int n0, n1;
                LV_RANGE(e1, e2, n0, n1);
#pragma unroll
                for (int u = 0; u < W; ++u) wc[u] = nw[u];
#pragma unroll
                for (int u = 0; u < W; ++u)
                    if (active && j + u < jend && !(MODE == 2 && wc[u] >= r)) LV_OWNER(wc[u], wo[u], clo[u], chi[u]);
#pragma unroll
                for (int u = 0; u < W; ++u) nw[u] = (active && n0 + u < n1) ? idx[n0 + u] : 0;
                if (active) e3 = LV_BOUND(k + 3);
                progressed = true;
            }
        }
        if (__any(progressed)) {
            spins = 0;
        } else {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kLvSpinLimit) {
                if ((threadIdx.x & 63) == 0) atomicExch(err, 1);
                break;
            }
        }
    }
#undef LV_BOUND
#undef LV_RANGE
#undef LV_OWNER
}

// position of every row, length of every row in position order
// The levels by relaxation (round 6): lev[r] = max over the rows r depends on of lev + 1, every row at once, again and again until
// nothing changes -- in place, so that within a pass a value travels as far as the order in which the hardware happens to run the rows
// lets it (rows are walked in the direction of the dependencies).  Any order of updates ends at the same numbers (the operator is
// monotone and starts from zero).  For patterns with SHORT rows whose levels are deep (a mesh that is no box: 766 levels at 256^3) the
// pass above walks the chains behind a window of resident blocks: 2.3 s, where this takes the passes' count times 0.1 ms.
// the farthest dependency of any row (rows sorted by column: the first entry of a lower part, the last of an upper one)
template <int MODE>
__global__ void __launch_bounds__(256)
k_lvl_reach(const int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t *__restrict__ reach)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    int d = 0;
    if (g < n) {
        const int r = (int)g, j0 = ptr[r], j1 = ptr[r + 1];
        if (j1 > j0) d = MODE == 1 ? idx[j1 - 1] - r : r - idx[j0];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d = max(d, __shfl_xor(d, o));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(reach, d);
}

template <int MODE>
__global__ void __launch_bounds__(256)
k_lvl_relax(const int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t *lev, int32_t *__restrict__ changed)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const int r = MODE == 1 ? n - 1 - (int)g : (int)g;
    int j0 = ptr[r], j1 = ptr[r + 1];
    if (j1 > j0) { if (MODE == 1) ++j0; if (MODE == 0) --j1; }
    int mx = 0;
    for (int j = j0; j < j1; ++j) {
        const int c = idx[j];
        if (MODE == 2 && c >= r) break;
        const int v = ld_agent_i32(lev + c) + 1;
        mx = v > mx ? v : mx;
    }
    if (mx > lev[r]) { st_agent_i32(lev + r, mx); *changed = 1; }
}

__global__ void k_lvl_pos(int32_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ ptr, int32_t *__restrict__ pos,
                          int32_t *__restrict__ plen)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n) return;
    if (t == n) { plen[n] = 0; return; }
    const int r = perm[t];
    pos[r] = (int32_t)t;
    plen[t] = ptr[r + 1] - ptr[r];
}

// the rows in position order, entries in application order, diagonal last, columns as positions
template <int KIND>
__global__ void k_lvl_fill(int32_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ pos, const int32_t *__restrict__ ptr,
                           const int32_t *__restrict__ idx, const double *__restrict__ val, const int32_t *__restrict__ ptrp,
                           int32_t *__restrict__ idxp, double *__restrict__ valp, int32_t *__restrict__ bad)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int r = perm[t];
    const int lo = ptr[r], hi = ptr[r + 1];
    if (hi <= lo) return;
    int o = ptrp[t];
    int wrong = 0;
    if (KIND == SWEEP_FWD_LAST_ASC) {
        for (int q = lo; q < hi - 1; ++q, ++o) { const int c = pos[idx[q]]; wrong |= c >= t; idxp[o] = c; valp[o] = val[q]; }
        idxp[o] = (int32_t)t; valp[o] = val[hi - 1];
    } else if (KIND == SWEEP_BWD_FIRST_ASC) {
        for (int q = lo + 1; q < hi; ++q, ++o) { const int c = pos[idx[q]]; wrong |= c >= t; idxp[o] = c; valp[o] = val[q]; }
        idxp[o] = (int32_t)t; valp[o] = val[lo];
    } else {
        for (int q = hi - 1; q > lo; --q, ++o) { const int c = pos[idx[q]]; wrong |= c >= t; idxp[o] = c; valp[o] = val[q]; }
        idxp[o] = (int32_t)t; valp[o] = val[lo];
    }
    if (wrong) atomicExch(bad, 1);
}

// ---------------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------------
template <int W, int BLOCK>                     // W: dependencies fetched per round trip
__global__ void __launch_bounds__(BLOCK)
k_sptrsv_lvl(int32_t n, const int32_t *__restrict__ ptrp, const int32_t *__restrict__ idxp, const double *__restrict__ valp,
             const int32_t *__restrict__ perm, double *rhs, double *xp, double *__restrict__ out, int32_t *ticket, int32_t *err)
{
    __shared__ unsigned wg_ticket;
    __shared__ unsigned long long xs[BLOCK];                      // this workgroup's unknowns, sentinel = not yet
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(ticket, 1);
    xs[threadIdx.x] = kSentinel;
    __syncthreads();
    const int64_t tb = (int64_t)wg_ticket * BLOCK;
    const int64_t t = tb + threadIdx.x;
    bool active = t < n;
    int j = 0, jend = 0, rn = 0;
    double acc = 0.0, dv = 1.0;
    if (active) {
        const int lo = ptrp[t], hi = ptrp[t + 1];
        rn = perm[t];
        acc = rhs[rn];
        reinterpret_cast<unsigned long long *>(rhs)[rn] = kSentinel;  // (the natural-order kernels use this buffer as the next sweep's output)
        j = lo; jend = hi - 1;
        if (hi > lo) dv = valp[hi - 1];
        else { j = jend = lo; dv = __longlong_as_double((long long)kCanonNaN); }
    }
    const unsigned long long *xpb = reinterpret_cast<const unsigned long long *>(xp);
    volatile unsigned long long *xsv = xs;
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
            const int left = jend - j;
            wn = left < W ? left : W;
            cur = 0;
#pragma unroll
            for (int u = 0; u < W; ++u) if (u < wn) { wc[u] = idxp[j + u]; wv[u] = valp[j + u]; }
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const long lb = (long)wc[u] - (long)tb;
                wb[u] = (u < wn && lb < 0) ? ld_agent_u64(xpb + wc[u]) : kSentinel;
            }
            progressed = true;
        } else if (active) {
            // every entry of the window that had not arrived is asked for again, all of them in one round trip (one at a time,
            // a row whose dependencies finished together -- the rule in level order -- paid a round trip for each)
#pragma unroll
            for (int u = 0; u < W; ++u)
                if (u >= cur && u < wn && wb[u] == kSentinel && (long)wc[u] < (long)tb) wb[u] = ld_agent_u64(xpb + wc[u]);
        }
        if (active) {
            // everything of the window that is there, in stored order (in level order that is nearly always all of it)
            bool stop = false;
#pragma unroll
            for (int u = 0; u < W; ++u) {
                if (!stop && u >= cur && u < wn) {
                    const int c = wc[u];
                    const long lb = (long)c - (long)tb;
                    unsigned long long b = wb[u];
                    if (lb >= 0) b = xsv[lb];
                    if (b != kSentinel) {
                        const double prod = wv[u] * __longlong_as_double((long long)b);
                        acc = acc - prod;                           // x[k] -= data[j]*x[indices[j]]  (sparse.hpp:4049, :4070)
                        ++j;
                        ++cur;
                        progressed = true;
                    } else {
                        stop = true;
                    }
                }
            }
            if (j == jend) {
                double x = acc / dv;                                // x[k] /= diagonal (by position)  (:4051, :4072)
                if (x != x) x = __longlong_as_double((long long)kCanonNaN);   // never store the sentinel
                st_agent_f64(xp + t, x);
                xsv[threadIdx.x] = (unsigned long long)__double_as_longlong(x);
                out[rn] = x;
                active = false;
                progressed = true;
            }
        }
        if (__any(progressed)) {
            spins = 0;
        } else {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kLvSpinLimit) {
                if ((threadIdx.x & 63) == 0) atomicExch(err, 1);
                break;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// rows of a triangular pattern (or of the lower part of a general one, mode 2) sorted by (level, row): perm[position] = row
bool lvl_order(hipStream_t st, int mode, int32_t n, int64_t nnz, const int32_t *ptr, const int32_t *idx, const Schedule &sch,
               int32_t **perm_out, int32_t *nlevels)
{
    *perm_out = nullptr;
    if (sch.nb <= 0 || !sch.start || sch.fwd != (mode != 1)) return false;
    const unsigned grid = (unsigned)((sch.nb + kLvBlock - 1) / kLvBlock);
    constexpr int kLvLds = kLvRing * kLvBlock * (int)sizeof(unsigned long long);
    static_assert(kLvLds <= 65536, "the level pass relies on the default limit of dynamic LDS");
    int32_t *lev = nullptr, *lev2 = nullptr, *iota = nullptr, *ctl = nullptr, *perm = nullptr;
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&lev, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&lev2, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&iota, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&ctl, 64));
    ILUPP_HIP(pool_malloc(&perm, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(hipMemsetAsync(lev, 0xff, sizeof(int32_t) * (size_t)n, st));
    ILUPP_HIP(hipMemsetAsync(ctl, 0, 64, st));
    // window = the entries an average row of the part has (every trip of the wait loop handles a whole window)
    const double per_row = mode == 2 ? 0.5 * ((double)nnz / (double)n - 1.0) : (double)nnz / (double)n - 1.0;
    const int w = per_row <= 4.5 ? 4 : (per_row <= 8.5 ? 8 : 16);
#define LV_LAUNCH(M, WW)                                                                                                        \
    hipLaunchKernelGGL((k_lvl_levels<M, WW>), dim3(grid), dim3(kLvBlock), kLvLds, st, n, ptr, idx, sch.nb, sch.start, sch.B, lev, ctl, ctl + 1)
#define LV_LAUNCH_W(M) do { if (w == 4) LV_LAUNCH(M, 4); else if (w == 8) LV_LAUNCH(M, 8); else LV_LAUNCH(M, 16); } while (0)
    static const bool no_relax = getenv("ILUPP_LVL_NO_RELAX") != nullptr;
    bool relaxed = false;
    static const bool relax_all = getenv("ILUPP_LVL_RELAX_ALL") != nullptr;
    // The relaxation takes about as many passes as there are levels (measured: 672 for 766 levels, 5 500 for 6 141), the pass above a
    // time per row that depends on how far a row's dependencies reach (9 ns per row on a 9-point 2048^2 matrix, reach 2 049; 140 ns on
    // the 3-D mesh with holes, reach 63 000, at every size).  For mesh-like patterns the number of levels is about 3 n / reach
    // (2-D: reach = sqrt n, 3 sqrt n levels; 3-D: n^(2/3), 3 n^(1/3)): the relaxation where that estimate is at most 1 024.
    bool relax_ok = false;
    if ((per_row <= 4.5 || relax_all) && n >= 65536 && !no_relax) {
        const dim3 rg((unsigned)(((int64_t)n + 255) / 256)), rb(256);
        if (mode == 1) hipLaunchKernelGGL((k_lvl_reach<1>), rg, rb, 0, st, n, ptr, idx, ctl + 4);
        else hipLaunchKernelGGL((k_lvl_reach<0>), rg, rb, 0, st, n, ptr, idx, ctl + 4);
        int32_t reach = 0;
        ILUPP_HIP(d2h_async(st, &reach, ctl + 4, sizeof(reach)));
        ILUPP_HIP(stream_sync(st));
        relax_ok = relax_all || (reach > 0 && 3.0 * (double)n / (double)reach <= 1024.0);
        if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] level numbering (mode %d, n %d): dependencies reach %d rows: %s\n", mode, n, reach, relax_ok ? "by relaxation" : "in natural order");
    }
    if (relax_ok) {
        // short rows: by relaxation, 32 passes between two looks at the flags (one per pass; the last one clear = nothing changed any more)
        int32_t *flags = nullptr;
        ILUPP_HIP(pool_malloc(&flags, 32 * sizeof(int32_t)));
        ILUPP_HIP(hipMemsetAsync(lev, 0, sizeof(int32_t) * (size_t)n, st));
        const dim3 rg((unsigned)(((int64_t)n + 255) / 256)), rb(256);
        int32_t hf[32];
        static const int max_batches = getenv("ILUPP_LVL_RELAX_BATCHES") ? atoi(getenv("ILUPP_LVL_RELAX_BATCHES")) : 64;
        int batches = 0;
        for (int batch = 0; batch < max_batches && !relaxed; ++batch) {
            ++batches;
            ILUPP_HIP(hipMemsetAsync(flags, 0, 32 * sizeof(int32_t), st));
            for (int k = 0; k < 32; ++k) {
                if (mode == 0) hipLaunchKernelGGL((k_lvl_relax<0>), rg, rb, 0, st, n, ptr, idx, lev, flags + k);
                else if (mode == 1) hipLaunchKernelGGL((k_lvl_relax<1>), rg, rb, 0, st, n, ptr, idx, lev, flags + k);
                else hipLaunchKernelGGL((k_lvl_relax<2>), rg, rb, 0, st, n, ptr, idx, lev, flags + k);
            }
            ILUPP_HIP(d2h_async(st, hf, flags, sizeof(hf)));
            ILUPP_HIP(stream_sync(st));
            relaxed = hf[31] == 0;
        }
        (void)pool_free(flags);
        if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] level numbering by relaxation (mode %d, n %d): %d batches of 32 passes, %s\n", mode, n, batches, relaxed ? "converged" : "given up");
        if (!relaxed) ILUPP_HIP(hipMemsetAsync(lev, 0xff, sizeof(int32_t) * (size_t)n, st));     // (2 048 passes were not enough: the pass below)
    }
    if (relaxed) { }
    else if (mode == 0) LV_LAUNCH_W(0); else if (mode == 1) LV_LAUNCH_W(1); else LV_LAUNCH_W(2);
#undef LV_LAUNCH_W
#undef LV_LAUNCH
    size_t b1 = 0, b2 = 0;
    ILUPP_HIP(hipcub::DeviceReduce::Max(nullptr, b1, lev, ctl + 2, n, st));
    ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, b2, lev, lev2, iota, perm, n, 0, 32, st));
    const size_t tb = b1 > b2 ? b1 : b2;
    ILUPP_HIP(pool_malloc(&tmp, tb));
    size_t bb = tb;
    ILUPP_HIP(hipcub::DeviceReduce::Max(tmp, bb, lev, ctl + 2, n, st));
    int32_t h[4] = {0, 1, 0, 0};
    ILUPP_HIP(d2h_async(st, h, ctl, sizeof(h)));
    ILUPP_HIP(stream_sync(st));
    const bool ok = h[1] == 0 && h[2] >= 0;
    if (ok) {
        int bits = 1;
        while (bits < 31 && (1 << bits) <= h[2]) ++bits;
        iota_i32(st, iota, n);
        bb = tb;
        ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, bb, lev, lev2, iota, perm, n, 0, bits, st));
        ILUPP_HIP(stream_sync(st));                      // (the scratch goes back to the pool)
    }
    for (void *q : {(void *)lev, (void *)lev2, (void *)iota, (void *)ctl, tmp}) if (q) (void)pool_free(q);
    if (!ok) { (void)pool_free(perm); return false; }
    *perm_out = perm;
    *nlevels = h[2] + 1;
    return true;
}

// perm_given (with its number of levels): an order somebody has already computed for this pattern (the ILU(0) factor kernel's for L)
bool lvl_build(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, LevelSweep *ls, const int32_t *perm_given,
               int32_t nlevels_given)
{
    static const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    ls->release();
    ls->tried = true;
    const int32_t n = M.n;
    if (n < 1024 || !M.ptr || !M.idx || !M.val || M.nnz >= 0x7fffffffLL) return false;
    int32_t nlev = nlevels_given;
    if (perm_given) {
        ILUPP_HIP(pool_malloc(&ls->perm, sizeof(int32_t) * (size_t)n));
        ILUPP_HIP(hipMemcpyAsync(ls->perm, perm_given, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    } else if (!lvl_order(st, kind == SWEEP_FWD_LAST_ASC ? 0 : 1, n, M.nnz, M.ptr, M.idx, sch, &ls->perm, &nlev)) {
        if (dbg) fprintf(stderr, "[ilupp] level order of a factor (kind %d, n %d): declined\n", (int)kind, n);
        return false;
    }
    int32_t *pos = nullptr, *plen = nullptr, *ctl = nullptr;
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&pos, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&plen, sizeof(int32_t) * ((size_t)n + 1)));
    ILUPP_HIP(pool_malloc(&ctl, 64));
    ILUPP_HIP(hipMemsetAsync(ctl, 0, 64, st));
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, plen, plen, n + 1, st));
    ILUPP_HIP(pool_malloc(&tmp, tb));
    hipLaunchKernelGGL(k_lvl_pos, dim3((unsigned)(((int64_t)n + 1 + 255) / 256)), dim3(256), 0, st, n, ls->perm, M.ptr, pos, plen);
    ILUPP_HIP(pool_malloc(&ls->ptr, sizeof(int32_t) * ((size_t)n + 1)));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, plen, ls->ptr, n + 1, st));
    ILUPP_HIP(pool_malloc(&ls->idx, sizeof(int32_t) * (size_t)(M.nnz > 0 ? M.nnz : 1)));
    ILUPP_HIP(pool_malloc(&ls->val, sizeof(double) * (size_t)(M.nnz > 0 ? M.nnz : 1)));
    ILUPP_HIP(pool_malloc(&ls->xp, sizeof(double) * (size_t)n));
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    switch (kind) {
    case SWEEP_FWD_LAST_ASC:
        hipLaunchKernelGGL((k_lvl_fill<SWEEP_FWD_LAST_ASC>), g, b, 0, st, n, ls->perm, pos, M.ptr, M.idx, M.val, ls->ptr, ls->idx, ls->val, ctl + 3);
        break;
    case SWEEP_BWD_FIRST_ASC:
        hipLaunchKernelGGL((k_lvl_fill<SWEEP_BWD_FIRST_ASC>), g, b, 0, st, n, ls->perm, pos, M.ptr, M.idx, M.val, ls->ptr, ls->idx, ls->val, ctl + 3);
        break;
    default:
        hipLaunchKernelGGL((k_lvl_fill<SWEEP_BWD_FIRST_DESC>), g, b, 0, st, n, ls->perm, pos, M.ptr, M.idx, M.val, ls->ptr, ls->idx, ls->val, ctl + 3);
        break;
    }
    int32_t h[4] = {0, 0, 0, 1};
    ILUPP_HIP(d2h_async(st, h, ctl, sizeof(h)));
    ILUPP_HIP(stream_sync(st));
    const bool ok = h[3] == 0;
    if (dbg) fprintf(stderr, "[ilupp] level order of a factor (kind %d, n %d, %.1f entries per row): %s, %d levels\n", (int)kind, n, (double)M.nnz / n, ok ? "built" : "declined", nlev);
    for (void *q : {(void *)pos, (void *)plen, (void *)ctl, tmp}) if (q) (void)pool_free(q);
    if (!ok) { ls->release(); ls->tried = true; return false; }
    ls->n = n;
    ls->nlevels = nlev;
    // window = the off-diagonal entries of an average row (a wider one makes every trip of the wait loop longer: 9-point, 4 entries:
    // 23 ms with 8, 34 ms with 16; a narrower one costs a round trip per refill: 27-point, 13 entries: 12 ms with 8, 7.5 ms with 16)
    const double offd = (double)M.nnz / (double)n - 1.0;
    ls->w = offd <= 4.5 ? 4 : (offd <= 8.5 ? 8 : 16);
    ls->block = (int64_t)n / ls->nlevels < 4096 ? 256 : 1024;
    ls->valid = true;
    return true;
}

int sptrsv_lvl(hipStream_t st, const LevelSweep &ls, double *rhs_and_reset, double *out, int32_t *d_ticket, int32_t *d_err)
{
    fill_u64(st, reinterpret_cast<unsigned long long *>(ls.xp), ls.n, kSentinel);
#define LVL_LAUNCH(W, B)                                                                                                     \
    hipLaunchKernelGGL((k_sptrsv_lvl<W, B>), dim3((unsigned)((ls.n + (B) - 1) / (B))), dim3(B), 0, st, ls.n, ls.ptr, ls.idx, ls.val, \
                       ls.perm, rhs_and_reset, ls.xp, out, d_ticket, d_err)
    if (ls.block == 256) {
        if (ls.w == 4) LVL_LAUNCH(4, 256); else if (ls.w == 8) LVL_LAUNCH(8, 256); else LVL_LAUNCH(16, 256);
    } else {
        if (ls.w == 4) LVL_LAUNCH(4, 1024); else if (ls.w == 8) LVL_LAUNCH(8, 1024); else LVL_LAUNCH(16, 1024);
    }
#undef LVL_LAUNCH
    ILUPP_HIP(hipGetLastError());
    return ILUPP_OK;
}

}  // namespace ilupp
