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

static constexpr int kLvBlock = 1024;           // (the level pass; the sweep kernel has two block sizes)
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
template <bool FWD>
__global__ void __launch_bounds__(kLvBlock)
k_lvl_levels(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t *lev, int32_t *ticket, int32_t *err)
{
    constexpr int W = 8;
    __shared__ unsigned wg_ticket;
    __shared__ int ls[kLvBlock];
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(ticket, 1);
    ls[threadIdx.x] = -1;
    __syncthreads();
    const int64_t tb = (int64_t)wg_ticket * kLvBlock;
    const int64_t t = tb + threadIdx.x;
    bool active = t < n;
    const int r = active ? (FWD ? (int)t : (int)(n - 1 - t)) : 0;
    const long row0 = FWD ? (long)tb : (long)n - 1 - (long)tb;
    int j = 0, jend = 0, mx = 0;
    if (active) {
        const int lo = ptr[r], hi = ptr[r + 1];
        if (hi > lo) { j = FWD ? lo : lo + 1; jend = FWD ? hi - 1 : hi; }
    }
    volatile int *lsv = ls;
    unsigned spins = 0;
    for (;;) {
        if (__ballot(active) == 0ull) break;
        bool progressed = false;
        if (active) {
            int wc[W], wl[W];
            const int left = jend - j;
            const int wn = left < W ? left : W;
#pragma unroll
            for (int u = 0; u < W; ++u) wc[u] = u < wn ? idx[j + u] : 0;
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const long lb = FWD ? (long)wc[u] - row0 : row0 - (long)wc[u];
                wl[u] = u >= wn ? -1 : ((unsigned long)lb < (unsigned long)kLvBlock ? lsv[lb] : ld_agent_i32(lev + wc[u]));
            }
            bool stop = false;
#pragma unroll
            for (int u = 0; u < W; ++u) {
                if (!stop && u < wn) {
                    if (wl[u] >= 0) { mx = wl[u] + 1 > mx ? wl[u] + 1 : mx; ++j; progressed = true; }
                    else stop = true;
                }
            }
            if (j == jend) {
                st_agent_i32(lev + r, mx);
                lsv[threadIdx.x] = mx;
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

// position of every row, length of every row in position order
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
bool lvl_build(hipStream_t st, SweepKind kind, const DevMat &M, LevelSweep *ls)
{
    static const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    ls->release();
    ls->tried = true;
    const int32_t n = M.n;
    if (n < 1024 || !M.ptr || !M.idx || !M.val || M.nnz >= 0x7fffffffLL) return false;
    const bool fwd = kind == SWEEP_FWD_LAST_ASC;
    const unsigned grid = (unsigned)((n + kLvBlock - 1) / kLvBlock);
    int32_t *lev = nullptr, *lev2 = nullptr, *iota = nullptr, *pos = nullptr, *ctl = nullptr, *plen = nullptr;
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&lev, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&lev2, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&iota, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&pos, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&plen, sizeof(int32_t) * ((size_t)n + 1)));
    ILUPP_HIP(pool_malloc(&ctl, 64));
    ILUPP_HIP(pool_malloc(&ls->perm, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(hipMemsetAsync(lev, 0xff, sizeof(int32_t) * (size_t)n, st));
    ILUPP_HIP(hipMemsetAsync(ctl, 0, 64, st));
    if (fwd) hipLaunchKernelGGL((k_lvl_levels<true>), dim3(grid), dim3(kLvBlock), 0, st, n, M.ptr, M.idx, lev, ctl, ctl + 1);
    else     hipLaunchKernelGGL((k_lvl_levels<false>), dim3(grid), dim3(kLvBlock), 0, st, n, M.ptr, M.idx, lev, ctl, ctl + 1);
    size_t b1 = 0, b2 = 0, b3 = 0;
    ILUPP_HIP(hipcub::DeviceReduce::Max(nullptr, b1, lev, ctl + 2, n, st));
    ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, b2, lev, lev2, iota, ls->perm, n, 0, 32, st));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b3, plen, plen, n + 1, st));
    const size_t tb = b1 > b2 ? (b1 > b3 ? b1 : b3) : (b2 > b3 ? b2 : b3);
    ILUPP_HIP(pool_malloc(&tmp, tb));
    size_t bb = tb;
    ILUPP_HIP(hipcub::DeviceReduce::Max(tmp, bb, lev, ctl + 2, n, st));
    int32_t h[4] = {0, 1, 0, 0};
    ILUPP_HIP(d2h_async(st, h, ctl, sizeof(h)));
    ILUPP_HIP(stream_sync(st));
    bool ok = h[1] == 0 && h[2] >= 0;
    if (ok) {
        int bits = 1;
        while (bits < 31 && (1 << bits) <= h[2]) ++bits;
        iota_i32(st, iota, n);
        bb = tb;
        ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, bb, lev, lev2, iota, ls->perm, n, 0, bits, st));
        hipLaunchKernelGGL(k_lvl_pos, dim3((unsigned)(((int64_t)n + 1 + 255) / 256)), dim3(256), 0, st, n, ls->perm, M.ptr, pos, plen);
        ILUPP_HIP(pool_malloc(&ls->ptr, sizeof(int32_t) * ((size_t)n + 1)));
        bb = tb;
        ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, bb, plen, ls->ptr, n + 1, st));
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
        ILUPP_HIP(d2h_async(st, h, ctl, sizeof(h)));
        ILUPP_HIP(stream_sync(st));
        ok = h[3] == 0;
    }
    if (dbg) fprintf(stderr, "[ilupp] level order of a factor (kind %d, n %d, %.1f entries per row): %s, %d levels\n", (int)kind, n, (double)M.nnz / n, ok ? "built" : "declined", h[2] + 1);
    for (void *q : {(void *)lev, (void *)lev2, (void *)iota, (void *)pos, (void *)plen, (void *)ctl, tmp}) if (q) (void)pool_free(q);
    if (!ok) { ls->release(); ls->tried = true; return false; }
    ls->n = n;
    ls->nlevels = h[2] + 1;
    // window = the off-diagonal entries of an average row (a wider one makes every trip of the wait loop longer: 9-point, 4 entries:
    // 23 ms with 8, 34 ms with 16; a narrower one costs a round trip per refill: 27-point, 13 entries: 12 ms with 8, 7.5 ms with 16)
    const double offd = (double)M.nnz / (double)n - 1.0;
    ls->w = offd <= 4.5 ? 4 : (offd <= 8.5 ? 8 : 16);
    ls->block = (int64_t)n / ls->nlevels < 4096 ? 256 : 1024;
#ifdef LV_FORCE_W
    ls->w = LV_FORCE_W;
#endif
#ifdef LV_FORCE_BLOCK
    ls->block = LV_FORCE_BLOCK;
#endif
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
