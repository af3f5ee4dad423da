// ilupp_amd/csrc/grid.hip -- the first analysis pass of ILU(0) for matrices whose pattern is a lexicographic box-grid stencil
// (5-point on nx x ny, 7-point on nx x ny x nz: the matrices of BASELINE configs C1, C2, C4).
//
// What the general first pass (symbolic.hip: k_row_cuts_counts, k_reduce_stats, k_block_starts; schedule.hip: the tiling samples)
// finds out about such a matrix -- rows r - 1 and r are linked inside an x-line and nowhere else, all lines are alike, every row has its
// diagonal, the line grid has the (1, ny) dependency structure -- follows from THREE numbers.  The host reads them off row 0 (its
// columns are 0, 1, nx, nx ny), checks the entry count against the closed form, and then
//   * writes the two schedules' block starts (start[b] = b nx) without looking at the pattern,
//   * lets ONE streaming kernel on a side stream prove the guess for every row: k_grid_check compares each row's pointer and columns
//     with the closed form (0.54 GB at 256^3, no reduction, no output but one flag), next to the lane-table kernels of the static
//     analysis, which only sample rows and need nothing from it;
//   * takes the flag home with the read-back the static analysis makes anyway.  A matrix that merely begins like a grid fails the
//     proof: everything built on the guess is dropped and the general pass runs (api.hip: ilu0_factor).
// Nothing downstream changes: the lane tables, the factor kernel and the sweeps are the ones of st.hip / st_wave.hip; the reference
// semantics are ILU0.hpp:26-66 as before (this file only replaces how the row blocks are found, ILU0.hpp has no counterpart).
#include <mutex>
#include <utility>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "st_common.h"

namespace ilupp {

// entries of a box-grid stencil matrix: n + 2 (links in x + links in y + links in z)
static int64_t grid_links(const GridDims &g)
{
    const int64_t nx = g.nx, ny = g.ny, nz = g.nz;
    return (nx - 1) * ny * nz + nx * (ny - 1) * nz + nx * ny * (nz - 1);
}

bool grid_guess(int32_t n, int64_t nnz, const int32_t *head, GridDims *g)
{
    static const bool off = getenv("ILUPP_NO_GRID") != nullptr;
    if (off || n < (1 << 16)) return false;
    // head: {ptr[0], ptr[1], idx[0..7]}
    if (head[0] != 0) return false;
    const int len0 = head[1];
    const int32_t *c = head + 2;
    int64_t nx = 0, ny = 0, nz = 0;
    if (len0 == 4 && c[0] == 0 && c[1] == 1 && c[2] >= 2 && c[3] > c[2]) {
        nx = c[2];
        if (c[3] % nx != 0 || (int64_t)n % c[3] != 0) return false;
        ny = c[3] / nx; nz = (int64_t)n / c[3];
        if (nz < 2) return false;
    } else if (len0 == 3 && c[0] == 0 && c[1] == 1 && c[2] >= 2) {
        nx = c[2];
        if ((int64_t)n % nx != 0) return false;
        ny = (int64_t)n / nx; nz = 1;
    } else {
        return false;
    }
    // lines long enough to be lanes' chains, enough of them to fill workgroups (anything smaller: the general pass, which is quick there)
    if (nx < 16 || ny < 4 || ny * nz < 2 * kThreads || nx > (1 << 20) || ny > (1 << 20) || nz > (1 << 20)) return false;
    g->nx = (int32_t)nx; g->ny = (int32_t)ny; g->nz = (int32_t)nz;
    return nnz == (int64_t)n + 2 * grid_links(*g) && nnz < (1LL << 30);        // (k_grid_check covers the index array with one buffer resource)
}

// the dimensions of the box grids this process has factored, by (n, nnz): ilupp_hip_ilu0_create_device_nnz guesses them again without
// reading the matrix' head back (a guess, nothing more: the proof runs every time)
namespace {
// (n, nnz) -> dimensions, with the index array's address as a hint: two matrices of equal size and entry count but other dimensions
// (64 x 128 x 256 and 128 x 64 x 256) live in different arrays and keep an entry each; a lookup prefers the entry of the same array
struct ShapeEntry { int64_t n, nnz; const void *idx; GridDims g; };
struct ShapeMemo { std::mutex mu; std::vector<ShapeEntry> seen; } g_shapes;
}
bool grid_shape_recall(int32_t n, int64_t nnz, GridDims *g, const void *idx)
{
    std::lock_guard<std::mutex> lk(g_shapes.mu);
    const ShapeEntry *any = nullptr;
    for (const auto &e : g_shapes.seen) {
        if (e.n != n || e.nnz != nnz) continue;
        if (e.idx == idx) { *g = e.g; return true; }
        any = &e;                                       // (the most recent one of that size)
    }
    if (any) { *g = any->g; return true; }
    return false;
}
void grid_shape_remember(int32_t n, int64_t nnz, const GridDims &g, const void *idx)
{
    std::lock_guard<std::mutex> lk(g_shapes.mu);
    for (size_t i = 0; i < g_shapes.seen.size(); ++i) {
        auto &e = g_shapes.seen[i];
        if (e.n == n && e.nnz == nnz && e.idx == idx) {
            // (moved to the back: the most recent entry of a size is what a lookup from another array gets)
            ShapeEntry m = e; m.g = g;
            g_shapes.seen.erase(g_shapes.seen.begin() + (long)i);
            g_shapes.seen.push_back(m);
            return;
        }
    }
    if (g_shapes.seen.size() >= 64) g_shapes.seen.erase(g_shapes.seen.begin());
    g_shapes.seen.push_back({n, nnz, idx, g});
}
void grid_shape_forget(int32_t n, int64_t nnz, const void *idx)
{
    std::lock_guard<std::mutex> lk(g_shapes.mu);
    // the entry of this array if there is one, else the one a lookup from this array got (the most recent of that size)
    long hit = -1, last = -1;
    for (size_t i = 0; i < g_shapes.seen.size(); ++i) {
        const auto &e = g_shapes.seen[i];
        if (e.n != n || e.nnz != nnz) continue;
        last = (long)i;
        if (e.idx == idx) hit = (long)i;
    }
    const long victim = hit >= 0 ? hit : last;
    if (victim >= 0) g_shapes.seen.erase(g_shapes.seen.begin() + victim);
}

// One row per lane.  The eight index words a row can reach from its expected start are fetched with two 16-byte loads whatever the
// row turns out to hold (buffer loads: past the end of the array they return zeros), so nothing about a row waits for anything else
// about it; a wave's 64 rows read one contiguous run of the index array (1.8 KB).
__global__ void __launch_bounds__(256)
k_grid_check(const int32_t n, const long long nnz, const GridDims g, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
             int32_t *__restrict__ bad)
{
    typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
    bool ok = true;
    const unsigned unx = (unsigned)g.nx, uny = (unsigned)g.ny;
    const int sxy = g.nx * g.ny;
    // (an index array of more than 4 GB cannot be covered by one buffer resource: the host does not take such a matrix here)
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(idx), 0, (int)((unsigned)nnz * 4u), 0x00020000);
    for (long long r0 = (long long)blockIdx.x * 256; r0 < n; r0 += (long long)gridDim.x * 256) {
        const long long rr = r0 + threadIdx.x;
        if (rr >= n) break;
        const unsigned r = (unsigned)rr;
        const unsigned l = r / unx, x = r - l * unx;
        const unsigned z = l / uny, y = l - z * uny;
        const long long e = grid_row_start((int)x, (int)y, (int)z, g);
        const unsigned eo = (unsigned)e * 4u;
        const v4u_ c0 = __builtin_amdgcn_raw_buffer_load_b128(ri, eo, 0, 0);
        const v4u_ c1 = __builtin_amdgcn_raw_buffer_load_b128(ri, eo + 16u, 0, 0);
        const int p = ptr[r];
        const int pn = rr == n - 1 ? ptr[n] : 0;
        ok = ok && (long long)p == e && (rr != n - 1 || (long long)pn == nnz);
        // the expected columns, in stored order, against the words that were fetched
        const int ri_ = (int)r;
        int want[7];
        int m = 0;
        if (z > 0) want[m++] = ri_ - sxy;
        if (y > 0) want[m++] = ri_ - g.nx;
        if (x > 0) want[m++] = ri_ - 1;
        want[m++] = ri_;
        if ((int)x < g.nx - 1) want[m++] = ri_ + 1;
        if ((int)y < g.ny - 1) want[m++] = ri_ + g.nx;
        if ((int)z < g.nz - 1) want[m++] = ri_ + sxy;
        const int got[8] = {(int)c0.x, (int)c0.y, (int)c0.z, (int)c0.w, (int)c1.x, (int)c1.y, (int)c1.z, (int)c1.w};
#pragma unroll
        for (int j = 0; j < 7; ++j) ok = ok && (j >= m || got[j] == want[j]);
    }
    if (__builtin_amdgcn_ballot_w64(!ok) != 0 && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

__global__ void k_grid_starts(const int32_t n, const int32_t nx, const int32_t nb, int32_t *__restrict__ sf, int32_t *__restrict__ sb)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    const long long s = (long long)b * nx;
    const int32_t v = b == nb ? n : (int32_t)(s < n ? s : n);
    sf[b] = v; sb[b] = v;
}

// ---------------------------------------------------------------------------------------------
// The slot tables (schedule.hip: k_slot_tables), the lane templates (st.hip: st_template_body) and the link between the two
// schedules (sptrsv_lm.hip: k_lm_uslot) of a box grid, from its dimensions: what those kernels find by following ptr -> idx -> start ->
// blk2slot for three sampled rows of every lane is known here -- line (y, z) depends on the lines (y - 1, z) and (y, z - 1) at equal x
// and on its own previous row.  One launch, no dependent loads; block (w, d): workgroup w of the forward (d = 0) / backward schedule.
// ---------------------------------------------------------------------------------------------
struct GridPlace { int32_t nx, ny, nz, nb, s2, ty, tz, NY, nslots; };

// slot of the block with index bs in SWEEP order (forward: the block's own index, backward: nb - 1 - index); schedule.hip: tiled_block_of
__device__ __forceinline__ int grid_place(const int bs, const GridPlace &g)
{
    if (g.s2 <= 0) return bs;
    const int by = bs % g.s2, bz = bs / g.s2;
    return ((bz / g.tz) * g.NY + by / g.ty) * kThreads + (bz % g.tz) * g.ty + by % g.ty;
}
__device__ __forceinline__ int grid_block_at(const int slot, const GridPlace &g)
{
    if (g.s2 <= 0) return slot < g.nb ? slot : -1;
    const int T = slot / kThreads, lane = slot % kThreads;
    const int by = (T % g.NY) * g.ty + lane % g.ty;
    const int bz = (T / g.NY) * g.tz + lane / g.ty;
    if (by >= g.s2 || lane / g.ty >= g.tz) return -1;
    const long bs = (long)bz * g.s2 + by;
    return bs < g.nb ? (int)bs : -1;
}

struct GridLaneArgs { int32_t *slot2blk, *blk2slot, *sfirst, *scount, *exported, *ltab, *flags, *skew, *wtab; };

__global__ void __launch_bounds__(kThreads)
k_grid_lanes(const GridPlace g, const GridLaneArgs F, const GridLaneArgs Bk, int32_t *__restrict__ uslot, const int link)
{
    const bool fwd = blockIdx.y == 0;
    const GridLaneArgs &X = fwd ? F : Bk;
    const int wg = blockIdx.x, t = threadIdx.x;
    const int slot = wg * kThreads + t;
    if (slot >= g.nslots) return;
    const int bs = grid_block_at(slot, g);
    const int b = bs < 0 ? -1 : (fwd ? bs : g.nb - 1 - bs);
    const bool has = b >= 0;
    const int cnt = has ? g.nx : 0;
    const int first = has ? (fwd ? b * g.nx : (b + 1) * g.nx - 1) : 0;        // first row in processing order
    const int y = has ? b % g.ny : 0, z = has ? b / g.ny : 0;
    const int sxy = g.nx * g.ny;
    // The (at most three) lines this one depends on as CANDIDATES in ascending column offset (= the reference's elimination /
    // accumulation order) -- forward: the line below (z - 1), the line before (y - 1), the own chain; backward: the own chain, the line
    // after (y + 1), the line above (z + 1) -- then the ones that exist moved up to the front of the template.  Everything stays in
    // registers (no array is indexed with a run-time value) and the row of the table leaves as eight 16-byte stores.
    const bool p0 = has && (fwd ? z > 0 : g.nx > 1), p1 = has && (fwd ? y > 0 : y < g.ny - 1), p2 = has && (fwd ? g.nx > 1 : z < g.nz - 1);
    const int cb0 = fwd ? b - g.ny : b, cb1 = fwd ? b - 1 : b + 1, cb2 = fwd ? b : b + g.ny;      // candidate lines
    const int co0 = fwd ? -sxy : 1, co1 = fwd ? -g.nx : g.nx, co2 = fwd ? -1 : sxy;               // their column offsets
    const bool own0 = !fwd, own2 = fwd;                                                           // which candidate is the own chain
    int bad = 0, ngh = 0;
    auto source = [&](const bool present, const bool own, const int bo) -> int {
        if (!present) return 0;
        if (own) return ST_OWN | (slot << 2);
        const int os = grid_place(fwd ? bo : g.nb - 1 - bo, g);
        if ((os >> 8) == wg) return ST_LOCAL | (os << 2);
        ++ngh; if ((os >> 8) >= wg) bad = 1;
        return ST_GHOST | (os << 2);
    };
    const int cs0 = source(p0, own0, cb0), cs1 = source(p1, false, cb1), cs2 = source(p2, own2, cb2);
    // j-th present candidate
    const int i0 = p0 ? 0 : (p1 ? 1 : (p2 ? 2 : -1));
    const int i1 = p0 ? (p1 ? 1 : (p2 ? 2 : -1)) : ((p1 && p2) ? 2 : -1);
    const int i2 = (p0 && p1 && p2) ? 2 : -1;
    const int nd = (p0 ? 1 : 0) + (p1 ? 1 : 0) + (p2 ? 1 : 0);
#define GRID_PICK(i_, a0_, a1_, a2_) ((i_) == 0 ? (a0_) : ((i_) == 1 ? (a1_) : ((i_) == 2 ? (a2_) : 0)))
    const int of0 = GRID_PICK(i0, co0, co1, co2), of1 = GRID_PICK(i1, co0, co1, co2), of2 = GRID_PICK(i2, co0, co1, co2);
    const int sr0 = GRID_PICK(i0, cs0, cs1, cs2), sr1 = GRID_PICK(i1, cs0, cs1, cs2), sr2 = GRID_PICK(i2, cs0, cs1, cs2);
#undef GRID_PICK
    const bool o0 = (sr0 & 3) == ST_OWN, o1 = (sr1 & 3) == ST_OWN, o2 = (sr2 & 3) == ST_OWN;
    // (own chain: every row but the first has the entry, from the row before; the other lines: every row, from the row at the same place)
    const int ka0 = o0 ? -1 : 0, ka1 = o1 ? -1 : 0, ka2 = o2 ? -1 : 0;
    const int kl0 = o0 ? 1 : 0, kl1 = o1 ? 1 : 0, kl2 = o2 ? 1 : 0;
    const int kh0 = i0 >= 0 ? cnt : 0, kh1 = i1 >= 0 ? cnt : 0, kh2 = i2 >= 0 ? cnt : 0;
    if (nd == 3 && ngh == 3) bad = 1;
    // the lines that depend on this one: a slot some OTHER workgroup reads is exported
    bool ex = false;
    if (has) {
        const int r0 = fwd ? b + 1 : b - 1, r1 = fwd ? b + g.ny : b - g.ny;
        const bool q0 = fwd ? y < g.ny - 1 : y > 0, q1 = fwd ? z < g.nz - 1 : z > 0;
        if (q0 && (grid_place(fwd ? r0 : g.nb - 1 - r0, g) >> 8) != wg) ex = true;
        if (q1 && (grid_place(fwd ? r1 : g.nb - 1 - r1, g) >> 8) != wg) ex = true;
    }
    // (16 x 16 patches, wave-exchange skews: `link`) what k_st_link's fixpoint arrives at: a step per neighbour inside the patch, kWrLag
    // steps where the neighbour below sits in another wave; its proof about the eliminations (they meet the eliminated row on its
    // diagonal only) holds for every box stencil with nx >= 3, ny >= 3 (grid_guess asks for more)
    const int sk = (link && has) ? (t & 15) + (t >> 4) + (t >> 6) * (kWrLag - 1) : 0;
    int dt0 = 1, dt1 = 1, dt2 = 1;
    if (link) {
        if ((sr0 & 3) == ST_LOCAL) dt0 = wr_edge_lag(t, (sr0 >> 2) & 255, true);
        if ((sr1 & 3) == ST_LOCAL) dt1 = wr_edge_lag(t, (sr1 >> 2) & 255, true);
        if ((sr2 & 3) == ST_LOCAL) dt2 = wr_edge_lag(t, (sr2 >> 2) & 255, true);
    }
    // the row: {FIRST, CNT, SKEW, ND} {OFF x3, SRC0} {SRC1, SRC2, KLO0, KLO1} {KLO2, KAP x3} {UP0, DT x3} {KHI x3, SCAT0} {SCAT1, SCAT2, -, -} {- x4}
    static_assert(ST_FIRST == 0 && ST_CNT == 1 && ST_SKEW == 2 && ST_ND == 3 && ST_OFF == 4 && ST_SRC == 7 && ST_KLO == 10 && ST_KAP == 13 &&
                  ST_UP0 == 16 && ST_DT == 17 && ST_KHI == 20 && ST_SCAT == 23 && kStTab == 32, "layout of a lane table row");
    int4 *T4 = reinterpret_cast<int4 *>(X.ltab + (size_t)slot * kStTab);
    T4[0] = make_int4(first, cnt, sk, nd);
    T4[1] = make_int4(of0, of1, of2, sr0);
    T4[2] = make_int4(sr1, sr2, kl0, kl1);
    T4[3] = make_int4(kl2, ka0, ka1, ka2);
    T4[4] = make_int4(0, dt0, dt1, dt2);
    T4[5] = make_int4(kh0, kh1, kh2, -1);
    T4[6] = make_int4(-1, -1, 0, 0);
    T4[7] = make_int4(0, 0, 0, 0);
    X.slot2blk[slot] = b;
    X.sfirst[slot] = first; X.scount[slot] = cnt; X.exported[slot] = ex ? 1 : 0;
    if (has) X.blk2slot[b] = slot;
    if (fwd) uslot[slot] = has ? grid_place(g.nb - 1 - b, g) : -1;       // the slot of the backward schedule that owns the same line
    if (bad) atomicOr(&X.flags[0], 2);
    if (link) {
        X.skew[slot] = sk;
        // may the wave-exchange kernels run this lane?  (st_common.h: wx_lane_ok, on the row as it stands in registers)
        if (has) {
            int32_t row[kStTab];
#pragma unroll
            for (int i = 0; i < kStTab; ++i) row[i] = 0;
            row[ST_FIRST] = first; row[ST_CNT] = cnt; row[ST_SKEW] = sk; row[ST_ND] = nd;
            row[ST_OFF] = of0; row[ST_OFF + 1] = of1; row[ST_OFF + 2] = of2; row[ST_SRC] = sr0; row[ST_SRC + 1] = sr1; row[ST_SRC + 2] = sr2;
            row[ST_KLO] = kl0; row[ST_KLO + 1] = kl1; row[ST_KLO + 2] = kl2; row[ST_KAP] = ka0; row[ST_KAP + 1] = ka1; row[ST_KAP + 2] = ka2;
            row[ST_DT] = dt0; row[ST_DT + 1] = dt1; row[ST_DT + 2] = dt2; row[ST_KHI] = kh0; row[ST_KHI + 1] = kh1; row[ST_KHI + 2] = kh2;
            if (!wx_lane_ok(row, t, !fwd)) atomicOr(&X.flags[9], 1);
        }
        // the wave's chunk range
        int w_lo = has ? sk : 0x7fffffff, w_hi = has ? sk + cnt : -0x7fffffff;
        for (int off = 32; off > 0; off >>= 1) { w_lo = min(w_lo, __shfl_xor(w_lo, off)); w_hi = max(w_hi, __shfl_xor(w_hi, off)); }
        if ((t & 63) == 0) {
            const int nch = w_hi > w_lo ? w_hi - w_lo : 0;
            *reinterpret_cast<int4 *>(X.wtab + (size_t)(wg * 4 + (t >> 6)) * 4) = make_int4(0, nch > 0 ? w_lo : 0, nch, 0);
        }
    }
}

bool grid_lane_tables(hipStream_t st, const GridDims &gd, const Schedule &fwd, const Schedule &bwd, int32_t *ltabF, int32_t *ltabB,
                      int32_t *flagsF, int32_t *flagsB, int32_t *uslot, int32_t *skewF, int32_t *skewB, int32_t *wtabF, int32_t *wtabB, bool wx)
{
    GridPlace g;
    g.nx = gd.nx; g.ny = gd.ny; g.nz = gd.nz; g.nb = fwd.nb;
    g.s2 = fwd.tile_s2; g.ty = fwd.tile_ty; g.tz = fwd.tile_tz;
    g.NY = g.s2 > 0 ? (g.s2 + g.ty - 1) / g.ty : 0;
    g.nslots = fwd.nslots;
    GridLaneArgs F = {fwd.slot2blk, fwd.blk2slot, fwd.sfirst, fwd.scount, fwd.exported, ltabF, flagsF, skewF, wtabF};
    GridLaneArgs Bk = {bwd.slot2blk, bwd.blk2slot, bwd.sfirst, bwd.scount, bwd.exported, ltabB, flagsB, skewB, wtabB};
    // skews, ages of the hand-off values and the waves' chunk ranges too (what k_st_link_pair makes) where the patches are 16 x 16 and
    // the wave-exchange skews are wanted; ILUPP_GRID_LINK=0: by the general kernel
    static const bool link_on = []() { const char *e = getenv("ILUPP_GRID_LINK"); return !(e && atoi(e) == 0); }();
    const bool link = link_on && wx && g.s2 > 0 && g.ty == 16 && g.tz == 16 && gd.nx >= 3 && gd.ny >= 3;
    hipLaunchKernelGGL(k_grid_lanes, dim3((unsigned)(fwd.nslots / kThreads), 2), dim3(kThreads), 0, st, g, F, Bk, uslot, link ? 1 : 0);
    ILUPP_HIP(hipGetLastError());
    return link;
}

// ---------------------------------------------------------------------------------------------
// Everything else the static analysis makes from the lane tables of a box grid in 16 x 16 patches -- the waves' first chunks (st.hip:
// st_scan_body), where a lane's rows sit in the other schedule's records and where their upper entries go as transposed entries
// (k_st_scat), the direct-feed fields (st_common.h: sd_tab_lane), the forward <-> backward slot maps (k_st_inv_ysrc) and the exchange
// layout (k_st_xch_pair) -- in closed form from the lane's place: no table of another lane is read, nothing is scanned.  One launch,
// block (w, d): workgroup w of the forward (d = 0) / backward (d = 1) schedule.  Both schedules place their blocks the same way (the
// backward one on the mirrored grid), so one set of formulas serves both.
// ---------------------------------------------------------------------------------------------
struct GridGeom { int32_t nx, ny, nz, nb, NY, NZ, nyl, nzl; };                  // nyl / nzl: lines of the last patch in y / z

__device__ __forceinline__ int gt_skew(const int t) { return (t & 15) + (t >> 4) + (t >> 6) * (kWrLag - 1); }
__device__ __forceinline__ int gt_wave_nch(const GridGeom &g, const int nyt, const int nzt, const int w)
{
    if (4 * w >= nzt) return 0;
    const int zmax = min(4 * w + 3, nzt - 1);
    return g.nx + nyt - 1 + zmax + (zmax >> 2) * (kWrLag - 1) - (4 * w + w * (kWrLag - 1));
}
__device__ __forceinline__ int gt_tile_chunks(const GridGeom &g, const int nyt, const int nzt)
{
    return gt_wave_nch(g, nyt, nzt, 0) + gt_wave_nch(g, nyt, nzt, 1) + gt_wave_nch(g, nyt, nzt, 2) + gt_wave_nch(g, nyt, nzt, 3);
}
// first chunk of wave w of workgroup T (row-major over the NY x NZ patches; only the last row / column of patches is partial)
__device__ __forceinline__ int gt_wave_base(const GridGeom &g, const int T, const int w)
{
    const int Y = T % g.NY, Z = T / g.NY;
    const int nyt = Y == g.NY - 1 ? g.nyl : 16, nzt = Z == g.NZ - 1 ? g.nzl : 16;
    const long row_full = (long)(g.NY - 1) * gt_tile_chunks(g, 16, 16) + gt_tile_chunks(g, g.nyl, 16);
    long base = (long)Z * row_full + (long)Y * gt_tile_chunks(g, 16, nzt);
    for (int q = 0; q < w; ++q) base += gt_wave_nch(g, nyt, nzt, q);
    return (int)base;
}
__device__ __forceinline__ int gt_wave_tmin(const int w) { return 4 * w + w * (kWrLag - 1); }
// exchange of workgroup T: exported lanes rounded up to 16, steps (first step 0), entries
__device__ __forceinline__ void gt_xch(const GridGeom &g, const int Y, const int Z, int *E, int *steps)
{
    const int nyt = Y == g.NY - 1 ? g.nyl : 16, nzt = Z == g.NZ - 1 ? g.nzl : 16;
    const bool ey = Y < g.NY - 1, ez = Z < g.NZ - 1;
    const int ex = (ey ? nzt : 0) + (ez ? nyt : 0) - ((ey && ez) ? 1 : 0);
    int thi = 0;
    for (int w = 0; 4 * w < nzt; ++w) thi = max(thi, gt_wave_tmin(w) + gt_wave_nch(g, nyt, nzt, w));
    *E = (ex + 15) & ~15; *steps = thi;
}
__device__ __forceinline__ long gt_xch_offset(const GridGeom &g, const int T)
{
    const int Y = T % g.NY, Z = T / g.NY;
    int E, st;
    long row_full = 0;
    gt_xch(g, 0, 0, &E, &st);                      // (a patch that is last in neither direction: only counted when there is one)
    if (g.NY > 1 && g.NZ > 1) row_full += (long)(g.NY - 1) * E * st;
    if (g.NZ > 1) { gt_xch(g, g.NY - 1, 0, &E, &st); row_full += (long)E * st; }
    long off = (long)Z * row_full;
    if (Y > 0) { gt_xch(g, 0, Z, &E, &st); off += (long)Y * E * st; }
    return off;
}

struct GridScatArgs { int32_t *ltabF, *ltabB, *wtabF, *wtabB, *ysrc, *rtab, *xeF, *xeB, *xwF, *xwB, *flagsF, *flagsB, *tot; };

__global__ void __launch_bounds__(kThreads)
k_grid_scat(const GridPlace gp, const GridGeom g, const GridScatArgs X)
{
    const bool fwd = blockIdx.y == 0;
    const int wg = blockIdx.x, t = threadIdx.x, w = t >> 6;
    const int slot = wg * kThreads + t;
    const int bs = grid_block_at(slot, gp);
    const int b = bs < 0 ? -1 : (fwd ? bs : g.nb - 1 - bs);
    const bool has = b >= 0;
    const int Y = wg % g.NY, Z = wg / g.NY;
    // the waves' chunk tables get their first chunks (scan), the totals go to the flags
    if ((t & 63) == 0) (fwd ? X.wtabF : X.wtabB)[(size_t)(wg * 4 + w) * 4] = gt_wave_base(g, wg, w);
    if (wg == 0 && t == 0) {
        int32_t *fl = fwd ? X.flagsF : X.flagsB;
        const int last = g.NY * g.NZ - 1;
        const int nyt = g.nyl, nzt = g.nzl;
        fl[1] = gt_wave_base(g, last, 0) + gt_tile_chunks(g, nyt, nzt);
        int mx = 0;
        for (int cy = 0; cy < 2; ++cy)
            for (int cz = 0; cz < 2; ++cz) {
                if ((cy == 0 && g.NY < 2) || (cz == 0 && g.NZ < 2)) continue;          // (no patch of that class)
                for (int q = 0; q < 4; ++q) mx = max(mx, gt_wave_nch(g, cy ? g.nyl : 16, cz ? g.nzl : 16, q));
            }
        fl[2] = mx;
    }
    // the exchange: ordinal of an exported lane, {exported lanes rounded up to 16, first step, steps, first entry} of the workgroup
    {
        const bool ey = Y < g.NY - 1, ez = Z < g.NZ - 1;
        const int yl = t & 15, zl = t >> 4;
        const bool ex = has && ((ey && yl == 15) || (ez && zl == 15));
        (fwd ? X.xeF : X.xeB)[slot] = ex ? zl * (ey ? 1 : 0) + ((ez && zl == 15) ? yl : 0) : -1;
        if (t == 0) {
            int E, st;
            gt_xch(g, Y, Z, &E, &st);
            const int off = (int)gt_xch_offset(g, wg);
            *reinterpret_cast<int4 *>((fwd ? X.xwF : X.xwB) + (size_t)wg * 4) = make_int4(E, 0, st, off);
            // (what k_st_xch_pair reports to the host: first entry and size of the last workgroup)
            if (wg == g.NY * g.NZ - 1) { X.tot[fwd ? 0 : 2] = off; X.tot[fwd ? 1 : 3] = E * st; }
        }
    }
    if (!fwd) {
        if (!has) X.ysrc[slot] = 0;                  // (a lane without rows: the entry the forward lanes never write)
        return;
    }
    // ---- forward lanes: k_st_scat + sd_tab_lane + k_st_inv_ysrc
    int32_t *T = X.ltabF + (size_t)slot * kStTab;
    int32_t *R = X.rtab + (size_t)slot * 32;
    if (!has) {
        *reinterpret_cast<int4 *>(R) = make_int4(0, 0, 0, 0);
        *reinterpret_cast<int4 *>(T + 24) = make_int4(-1, -1, 0, 0);
        *reinterpret_cast<int4 *>(T + 28) = make_int4(-1, -1, -1, 0);
        return;
    }
    const int y = b % g.ny, z = b / g.ny;
    const int cnt = g.nx, first = b * g.nx, sk = gt_skew(t);
    const bool own = g.nx > 1;
    // this line in the backward schedule
    const int su = grid_place(g.nb - 1 - b, gp), tu = su & 255, wgu = su >> 8, wu = su >> 6;
    const int skB = gt_skew(tu);
    const int baseF = gt_wave_base(g, wg, w), baseU = gt_wave_base(g, wgu, (su >> 6) & 3);
    X.ysrc[su] = (baseF + (cnt - 1 + sk - gt_wave_tmin(w))) * 64 + (t & 63);
    const int up0 = (baseU + (cnt - 1 + skB - gt_wave_tmin(wu & 3))) * 128 + (su & 63);
    // forward template (ascending offset: line below, line before, own chain) and backward template (own chain, line after, line above)
    const bool f0 = z > 0, f1 = y > 0, f2 = own, b0 = own, b1 = y < g.ny - 1, b2 = z < g.nz - 1;
    const int sxy = g.nx * g.ny;
    const int nd = (f0 ? 1 : 0) + (f1 ? 1 : 0) + (f2 ? 1 : 0), nu = (b0 ? 1 : 0) + (b1 ? 1 : 0) + (b2 ? 1 : 0);
    // where the upper entries of this lane's rows go: into the record of row r + o, as the transposed entry of its elimination with row r
    // (candidate c of the backward template -> chain cs, its dependency index jc with offset -o, the index kc of row r + o in that chain)
    auto scat_of = [&](const int c) -> int {
        int cs, jc, kc;
        if (c == 0) { cs = slot; jc = nd - 1; kc = 1; }                                  // own chain: column r + 1
        else if (c == 1) { cs = grid_place(b + 1, gp); jc = f0 ? 1 : 0; kc = 0; }        // line (y + 1, z): its dependency "line before"
        else { cs = grid_place(b + g.ny, gp); jc = 0; kc = 0; }                         // line (y, z + 1): its dependency "line below"
        const int cw = cs >> 6;
        const int chunk0 = gt_wave_base(g, cs >> 8, cw & 3) + (kc + gt_skew(cs & 255) - gt_wave_tmin(cw & 3));
        return (chunk0 * 256 + (2 + (jc >> 1)) * 64 + (cs & 63)) * 2 + (jc & 1);
    };
    const int j0 = b0 ? 0 : (b1 ? 1 : (b2 ? 2 : -1));
    const int j1 = b0 ? (b1 ? 1 : (b2 ? 2 : -1)) : ((b1 && b2) ? 2 : -1);
    const int j2 = (b0 && b1 && b2) ? 2 : -1;
    const int sc0 = j0 >= 0 ? scat_of(j0) : -1, sc1 = j1 >= 0 ? scat_of(j1) : -1, sc2 = j2 >= 0 ? scat_of(j2) : -1;
    // which entry right of the pivot row's diagonal is the transposed entry of forward dependency j (sd_tab_lane: ST_Q)
    const int qz = (own ? 1 : 0) + (y < g.ny - 1 ? 1 : 0);       // line (y, z - 1): its entry + sxy comes after + 1 and + nx
    const int qy = own ? 1 : 0;                                   // line (y - 1, z): its entry + nx comes after + 1
    const int i0 = f0 ? 0 : (f1 ? 1 : (f2 ? 2 : -1));
    const int i1 = f0 ? (f1 ? 1 : (f2 ? 2 : -1)) : ((f1 && f2) ? 2 : -1);
    const int i2 = (f0 && f1 && f2) ? 2 : -1;
#define GS_Q(i_) ((i_) == 0 ? qz : ((i_) == 1 ? qy : ((i_) == 2 ? 0 : -1)))
    const int q0 = GS_Q(i0), q1 = GS_Q(i1), q2 = GS_Q(i2);
#undef GS_Q
    const int p0 = (int)grid_row_start(0, y, z, GridDims{g.nx, g.ny, g.nz});
    const int dfl = nu | ((own ? 1 : 0) << 2) | ((own ? 1 : 0) << 3) | ((nd + 1 + nu) << 4);
    // lane table: UP0, SCAT, P0, DFL, Q (the rest stands since k_grid_lanes)
    T[ST_UP0] = up0;
    T[ST_SCAT] = sc0;
    *reinterpret_cast<int4 *>(T + 24) = make_int4(sc1, sc2, p0, dfl);
    *reinterpret_cast<int4 *>(T + 28) = make_int4(q0, q1, q2, 0);
    // the rows pass' / transposed records' view of the lane (k_st_scat: rtab)
#define GS_OFF_F(i_) ((i_) == 0 ? -sxy : ((i_) == 1 ? -g.nx : ((i_) == 2 ? -1 : 0)))
#define GS_OFF_B(i_) ((i_) == 0 ? 1 : ((i_) == 1 ? g.nx : ((i_) == 2 ? sxy : 0)))
    const int oF0 = GS_OFF_F(i0), oF1 = GS_OFF_F(i1), oF2 = GS_OFF_F(i2), oB0 = GS_OFF_B(j0), oB1 = GS_OFF_B(j1), oB2 = GS_OFF_B(j2);
#undef GS_OFF_F
#undef GS_OFF_B
    const int klF0 = i0 == 2 ? 1 : 0, klF1 = i1 == 2 ? 1 : 0, klF2 = i2 == 2 ? 1 : 0;       // (own chain: every row but the first)
    const int klB0 = j0 == 0 ? 1 : 0, klB1 = 0, klB2 = 0;
    const int khF0 = i0 >= 0 ? cnt : 0, khF1 = i1 >= 0 ? cnt : 0, khF2 = i2 >= 0 ? cnt : 0;
    const int khB0 = j0 >= 0 ? cnt : 0, khB1 = j1 >= 0 ? cnt : 0, khB2 = j2 >= 0 ? cnt : 0;
    // the forward dependency's producer lane when it is one of the 64 lanes of the same wave: lane | (k' - k + 128) << 8
    auto samewave = [&](const int i) -> int {
        if (i < 0) return -1;
        if (i == 2) return (t & 63) | ((-1 + 128) << 8);
        const int os = grid_place(i == 0 ? b - g.ny : b - 1, gp);
        return ((os >> 6) == (slot >> 6)) ? ((os & 63) | (128 << 8)) : -1;
    };
    int4 *R4 = reinterpret_cast<int4 *>(R);
    R4[0] = make_int4(first, cnt, sk, nd);
    R4[1] = make_int4(oF0, oF1, oF2, nu);
    R4[2] = make_int4(oB0, oB1, oB2, up0);
    R4[3] = make_int4(klF0, klF1, klF2, 0);
    R4[4] = make_int4(khF0, khF1, khF2, 0);
    R4[5] = make_int4(klB0, klB1, klB2, 0);
    R4[6] = make_int4(khB0, khB1, khB2, 0);
    R4[7] = make_int4(samewave(i0), samewave(i1), samewave(i2), 0);
}

void grid_scat_tables(hipStream_t st, const GridDims &gd, const Schedule &fwd, int32_t *ltabF, int32_t *ltabB, int32_t *wtabF, int32_t *wtabB,
                      int32_t *ysrc, int32_t *rtab, int32_t *xeF, int32_t *xeB, int32_t *xwF, int32_t *xwB, int32_t *flagsF, int32_t *flagsB,
                      int32_t *tot)
{
    GridPlace gp;
    gp.nx = gd.nx; gp.ny = gd.ny; gp.nz = gd.nz; gp.nb = fwd.nb;
    gp.s2 = fwd.tile_s2; gp.ty = fwd.tile_ty; gp.tz = fwd.tile_tz;
    gp.NY = (gp.s2 + gp.ty - 1) / gp.ty;
    gp.nslots = fwd.nslots;
    GridGeom g;
    g.nx = gd.nx; g.ny = gd.ny; g.nz = gd.nz; g.nb = fwd.nb;
    g.NY = (gd.ny + 15) / 16; g.NZ = (gd.nz + 15) / 16;
    g.nyl = gd.ny - 16 * (g.NY - 1); g.nzl = gd.nz - 16 * (g.NZ - 1);
    GridScatArgs X = {ltabF, ltabB, wtabF, wtabB, ysrc, rtab, xeF, xeB, xwF, xwB, flagsF, flagsB, tot};
    hipLaunchKernelGGL(k_grid_scat, dim3((unsigned)(fwd.nslots / kThreads), 2), dim3(kThreads), 0, st, gp, g, X);
    ILUPP_HIP(hipGetLastError());
}

// Host: what the lane-table kernels will find for a box grid in 16 x 16 patches (st.hip: st_link_body -- skew of lane (y, z) of a patch =
// y + z + (z / 4)(kWrLag - 1): one step per neighbour, one more across a wave's border --, st_scan_body, k_st_xch_pair).  Both
// schedules have the same numbers (the backward one is the mirror image).
bool grid_predict_sizes(const GridDims &g, int ty, int tz, int64_t *nchunks, int32_t *maxch, int64_t *xoff_last, int32_t *xsz_last)
{
    if (ty != 16 || tz != 16 || g.nz < 2) return false;
    const int lagx = 1;                      // kWrLag - 1 (st_common.h)
    const int NY = (g.ny + 15) / 16, NZ = (g.nz + 15) / 16;
    int64_t total = 0, xrun = 0;
    int32_t mx = 0, xlast = 0;
    for (int Z = 0; Z < NZ; ++Z) {
        for (int Y = 0; Y < NY; ++Y) {
            const int nyt = g.ny - 16 * Y < 16 ? g.ny - 16 * Y : 16, nzt = g.nz - 16 * Z < 16 ? g.nz - 16 * Z : 16;
            int thi = 0;
            for (int w = 0; 4 * w < nzt; ++w) {
                const int zmax = 4 * w + 3 < nzt - 1 ? 4 * w + 3 : nzt - 1;
                const int tmin = 4 * w + w * lagx;
                const int smax = nyt - 1 + zmax + (zmax / 4) * lagx;
                const int nch = smax + g.nx - tmin;
                total += nch;
                if (nch > mx) mx = nch;
                if (tmin + nch > thi) thi = tmin + nch;
            }
            const bool ey = Y < NY - 1, ez = Z < NZ - 1;
            const int ex = (ey ? nzt : 0) + (ez ? nyt : 0) - ((ey && ez) ? 1 : 0);
            const int E = (ex + 15) & ~15;
            const int64_t xsz = (int64_t)E * thi;            // (first step of a patch: 0)
            if (xrun + xsz > 0x7fffffffLL) return false;
            if (Z == NZ - 1 && Y == NY - 1) xlast = (int32_t)xsz; else xrun += xsz;
        }
    }
    *nchunks = total; *maxch = mx; *xoff_last = xrun; *xsz_last = xlast;
    return total > 0 && total < 0x7fffffffLL;
}

void grid_check_launch(hipStream_t side, const DevMat &A, const GridDims &g, int32_t *d_bad)
{
    unsigned gb = (unsigned)(((int64_t)A.n + 255) / 256);
    if (gb > (1u << 20)) gb = 1u << 20;
    hipLaunchKernelGGL(k_grid_check, dim3(gb), dim3(256), 0, side, A.n, (long long)A.nnz, g, A.ptr, A.idx, d_bad);
    ILUPP_HIP(hipGetLastError());
}

// What ilu0_symbolic_and_schedule + choose_tiling_pair + finish_chains leave behind, for a matrix that IS the guessed grid (the proof
// runs next to what follows).  The placement of the lines on workgroups is the one choose_tiling makes for a (1, ny) line grid.
void grid_schedules(hipStream_t st, const DevMat &A, const GridDims &g, DevMat *L, DevMat *U, Schedule *fwd, Schedule *bwd,
                    int32_t *max_row_len, int max_wgs)
{
    const int32_t n = A.n;
    const int64_t links = grid_links(g);
    L->n = U->n = n; L->is_csr = U->is_csr = true; L->owns = U->owns = true;
    L->nnz = (int64_t)n + links;            // strictly lower + unit diagonal (ILU0.hpp:93)
    U->nnz = A.nnz - links;
    if (max_row_len) *max_row_len = 1 + 2 * ((g.nx > 1 ? 1 : 0) + (g.ny > 1 ? 1 : 0) + (g.nz > 1 ? 1 : 0));
    const int32_t nb = g.ny * g.nz;
    for (Schedule *s : {fwd, bwd}) {
        s->nb = nb; s->B = g.nx;
        ILUPP_HIP(pool_malloc(&s->start, sizeof(int32_t) * (size_t)(nb + 1)));
        s->chains = s->chains_pre = true; s->ragged = 0;
        s->tile_s2 = s->tile_ty = s->tile_tz = 0;
    }
    hipLaunchKernelGGL(k_grid_starts, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0, st, n, g.nx, nb, fwd->start, bwd->start);
    ILUPP_HIP(hipGetLastError());
    // patches of 16 x 16 lines, as square as the grid allows (schedule.hip: tiling_decide)
    if (g.nz >= 2 && getenv("ILUPP_NO_TILES") == nullptr) {
        const int s2 = g.ny, nbz = g.nz;
        int ty = 16, tz = 16;
        while (ty > s2 && ty > 1) { ty >>= 1; tz <<= 1; }
        while (tz > nbz && tz > 1) { tz >>= 1; ty <<= 1; }
        if (ty <= s2 && ty * tz == kThreads) {
            const int NY = (s2 + ty - 1) / ty, NZ = (nbz + tz - 1) / tz;
            if ((long)NY * NZ <= max_wgs)
                for (Schedule *s : {fwd, bwd}) { s->tile_s2 = s2; s->tile_ty = ty; s->tile_tz = tz; }
        }
    }
}

// The backward schedule of L^T of an LL^T object on a box grid (icholt_grid.hip): what count_cuts_and_schedule + choose_tiling find in
// the pattern -- one block per x-line, patches of 16 x 16 lines -- from the dimensions.  false: the general pass would cut differently
// (more rows than lanes x line length), nothing was made.
bool grid_llt_schedule(hipStream_t st, int32_t n, const GridDims &g, int max_lanes, Schedule *bwd)
{
    if (((int64_t)n + max_lanes - 1) / max_lanes > g.nx || g.nx <= 4) return false;
    const int32_t nb = g.ny * g.nz;
    bwd->nb = nb; bwd->B = g.nx;
    ILUPP_HIP(pool_malloc(&bwd->start, sizeof(int32_t) * (size_t)(nb + 1)));
    bwd->tile_s2 = bwd->tile_ty = bwd->tile_tz = 0;
    hipLaunchKernelGGL(k_grid_starts, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0, st, n, g.nx, nb, bwd->start, bwd->start);
    ILUPP_HIP(hipGetLastError());
    // patches of 16 x 16 lines, as square as the grid allows (schedule.hip: tiling_decide; grid_schedules above)
    if (g.nz >= 2 && getenv("ILUPP_NO_TILES") == nullptr) {
        const int s2 = g.ny, nbz = g.nz;
        int ty = 16, tz = 16;
        while (ty > s2 && ty > 1) { ty >>= 1; tz <<= 1; }
        while (tz > nbz && tz > 1) { tz >>= 1; ty <<= 1; }
        if (ty <= s2 && ty * tz == kThreads) {
            const int NY = (s2 + ty - 1) / ty, NZ = (nbz + tz - 1) / tz;
            if ((long)NY * NZ <= max_lanes / kThreads) { bwd->tile_s2 = s2; bwd->tile_ty = ty; bwd->tile_tz = tz; }
        }
    }
    return true;
}


// ---------------------------------------------------------------------------------------------
// The row-major copy of an LL^T factor whose pattern is the grid's lower triangle, stored by columns (ICholT(0, 0) on the speculative
// static path, api.hip): row i of the copy is l(i, i - nx ny), l(i, i - nx), l(i, i - 1), l(i, i) -- column j holds l(j, j), then the
// entries below it in the order + 1, + nx, + nx ny as far as the grid has them, so where l(i, j) sits in column j follows from j's
// coordinates.  No sort, no atomics; every fetched index is compared with the row it must be (a factor whose pattern is not the
// grid's raises the flag and the general transposition runs).
// ---------------------------------------------------------------------------------------------
__global__ void k_llt_grid_rowcount(const int32_t n, const GridDims g, int32_t *__restrict__ cnt)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { cnt[n] = 0; return; }
    const unsigned r = (unsigned)i, l = r / (unsigned)g.nx, x = r - l * (unsigned)g.nx, z = l / (unsigned)g.ny, y = l - z * (unsigned)g.ny;
    cnt[i] = 1 + (x > 0) + (y > 0) + (z > 0);
}
__global__ void k_llt_grid_rows(const int32_t n, const GridDims g, const int32_t *__restrict__ cptr, const int32_t *__restrict__ cidx,
                                const double *__restrict__ cval, const int32_t *__restrict__ rptr, int32_t *__restrict__ ridx,
                                double *__restrict__ rval, int32_t *__restrict__ bad)
{
    const long long ii = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ii >= n) return;
    const int i = (int)ii;
    const unsigned r = (unsigned)i, l = r / (unsigned)g.nx, x = r - l * (unsigned)g.nx, z = l / (unsigned)g.ny, y = l - z * (unsigned)g.ny;
    const int sxy = g.nx * g.ny;
    int out = rptr[i];
    bool ok = true;
    if (z > 0) {                                   // column j = (x, y, z - 1): behind its + 1 and + nx entries
        const int j = i - sxy, q = cptr[j] + 1 + ((int)x < g.nx - 1) + ((int)y < g.ny - 1);
        ok = ok && cidx[q] == i;
        ridx[out] = j; rval[out] = cval[q]; ++out;
    }
    if (y > 0) {                                   // column j = (x, y - 1, z): behind its + 1 entry
        const int j = i - g.nx, q = cptr[j] + 1 + ((int)x < g.nx - 1);
        ok = ok && cidx[q] == i;
        ridx[out] = j; rval[out] = cval[q]; ++out;
    }
    if (x > 0) {
        const int j = i - 1, q = cptr[j] + 1;
        ok = ok && cidx[q] == i;
        ridx[out] = j; rval[out] = cval[q]; ++out;
    }
    {
        const int q = cptr[i];
        ok = ok && cidx[q] == i && cptr[i + 1] - q == 1 + ((int)x < g.nx - 1) + ((int)y < g.ny - 1) + ((int)z < g.nz - 1);
        ridx[out] = i; rval[out] = cval[q];
    }
    if (__builtin_amdgcn_ballot_w64(!ok) != 0 && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}
// T's arrays are allocated by the caller (transpose_storage's sizes); false: the factor's pattern is not the grid's
bool llt_grid_rows(hipStream_t st, const DevMat &Lc, const GridDims &g, DevMat *T)
{
    const int32_t n = Lc.n;
    if ((int64_t)g.nx * g.ny * g.nz != n || n < 2) return false;
    int32_t *cnt = nullptr, *flag = nullptr;
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&cnt, sizeof(int32_t) * ((size_t)n + 1)));
    ILUPP_HIP(pool_malloc(&flag, 16));
    ILUPP_HIP(hipMemsetAsync(flag, 0, 16, st));
    hipLaunchKernelGGL(k_llt_grid_rowcount, dim3((unsigned)(((int64_t)n + 1 + 255) / 256)), dim3(256), 0, st, n, g, cnt);
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, T->ptr, n + 1, st));
    ILUPP_HIP(pool_malloc(&tmp, tb > 0 ? tb : 16));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cnt, T->ptr, n + 1, st));
    hipLaunchKernelGGL(k_llt_grid_rows, dim3((unsigned)(((int64_t)n + 255) / 256)), dim3(256), 0, st, n, g, Lc.ptr, Lc.idx, Lc.val, T->ptr, T->idx,
                       T->val, flag);
    int32_t h[2] = {1, 0};
    ILUPP_HIP(d2h_async(st, h, flag, sizeof(int32_t)));
    ILUPP_HIP(d2h_async(st, h + 1, T->ptr + n, sizeof(int32_t)));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(tmp)); ILUPP_HIP(pool_free(cnt)); ILUPP_HIP(pool_free(flag));
    return h[0] == 0 && (int64_t)h[1] == Lc.nnz;
}

}  // namespace ilupp
