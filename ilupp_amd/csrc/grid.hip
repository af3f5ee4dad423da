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
#include "common.h"

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

// entries before row r = (x, y, z): seven per row minus the neighbours that fall outside the box
__device__ __forceinline__ long long grid_row_start(const int x, const int y, const int z, const GridDims &g)
{
    const long long nx = g.nx, ny = g.ny;
    const long long r = x + nx * (y + ny * (long long)z);
    const bool z0 = z == 0, z1 = z == g.nz - 1, y0 = y == 0, y1 = y == g.ny - 1;
    long long miss = 0;
    // whole planes below: the ends of every line, the first and the last line, and all of plane 0 (no plane below it)
    miss += (long long)z * (2 * ny + 2 * nx) + (z > 0 ? nx * ny : 0);
    // whole lines of this plane before line y
    miss += 2LL * y + (y > 0 ? nx : 0) + (long long)y * nx * ((z0 ? 1 : 0) + (z1 ? 1 : 0));
    // rows of this line before x
    miss += (x > 0 ? 1 : 0) + (long long)x * ((y0 ? 1 : 0) + (y1 ? 1 : 0) + (z0 ? 1 : 0) + (z1 ? 1 : 0));
    return 7 * r - miss;
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

struct GridLaneArgs { int32_t *slot2blk, *blk2slot, *sfirst, *scount, *exported, *ltab, *flags; };

__global__ void __launch_bounds__(kThreads)
k_grid_lanes(const GridPlace g, const GridLaneArgs F, const GridLaneArgs Bk, int32_t *__restrict__ uslot)
{
    const bool fwd = blockIdx.y == 0;
    const GridLaneArgs &X = fwd ? F : Bk;
    const int wg = blockIdx.x, t = threadIdx.x;
    const int slot = wg * kThreads + t;
    if (slot >= g.nslots) return;
    const int bs = grid_block_at(slot, g);
    const int b = bs < 0 ? -1 : (fwd ? bs : g.nb - 1 - bs);
    X.slot2blk[slot] = b;
    int32_t *T = X.ltab + (size_t)slot * kStTab;
    if (b < 0) {
        X.sfirst[slot] = 0; X.scount[slot] = 0; X.exported[slot] = 0;
        if (fwd) uslot[slot] = -1;
#pragma unroll
        for (int i = 0; i < kStTab; ++i) T[i] = 0;
        T[ST_DT] = T[ST_DT + 1] = T[ST_DT + 2] = 1;
        T[ST_SCAT] = T[ST_SCAT + 1] = T[ST_SCAT + 2] = -1;
        return;
    }
    X.blk2slot[b] = slot;
    const int cnt = g.nx;
    const int first = fwd ? b * g.nx : (b + 1) * g.nx - 1;       // first row in processing order
    X.sfirst[slot] = first; X.scount[slot] = cnt;
    const int y = b % g.ny, z = b / g.ny;
    // the lines this one depends on, ascending column offset (= the reference's elimination / accumulation order), and the lines that
    // depend on it (a slot some OTHER workgroup reads is exported)
    // (every field is stored as it is found -- the table row is the only array: an array of their own would live in scratch memory)
    int nd = 0, bad = 0, ngh = 0;
    const int sxy = g.nx * g.ny;
#pragma unroll
    for (int i = ST_OFF; i < kStTab; ++i) T[i] = 0;
    T[ST_DT] = T[ST_DT + 1] = T[ST_DT + 2] = 1;
    T[ST_SCAT] = T[ST_SCAT + 1] = T[ST_SCAT + 2] = -1;
    bool ex = false;
#define GRID_DEP(o_, bo_)                                                                              \
    do {                                                                                               \
        const int os_ = grid_place(fwd ? (bo_) : g.nb - 1 - (bo_), g);                                 \
        int sw_;                                                                                       \
        if ((os_ >> 8) == wg) sw_ = ST_LOCAL | (os_ << 2);                                             \
        else { sw_ = ST_GHOST | (os_ << 2); ++ngh; if ((os_ >> 8) >= wg) bad = 1; }                    \
        T[ST_OFF + nd] = (o_); T[ST_SRC + nd] = sw_; T[ST_KHI + nd] = cnt; ++nd;                       \
    } while (0)
#define GRID_OWN(o_)                                                                                   \
    do {                                                                                               \
        T[ST_OFF + nd] = (o_); T[ST_SRC + nd] = ST_OWN | (slot << 2); T[ST_KAP + nd] = -1;             \
        T[ST_KLO + nd] = 1; T[ST_KHI + nd] = cnt; ++nd;                                                \
    } while (0)
#define GRID_READER(bo_) do { if ((grid_place(fwd ? (bo_) : g.nb - 1 - (bo_), g) >> 8) != wg) ex = true; } while (0)
    if (fwd) {
        if (z > 0) GRID_DEP(-sxy, b - g.ny);
        if (y > 0) GRID_DEP(-g.nx, b - 1);
        if (g.nx > 1) GRID_OWN(-1);
        if (y < g.ny - 1) GRID_READER(b + 1);
        if (z < g.nz - 1) GRID_READER(b + g.ny);
    } else {
        if (g.nx > 1) GRID_OWN(1);
        if (y < g.ny - 1) GRID_DEP(g.nx, b + 1);
        if (z < g.nz - 1) GRID_DEP(sxy, b + g.ny);
        if (y > 0) GRID_READER(b - 1);
        if (z > 0) GRID_READER(b - g.ny);
    }
#undef GRID_DEP
#undef GRID_OWN
#undef GRID_READER
    X.exported[slot] = ex ? 1 : 0;
    if (nd == 3 && ngh == 3) bad = 1;
    T[ST_FIRST] = first; T[ST_CNT] = cnt; T[ST_SKEW] = 0; T[ST_ND] = nd;
    if (bad) atomicOr(&X.flags[0], 2);
    // the slot of the backward schedule that owns the same line
    if (fwd) uslot[slot] = grid_place(g.nb - 1 - b, g);
}

void grid_lane_tables(hipStream_t st, const GridDims &gd, const Schedule &fwd, const Schedule &bwd, int32_t *ltabF, int32_t *ltabB,
                      int32_t *flagsF, int32_t *flagsB, int32_t *uslot)
{
    GridPlace g;
    g.nx = gd.nx; g.ny = gd.ny; g.nz = gd.nz; g.nb = fwd.nb;
    g.s2 = fwd.tile_s2; g.ty = fwd.tile_ty; g.tz = fwd.tile_tz;
    g.NY = g.s2 > 0 ? (g.s2 + g.ty - 1) / g.ty : 0;
    g.nslots = fwd.nslots;
    GridLaneArgs F = {fwd.slot2blk, fwd.blk2slot, fwd.sfirst, fwd.scount, fwd.exported, ltabF, flagsF};
    GridLaneArgs Bk = {bwd.slot2blk, bwd.blk2slot, bwd.sfirst, bwd.scount, bwd.exported, ltabB, flagsB};
    hipLaunchKernelGGL(k_grid_lanes, dim3((unsigned)(fwd.nslots / kThreads), 2), dim3(kThreads), 0, st, g, F, Bk, uslot);
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

}  // namespace ilupp
