// ilupp_amd/csrc/schedule.hip -- static products of the analysis phase that the persistent sweeps
// stream instead of chasing pointers (gfx950).
//
//  * slot tables: which row block every lane (slot = workgroup*256 + lane) of the persistent grid owns;
//  * solve descriptors: one int32 per stored entry of a triangular factor, replacing the column index
//    in the sweep: (owner slot << 15 | position of the wanted row in its owner's processing order), so
//    a consumer knows at once whether its producer sits in the same workgroup (LDS ring) or not;
//  * ILU(0) update program: per row, the list of eliminations and, for each, where the pivot row's
//    matching entries live -- the symbolic half of sparse_vec_update (reference ILU0.hpp:8-23) done
//    once, so the numeric kernel has no merge loop and no dependent index loads.
//
// All of it is integer streaming work with thread-per-row kernels over independent rows.
#include <stdlib.h>

#include <utility>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "st_common.h"

namespace ilupp {

// Placement of row blocks on the persistent grid.
//  identity: slot s owns block s (in sweep order).
//  tiled:    the blocks form a 2-D grid (b % s2, b / s2) -- on a lexicographic 3-D mesh: (y, z) of the
//            x-line -- and a workgroup owns a ty x tz patch of it, so that both neighbour dependencies of
//            a block stay inside the workgroup (LDS hand-off) except on two faces of the patch.  A
//            dependency path then crosses O(sqrt(#workgroups)) workgroup borders instead of O(#workgroups).
struct Tiling { int s2, ty, tz, NY, nb; };

__device__ __forceinline__ int tiled_block_of(int slot, const Tiling &t)
{
    const int T = slot / kThreads, lane = slot % kThreads;
    const int by = (T % t.NY) * t.ty + lane % t.ty;
    const int bz = (T / t.NY) * t.tz + lane / t.ty;
    if (by >= t.s2 || lane / t.ty >= t.tz) return -1;
    const long bs = (long)bz * t.s2 + by;
    return bs < t.nb ? (int)bs : -1;
}

__global__ void k_slot_tables(int32_t nb, int32_t nslots, int fwd, Tiling til, const int32_t *__restrict__ start,
                              int32_t *__restrict__ slot2blk, int32_t *__restrict__ blk2slot,
                              int32_t *__restrict__ sfirst, int32_t *__restrict__ scount, int32_t *__restrict__ exported)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    exported[s] = 0;                             // (nobody reads this slot across a workgroup border yet; a memset of its own was one launch more)
    // block index in SWEEP order (forward: b, backward: nb-1-b)
    int bs = (til.s2 > 0) ? tiled_block_of(s, til) : (s < nb ? s : -1);
    const int b = bs < 0 ? -1 : (fwd ? bs : nb - 1 - bs);
    slot2blk[s] = b;
    if (b >= 0) {
        blk2slot[b] = s;
        const int lo = start[b], hi = start[b + 1];
        sfirst[s] = fwd ? lo : hi - 1;       // first row in processing order
        scount[s] = hi - lo;
    } else {
        sfirst[s] = 0;
        scount[s] = 0;
    }
}

void build_slot_tables(hipStream_t st, Schedule *sch, bool fwd, bool fill)
{
    sch->fwd = fwd;
    Tiling til = {0, 0, 0, 0, sch->nb};
    if (sch->tile_s2 > 0) {
        til.s2 = sch->tile_s2; til.ty = sch->tile_ty; til.tz = sch->tile_tz;
        til.NY = (til.s2 + til.ty - 1) / til.ty;
        const int nbz = (sch->nb + til.s2 - 1) / til.s2;
        const int NZ = (nbz + til.tz - 1) / til.tz;
        sch->nslots = til.NY * NZ * kThreads;
    } else {
        sch->nslots = ((sch->nb + kThreads - 1) / kThreads) * kThreads;
    }
    const size_t bytes = sizeof(int32_t) * (size_t)sch->nslots;
    ILUPP_HIP(pool_malloc(&sch->slot2blk, bytes));
    ILUPP_HIP(pool_malloc(&sch->blk2slot, sizeof(int32_t) * (size_t)(sch->nb > 0 ? sch->nb : 1)));
    ILUPP_HIP(pool_malloc(&sch->sfirst, bytes));
    ILUPP_HIP(pool_malloc(&sch->scount, bytes));
    ILUPP_HIP(pool_malloc(&sch->exported, bytes));
    const size_t gbytes = sizeof(int32_t) * (size_t)(sch->nslots / kThreads) * kGhosts;
    ILUPP_HIP(pool_malloc(&sch->gtab, gbytes));
    if (!fill) return;
    hipLaunchKernelGGL(k_slot_tables, dim3((unsigned)(sch->nslots / kThreads)), dim3(kThreads), 0, st,
                       sch->nb, sch->nslots, fwd ? 1 : 0, til, sch->start, sch->slot2blk, sch->blk2slot, sch->sfirst, sch->scount, sch->exported);
}

// block-level dependency offsets of a sample of blocks (middle row of each sampled block)
__global__ void k_sample_block_offsets(int32_t nsamp, int32_t stride, int32_t B, int32_t nb, int fwd,
                                       const int32_t *__restrict__ start, const int32_t *__restrict__ ptr,
                                       const int32_t *__restrict__ idx, int32_t *__restrict__ offs)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamp) return;
    const int b = (int)(((long)s * stride) % nb);
    int32_t *o = offs + (size_t)s * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = 0;
    if (start[b + 1] <= start[b]) return;
    const int r = start[b] + (start[b + 1] - start[b]) / 2;
    int w = 0;
    for (int q = ptr[r]; q < ptr[r + 1] && w < 8; ++q) {
        const int c = idx[q];
        if (fwd ? (c >= r) : (c <= r)) continue;
        const int d = fwd ? (b - block_of(c, B, nb, start)) : (block_of(c, B, nb, start) - b);
        if (d > 0) o[w++] = d;
    }
}

// Decide whether the block grid has the (1, s2) dependency structure of a lexicographic mesh and pick
// the patch shape.  Purely a placement (performance) decision: any placement is correct.
static bool tiling_wanted(const Schedule *sch)
{
    static const bool off = getenv("ILUPP_NO_TILES") != nullptr;
    return !off && sch->nb >= 2 * kThreads;
}
static int32_t *tiling_sample(hipStream_t st, const int32_t *ptr, const int32_t *idx, const Schedule *sch, bool fwd, std::vector<int32_t> *h)
{
    const int nsamp = sch->nb < 2048 ? sch->nb : 2048;
    const int stride = sch->nb / nsamp > 0 ? sch->nb / nsamp : 1;
    int32_t *d_offs = nullptr;
    ILUPP_HIP(pool_malloc(&d_offs, sizeof(int32_t) * 8 * (size_t)nsamp));
    hipLaunchKernelGGL(k_sample_block_offsets, dim3((unsigned)((nsamp + 255) / 256)), dim3(256), 0, st, nsamp, stride,
                       sch->B, sch->nb, fwd ? 1 : 0, sch->start, ptr, idx, d_offs);
    h->resize((size_t)nsamp * 8);
    ILUPP_HIP(d2h_async(st, h->data(), d_offs, sizeof(int32_t) * h->size()));
    return d_offs;
}
static void tiling_decide(const std::vector<int32_t> &h, Schedule *sch, int max_wgs)
{
    const int nsamp = (int)(h.size() / 8);
    // most frequent offset > 1
    std::vector<std::pair<int, int>> cnt;   // (offset, count), tiny
    int has1 = 0, nonempty = 0;
    for (int s = 0; s < nsamp; ++s) {
        bool any = false;
        for (int k = 0; k < 8; ++k) {
            const int d = h[(size_t)s * 8 + k];
            if (d <= 0) continue;
            any = true;
            if (d == 1) { ++has1; continue; }
            bool found = false;
            for (auto &pr : cnt) if (pr.first == d) { ++pr.second; found = true; break; }
            if (!found && cnt.size() < 64) cnt.push_back({d, 1});
        }
        nonempty += any ? 1 : 0;
    }
    int s2 = 0, best = 0;
    for (auto &pr : cnt) if (pr.second > best) { best = pr.second; s2 = pr.first; }
    if (nonempty == 0 || s2 < 4 || best * 2 < nonempty || has1 * 2 < nonempty) return;
    const int nbz = (sch->nb + s2 - 1) / s2;
    // patch shape ty x tz = 256, as square as the grid allows
    int ty = 16, tz = 16;
    while (ty > s2 && ty > 1) { ty >>= 1; tz <<= 1; }
    while (tz > nbz && tz > 1) { tz >>= 1; ty <<= 1; }
    if (ty > s2 || ty * tz != kThreads) return;
    // (experiment hook: patches of fewer lines than lanes -- the lanes left over own no block)
    if (const char *e = getenv("ILUPP_TILE_TZ")) { const int v = atoi(e); if (v >= 1 && v <= tz) tz = v; }
    const int NY = (s2 + ty - 1) / ty, NZ = (nbz + tz - 1) / tz;
    // every workgroup must be resident at once (a patch may wait on a higher-numbered patch)
    if ((long)NY * NZ > max_wgs) return;
    sch->tile_s2 = s2; sch->tile_ty = ty; sch->tile_tz = tz;
}

void choose_tiling(hipStream_t st, int32_t n, const int32_t *ptr, const int32_t *idx, Schedule *sch, bool fwd, int max_wgs)
{
    (void)n;
    sch->tile_s2 = sch->tile_ty = sch->tile_tz = 0;
    if (!tiling_wanted(sch)) return;
    std::vector<int32_t> h;
    int32_t *d = tiling_sample(st, ptr, idx, sch, fwd, &h);
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(d));
    tiling_decide(h, sch, max_wgs);
}

// forward and backward schedule of the same matrix with one host round trip
void choose_tiling_pair(hipStream_t st, const int32_t *ptr, const int32_t *idx, Schedule *fwd, Schedule *bwd, int max_wgs)
{
    fwd->tile_s2 = fwd->tile_ty = fwd->tile_tz = 0;
    bwd->tile_s2 = bwd->tile_ty = bwd->tile_tz = 0;
    std::vector<int32_t> hf, hb;
    int32_t *df = tiling_wanted(fwd) ? tiling_sample(st, ptr, idx, fwd, true, &hf) : nullptr;
    int32_t *db = tiling_wanted(bwd) ? tiling_sample(st, ptr, idx, bwd, false, &hb) : nullptr;
    // (always one wait: read-backs queued by the caller -- ilu0_symbolic_and_schedule's block verdict -- arrive with it)
    ILUPP_HIP(stream_sync(st));
    if (df) { ILUPP_HIP(pool_free(df)); tiling_decide(hf, fwd, max_wgs); }
    if (db) { ILUPP_HIP(pool_free(db)); tiling_decide(hb, bwd, max_wgs); }
}

// ---------------------------------------------------------------------------------------------
// solve descriptors
// ---------------------------------------------------------------------------------------------
// desc = owner_slot << 15 | kloc, kloc = index of row c in its owner's processing order; -1 on the diagonal.
// Import table of a workgroup: the foreign producer slots its lanes read from, found on a sample of each
// lane's rows (first, middle, last -- the lanes of a mesh-like matrix read the same neighbours in every
// row).  Purely a fast path: a producer that is not in the table is polled directly by its consumer.
__global__ void __launch_bounds__(kThreads)
k_ghost_table(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t B, int32_t nb,
              const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot, int fwd,
              const int32_t *__restrict__ sfirst, const int32_t *__restrict__ scount, int32_t *__restrict__ gtab)
{
    __shared__ int tab[kGhosts];
    const int wg = blockIdx.x, t = threadIdx.x;
    if (t < kGhosts) tab[t] = -1;
    __syncthreads();
    const int slot = wg * kThreads + t;
    const int cnt = scount[slot], first = sfirst[slot];
    const int dr = fwd ? 1 : -1;
    for (int sample = 0; sample < 3 && cnt > 0; ++sample) {
        const int k = sample == 0 ? 0 : (sample == 1 ? cnt / 2 : cnt - 1);
        if (sample > 0 && k == (sample == 1 ? 0 : cnt / 2)) continue;
        const int r = first + dr * k;
        for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
            const int c = idx[q];
            if (c == r) continue;
            const int oslot = blk2slot[block_of(c, B, nb, start)];
            if ((oslot >> 8) == wg) continue;
            unsigned h = ((unsigned)oslot * 0x9E3779B1u) >> 26;
            for (int probe = 0; probe < kGhosts; ++probe, h = (h + 1) & (kGhosts - 1)) {
                const int cur = atomicCAS(&tab[h], -1, oslot);
                if (cur == -1 || cur == oslot) break;
            }
        }
    }
    __syncthreads();
    if (t < kGhosts) gtab[(size_t)wg * kGhosts + t] = tab[t];
}

__global__ void k_make_desc(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                            int32_t B, int32_t nb, int fwd, const int32_t *__restrict__ start,
                            const int32_t *__restrict__ blk2slot, int32_t *__restrict__ desc, int32_t *__restrict__ exported,
                            const int32_t *__restrict__ gtab)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int mywg = blk2slot[block_of(r, B, nb, start)] >> 8;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int c = idx[q];
        if (c == r) { desc[q] = -1; continue; }     // diagonal marker: also delimits the rows of the stream
        const int b = block_of(c, B, nb, start);
        const int kloc = fwd ? (c - start[b]) : (start[b + 1] - 1 - c);
        int oslot = blk2slot[b];
        if ((oslot >> 8) != mywg) {
            exported[oslot] = 1;     // read across a workgroup border: must be stored write-through
            // a producer listed in the import table is named by its ghost lane (sptrsv.hip)
            if (gtab && kloc < 0x7fff) {
                const int32_t *tab = gtab + (size_t)mywg * kGhosts;
                unsigned h = ((unsigned)oslot * 0x9E3779B1u) >> 26;
                for (int probe = 0; probe < kGhosts; ++probe, h = (h + 1) & (kGhosts - 1)) {
                    const int cur = tab[h];
                    if (cur == oslot) { oslot = kGhostBase + (int)h; break; }
                    if (cur == -1) break;
                }
            }
        }
        desc[q] = (oslot << 15) | kloc;
    }
}

void make_desc(hipStream_t st, const DevMat &M, const Schedule &sch, int32_t **desc, bool fill_import_table)
{
    ILUPP_HIP(pool_malloc(desc, sizeof(int32_t) * (size_t)(M.nnz > 0 ? M.nnz : 1)));
    // (fill_import_table = false: the table stands as somebody made it -- which producer gets which cell depends on the order of the
    // insertions, so descriptors are only comparable on ONE table)
    if (sch.gtab && fill_import_table)
        hipLaunchKernelGGL(k_ghost_table, dim3((unsigned)(sch.nslots / kThreads)), dim3(kThreads), 0, st, M.ptr, M.idx,
                           sch.B, sch.nb, sch.start, sch.blk2slot, sch.fwd ? 1 : 0, sch.sfirst, sch.scount, sch.gtab);
    hipLaunchKernelGGL(k_make_desc, dim3((unsigned)((M.n + 255) / 256)), dim3(256), 0, st, M.n, M.ptr, M.idx,
                       sch.B, sch.nb, sch.fwd ? 1 : 0, sch.start, sch.blk2slot, *desc, sch.exported, sch.gtab);
}

// The descriptors of L^T of an LL^T object on a box grid (icholt_grid.hip: columns of four entries -- diagonal, +1, +nx, +nx ny --, one block
// per x-line, backward schedule): what k_make_desc finds by ptr -> idx -> block_of -> start for every entry follows from (x, y, z).
__global__ void __launch_bounds__(256)
k_make_desc_llt_grid(const int32_t n, const GridDims g, const int32_t *__restrict__ blk2slot, int32_t *__restrict__ desc,
                     int32_t *__restrict__ exported, const int32_t *__restrict__ gtab, int32_t *__restrict__ Lptr, int32_t *__restrict__ Lidx,
                     const long long nnzL)
{
    const unsigned unx = (unsigned)g.nx, uny = (unsigned)g.ny;
    for (long long rr = (long long)blockIdx.x * 256 + threadIdx.x; rr < n; rr += (long long)gridDim.x * 256) {
        const unsigned r = (unsigned)rr;
        const unsigned l = r / unx, x = r - l * unx;
        const unsigned z = l / uny, y = l - z * uny;
        long long q = ig_col_start((int)x, (int)y, (int)z, g);
        if (Lptr) {
            // (the factor's own index arrays in the same pass: icholt_grid.hip's k_icholt_grid_pattern writes nothing else)
            long long w = q;
            Lptr[r] = (int32_t)q;
            Lidx[w++] = (int32_t)r;
            if ((int)x < g.nx - 1) Lidx[w++] = (int32_t)r + 1;
            if ((int)y < g.ny - 1) Lidx[w++] = (int32_t)r + g.nx;
            if ((int)z < g.nz - 1) Lidx[w++] = (int32_t)r + g.nx * g.ny;
            if (rr == n - 1) Lptr[n] = (int32_t)nnzL;
        }
        const int my = blk2slot[l];
        const int mywg = my >> 8;
        desc[q++] = -1;
        // (backward schedule: a row's place in its line counts from the line's end)
        if ((int)x < g.nx - 1) desc[q++] = (my << 15) | (g.nx - 2 - (int)x);
        const int kloc = g.nx - 1 - (int)x;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const bool has = d == 0 ? (int)y < g.ny - 1 : (int)z < g.nz - 1;
            if (!has) continue;
            int oslot = blk2slot[l + (d == 0 ? 1u : uny)];
            if ((oslot >> 8) != mywg) {
                exported[oslot] = 1;
                if (gtab && kloc < 0x7fff) {
                    const int32_t *tab = gtab + (size_t)mywg * kGhosts;
                    unsigned h = ((unsigned)oslot * 0x9E3779B1u) >> 26;
                    for (int probe = 0; probe < kGhosts; ++probe, h = (h + 1) & (kGhosts - 1)) {
                        const int cur = tab[h];
                        if (cur == oslot) { oslot = kGhostBase + (int)h; break; }
                        if (cur == -1) break;
                    }
                }
            }
            desc[q++] = (oslot << 15) | kloc;
        }
    }
}

// ... and the import table of such an object: a lane's foreign producers are the lines y + 1 and z + 1 of its own line, where those lie
// in another workgroup (k_ghost_table finds the same set on three sampled rows of every lane; which cell a producer gets depends on the
// order of the insertions there as here)
__global__ void __launch_bounds__(kThreads)
k_ghost_table_llt_grid(const GridDims g, const int32_t *__restrict__ slot2blk, const int32_t *__restrict__ blk2slot, int32_t *__restrict__ gtab)
{
    __shared__ int tab[kGhosts];
    const int wg = blockIdx.x, t = threadIdx.x;
    if (t < kGhosts) tab[t] = -1;
    __syncthreads();
    const int l = slot2blk[wg * kThreads + t];
    if (l >= 0 && g.nx > 1) {
        const int z = l / g.ny, y = l - z * g.ny;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const bool has = d == 0 ? y < g.ny - 1 : z < g.nz - 1;
            if (!has) continue;
            const int oslot = blk2slot[l + (d == 0 ? 1 : g.ny)];
            if ((oslot >> 8) == wg) continue;
            unsigned h = ((unsigned)oslot * 0x9E3779B1u) >> 26;
            for (int probe = 0; probe < kGhosts; ++probe, h = (h + 1) & (kGhosts - 1)) {
                const int cur = atomicCAS(&tab[h], -1, oslot);
                if (cur == -1 || cur == oslot) break;
            }
        }
    }
    __syncthreads();
    if (t < kGhosts) gtab[(size_t)wg * kGhosts + t] = tab[t];
}

void make_desc_llt_grid(hipStream_t st, const DevMat &M, const Schedule &sch, const GridDims &g, int32_t **desc, bool with_pattern)
{
    ILUPP_HIP(pool_malloc(desc, sizeof(int32_t) * (size_t)(M.nnz > 0 ? M.nnz : 1)));
    if (sch.gtab)
        hipLaunchKernelGGL(k_ghost_table_llt_grid, dim3((unsigned)(sch.nslots / kThreads)), dim3(kThreads), 0, st, g, sch.slot2blk, sch.blk2slot, sch.gtab);
    hipLaunchKernelGGL(k_make_desc_llt_grid, dim3(2048), dim3(256), 0, st, M.n, g, sch.blk2slot, *desc, sch.exported, sch.gtab,
                       with_pattern ? M.ptr : nullptr, with_pattern ? M.idx : nullptr, (long long)M.nnz);
    ILUPP_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// ILU(0) update program
// ---------------------------------------------------------------------------------------------
// record of row r (32-bit words):
//   [ len | cl << 16 ]
//   per strictly-lower entry e (ascending column k):
//       [ piv_pos ]                 index of U row k's diagonal in U.val
//       [ kloc ]                    row k's index in its owner's processing order
//       [ oslot << 8 | nm ]         owner slot, number of matches
//       nm x [ off | pp << 16 ]     U row k's entry piv_pos+off updates working-row position pp
// stats: [0] max words per row, [1] max U-row length, [2] limit violations
template <bool WRITE>
__global__ void k_ilu0_program(int32_t n, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx,
                               const int32_t *__restrict__ Uptr, int32_t B, int32_t nb,
                               const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot,
                               int32_t *__restrict__ nwords, const int32_t *__restrict__ prow,
                               int32_t *__restrict__ prog, int32_t *__restrict__ stats)
{
    const int rr = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = rr < n;
    const int r = in_range ? rr : n - 1;          // out-of-range threads shadow the last row and write nothing
    const int a0 = Aptr[r], a1 = Aptr[r + 1];
    const int len = a1 - a0;
    int cl = 0;
    while (cl < len && Aidx[a0 + cl] < r) ++cl;
    int w = 0, bad = 0;
    if (WRITE && !in_range) return;
    int32_t *out = WRITE ? prog + prow[r] : nullptr;
    if (WRITE) out[0] = len | (cl << 16);
    w = 1;
    if (len >= 65536) bad = 1;
    for (int e = 0; e < cl; ++e) {
        const int k = Aidx[a0 + e];
        const int k0 = Aptr[k], k1 = Aptr[k + 1];
        // first entry of row k with column > k; the diagonal sits just before it
        int ku = k0;
        while (ku < k1 && Aidx[ku] <= k) ++ku;
        const int kd = ku - 1;      // position of (k,k) in A's row k (presence checked by the count pass)
        int nm = 0, pp = e + 1;
        for (int j = ku; j < k1; ++j) {
            const int m = Aidx[j];
            while (pp < len && Aidx[a0 + pp] < m) ++pp;
            if (pp >= len) break;
            if (Aidx[a0 + pp] == m) {
                if (WRITE) out[w + 3 + nm] = (j - kd) | (pp << 16);
                ++nm; ++pp;
            }
        }
        if (nm > 255 || (k1 - kd) >= 65536) bad = 1;
        if (WRITE) {
            const int b = block_of(k, B, nb, start);
            const int oslot = blk2slot[b];
            out[w] = Uptr[k];
            out[w + 1] = k - start[b];
            out[w + 2] = (oslot << 8) | (nm & 255);
        }
        w += 3 + nm;
    }
    if (!WRITE) {
        if (in_range) nwords[r] = w;
        // U-row length of this row (diagonal + strictly upper); statistics leave through LDS, one set of
        // atomics per workgroup
        const int ulen = len - cl;
        __shared__ int red[3];
        if (threadIdx.x < 3) red[threadIdx.x] = 0;
        __syncthreads();
        int mw = w, mu = ulen;
        for (int off = 32; off > 0; off >>= 1) { mw = max(mw, __shfl_xor(mw, off)); mu = max(mu, __shfl_xor(mu, off)); }
        const unsigned long long anybad = __ballot(bad);
        if ((threadIdx.x & 63) == 0) { atomicMax(&red[0], mw); atomicMax(&red[1], mu); if (anybad) atomicAdd(&red[2], 1); }
        __syncthreads();
        if (threadIdx.x == 0) { atomicMax(&stats[0], red[0]); atomicMax(&stats[1], red[1]); if (red[2]) atomicAdd(&stats[2], 1); }
    }
}

// ---------------------------------------------------------------------------------------------
// fixed-size update program "F3" for short-row matrices (<= 3 eliminations, <= 5 matches, <= 8 entries per row,
// U rows of <= 4 entries): 8 words (32 B) per row, built in ONE pass with no scan, streamed by the loader waves
// of k_ilu0_numeric_lc.  The 5- and 7-point stencils fit; anything else takes the variable-length program.
//
// The working row is DIAGONAL-ALIGNED: its diagonal sits at position 3, the cl eliminations right-aligned in
// positions 3-cl..2, the strictly-upper entries in 4..6.  Every position the kernel touches is then a fixed
// register (dep slot s <-> position s), whatever the row's shape.
//   w0      len (4b) | cl (2b) << 4 | nmt (3b) << 6 | m0 << 9 | m1 << 16 | m2 << 23
//   w1      m3 | m4 << 7                      match m = s (2b) | off (2b) << 2 | pp (3b) << 4   (pp: aligned position)
//   w2+2s   dep slot s: owner_slot << 15 | kloc    (same encoding as a solve descriptor); slots < 3-cl unused
//   w3+2s   dep slot s: piv_pos (index of U row k's diagonal in U.val)
// matches are grouped by s ascending (= ascending k), ascending column inside: U entry piv_pos+off of dep s
// updates aligned position pp
__global__ void k_ilu0_program_f3(int32_t n, const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx,
                                  const int32_t *__restrict__ Uptr, int32_t B, int32_t nb,
                                  const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot,
                                  int32_t *__restrict__ prog, int32_t *__restrict__ ineligible, int32_t *__restrict__ exported)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int mywg = blk2slot[block_of(r, B, nb, start)] >> 8;
    const int a0 = Aptr[r], a1 = Aptr[r + 1];
    const int len = a1 - a0;
    int cl = 0;
    while (cl < len && Aidx[a0 + cl] < r) ++cl;
    bool bad = (len > 7) || (cl > 3) || (len - cl > 4);
    int d0 = 0, p0 = 0, d1 = 0, p1 = 0, d2 = 0, p2 = 0;
    unsigned long long mbits = 0;      // 5 x 7 bits
    int nmt = 0;
    if (!bad) {
        for (int e = 0; e < cl; ++e) {
            const int sl = e + 3 - cl;                    // right-aligned dep slot
            const int k = Aidx[a0 + e];
            const int k0 = Aptr[k], k1 = Aptr[k + 1];
            int ku = k0;
            while (ku < k1 && Aidx[ku] <= k) ++ku;
            const int kd = ku - 1;
            if (k1 - kd > 4) bad = true;                 // U row of k longer than the LDS ring entry
            int pp = e + 1;
            for (int j = ku; j < k1; ++j) {
                const int m = Aidx[j];
                while (pp < len && Aidx[a0 + pp] < m) ++pp;
                if (pp >= len) break;
                if (Aidx[a0 + pp] == m) {
                    if (nmt < 5) mbits |= (unsigned long long)(sl | (((j - kd) & 3) << 2) | (((pp - cl + 3) & 7) << 4)) << (7 * nmt);
                    ++nmt; ++pp;
                }
            }
            const int b = block_of(k, B, nb, start);
            const int kd_word = (blk2slot[b] << 15) | (k - start[b]);
            if ((blk2slot[b] >> 8) != mywg) exported[blk2slot[b]] = 1;
            const int pv = Uptr[k];
            if (sl == 0) { d0 = kd_word; p0 = pv; }
            else if (sl == 1) { d1 = kd_word; p1 = pv; }
            else { d2 = kd_word; p2 = pv; }
        }
        if (nmt > 5) bad = true;
    }
    // "simple" rows (bit 30): every elimination has exactly one match and it lands on the diagonal -- the shape of
    // every interior row of a 5-/7-point stencil; the kernel then runs a straight-line path
    bool simple = !bad && nmt == cl;
    for (int m = 0; m < 3 && simple; ++m)
        if (m < nmt) {
            const int mw = (int)((mbits >> (7 * m)) & 127);
            simple = ((mw & 3) == m + 3 - cl) && (((mw >> 4) & 7) == 3);
        }
    const int w0 = (len & 15) | ((cl & 3) << 4) | (((nmt > 5 ? 5 : nmt) & 7) << 6) | (int)((mbits & 0x1FFFFFull) << 9) | (simple ? (1 << 30) : 0);
    const int w1 = (int)((mbits >> 21) & 0x3FFFull);
    int4 *out = reinterpret_cast<int4 *>(prog + (size_t)r * 8);
    out[0] = make_int4(w0, w1, d0, p0);
    out[1] = make_int4(d1, p1, d2, p2);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicExch(ineligible, 1);
}

// F3 program; returns false (and frees it) when some row does not fit the fixed record
bool build_ilu0_program_f3(hipStream_t st, const DevMat &A, const DevMat &U, const Schedule &sch, int32_t **prog_out)
{
    const int32_t n = A.n;
    if (sch.B > 32768 || sch.nslots > kGhostBase) return false;
    int32_t *prog = nullptr, *flag = nullptr;
    ILUPP_HIP(pool_malloc(&prog, sizeof(int32_t) * 8 * (size_t)n + 64));
    ILUPP_HIP(pool_malloc(&flag, 16));
    ILUPP_HIP(hipMemsetAsync(flag, 0, 16, st));
    hipLaunchKernelGGL(k_ilu0_program_f3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, A.ptr, A.idx, U.ptr,
                       sch.B, sch.nb, sch.start, sch.blk2slot, prog, flag, sch.exported);
    int32_t h = 0;
    ILUPP_HIP(d2h_async(st, &h, flag, sizeof(int32_t)));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(flag));
    if (h != 0) { ILUPP_HIP(pool_free(prog)); return false; }
    *prog_out = prog;
    return true;
}

// builds the program for schedule `sch` (forward); returns false if the compact encoding cannot
// represent this matrix (then the generic kernel of ilu0.hip is used instead)
bool build_ilu0_program(hipStream_t st, const DevMat &A, const DevMat &U, const Schedule &sch, Ilu0Program *P)
{
    const int32_t n = A.n;
    int32_t *nwords = nullptr, *stats = nullptr;
    ILUPP_HIP(pool_malloc(&nwords, sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(pool_malloc(&stats, 16));
    ILUPP_HIP(hipMemsetAsync(stats, 0, 16, st));
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL((k_ilu0_program<false>), dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, U.ptr, sch.B, sch.nb,
                       sch.start, sch.blk2slot, nwords, (const int32_t *)nullptr, (int32_t *)nullptr, stats);
    ILUPP_HIP(pool_malloc(&P->prow, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(hipMemsetAsync(P->prow, 0, sizeof(int32_t), st));
    // 64-bit total first: the compact program indexes words with int32
    size_t tmp_bytes = 0;
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, nwords, P->prow + 1, n, st));
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, nwords, P->prow + 1, n, st));
    int32_t h[4], total = 0;
    ILUPP_HIP(d2h_async(st, h, stats, 16));
    ILUPP_HIP(d2h_async(st, &total, P->prow + n, sizeof(int32_t)));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(tmp));
    P->max_words = h[0];
    P->max_ulen = h[1];
    const bool ok = (h[2] == 0) && total > 0 && sch.B <= 32768 && sch.nslots <= kGhostBase;
    if (!ok) {
        ILUPP_HIP(pool_free(nwords)); ILUPP_HIP(pool_free(stats));
        ILUPP_HIP(pool_free(P->prow)); P->prow = nullptr;
        return false;
    }
    P->nwords = total;
    ILUPP_HIP(pool_malloc(&P->prog, sizeof(int32_t) * (size_t)total));
    hipLaunchKernelGGL((k_ilu0_program<true>), dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, U.ptr, sch.B, sch.nb,
                       sch.start, sch.blk2slot, nwords, P->prow, P->prog, stats);
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(nwords)); ILUPP_HIP(pool_free(stats));
    return true;
}

}  // namespace ilupp
