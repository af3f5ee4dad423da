// ilupp_amd/csrc/icholt_grid.hip -- ICholT(add_fill_in = 0, threshold = 0) of a box-grid stencil matrix (BASELINE config C4) as a
// SPECULATIVE static computation.
//
// The reference (IChol.hpp:78-155, restated in oracle/ilupp_oracle.c: orc_icholt) builds column j of L from a working column w:
//   w = A(j:, j);  D[j] += a_jj;  L_jj = sqrt(D[j]);  for every column k < j with L(j, k) != 0:  w(i) -= L(i, k) L(j, k)  (i > j);
//   w(i) /= L_jj and D[i] -= w(i)^2 for EVERY i > j of w (IChol.hpp:135-141: before anything is dropped);
//   keep the col_len = nnz(A(j:, j)) entries of largest magnitude, the diagonal competing (dropping.hpp:8-34), sorted by row.
// On a lexicographic 7-point grid (row j = (x, y, z), neighbours j+1, j+nx, j+nx ny) with L on A's pattern, the working column of j is
// A's four entries -- which receive NO update: none of the earlier columns j-1, j-nx, j-nx ny has an entry in the rows j+1, j+nx,
// j+nx ny -- plus three fill entries, each with exactly one contribution:
//   f1 = w(j+nx-1)    = -(L(j+nx-1, j-1)  L(j, j-1))  / L_jj      (column j-1:  e2 e1 of the lane's previous column)
//   f2 = w(j+nx ny-1) = -(L(j+nx ny-1, j-1) L(j, j-1)) / L_jj     (column j-1:  e3 e1)
//   f3 = w(j+nx ny-nx)= -(L(j+nx ny-nx, j-nx) L(j, j-nx)) / L_jj  (column j-nx: e3 e2 of the line y-1)
// with e1 = a(j, j+1) / L_jj, e2 = a(j, j+nx) / L_jj, e3 = a(j, j+nx ny) / L_jj.  So IF every column keeps exactly A's entries, then
//   D[i] = 0 - e3^2(i-nx ny) - f2^2(i-nx ny+1) - f3^2(i-nx ny+nx) - e2^2(i-nx) - f1^2(i-nx+1) - e1^2(i-1) + a_ii   (ascending columns)
// is a recurrence over six earlier columns: the lines (y, z-1) at x and x+1, (y+1, z-1) at x, (y-1, z) at x and x+1, and the own line at
// x-1 -- a wavefront x + 2y + 3z.  The kernel runs that recurrence with one lane per x-line (a workgroup = a 16 x 16 patch of lines,
// lane (y, z) works on column x = step - 2y - 3z), hands the six published squares / products of a column from lane to lane through
// LDS and from patch to patch through a step-major exchange in memory (a courier wave polls and exports, as in st_wave.hip), and
// VERIFIES for every column what it assumed: the diagonal and every A entry are strictly larger in magnitude than every fill entry
// (then the cut of dropping.hpp keeps exactly them, whatever it does with ties), nothing is NaN.  One violated column, or a matrix
// that is no box grid (grid.hip: k_grid_check), and the result is thrown away: the dataflow kernel (icholt_df.hip) builds the object.
// Values: the same operations in the same order as the reference, so the factor is bit-identical (tests/test_gpu_icholt_grid.py).
#include "st_common.h"

namespace ilupp {
namespace {

constexpr int kIgLanes = 256, kIgPairs = 64;
constexpr int kIgZero = kIgLanes + kIgPairs;              // a cell nobody writes: the source of a lane without that neighbour
constexpr int kIgRow = kIgLanes + kIgPairs + 8;           // doubles per (word, slot)
constexpr int kIgSlots = 8, kIgWords = 6;
constexpr unsigned kIgSlotB = kIgRow * 8u;
constexpr unsigned kIgWordB = kIgSlots * kIgSlotB;
constexpr unsigned kIgLds = kIgWords * kIgWordB;          // 125 952 bytes
constexpr int kIgExp = 32;                                // exported lanes of a patch: y' = 15 (16), z' = 15 (15 more)
constexpr int kIgThreads = 320;                           // four consumer waves + the courier
#ifndef IG_NP
#define IG_NP 0
#endif
constexpr int kIgNP = IG_NP;                              // the imports of a step are polled for kIgNP + 1 steps ahead (256^3: 3.7-3.85 ms with 0, 3.85 with 1, 4.0 with 3, 4.2 with 5)
constexpr int kIgRA = 4;                                  // steps ahead a lane reads A
constexpr int kIgMaxSkew = 2 * 15 + 15;
// words of a published record (column m of a lane): what the lanes (y+1, z), (y, z+1), (y-1, z+1) subtract from their diagonals
enum { IG_E2P = 0, IG_QP = 1, IG_F1 = 2, IG_F3 = 3, IG_E3P = 4, IG_F2 = 5 };   // e2^2, e3 e2 of column m-1; f1^2, f3^2 of m; e3^2 of m-1; f2^2 of m

struct IgArgs {
    GridDims g;
    int nty, ntz, S;
    const double *aval; unsigned abytes;
    double *lval; unsigned lbytes;
    unsigned long long *xch;          // [patch][kIgExp][nx + 1][8]
    const unsigned long long *idle;   // a word that is not the sentinel
    int32_t *ctrl;                    // [0] ticket, [1] time-out / error, [2] a column that does not keep A's pattern
};

// entries of L before column r = (x, y, z): four per column minus the neighbours beyond the box
__device__ __forceinline__ long long ig_col_start(const int x, const int y, const int z, const GridDims &g)
{
    const long long nx = g.nx, ny = g.ny;
    const long long r = x + nx * (y + ny * (long long)z);
    long long miss = r / nx;                                              // ends of the lines before
    miss += (long long)z * nx + (y == g.ny - 1 ? x : 0);                  // last lines of the planes before, of this plane
    miss += z == g.nz - 1 ? r - (long long)z * nx * ny : 0;               // the last plane
    return 4 * r - miss;
}

__global__ void __launch_bounds__(256)
k_icholt_grid_pattern(const int32_t n, const GridDims g, int32_t *__restrict__ ptr, int32_t *__restrict__ idx, const long long nnzL)
{
    const unsigned unx = (unsigned)g.nx, uny = (unsigned)g.ny;
    for (long long rr = (long long)blockIdx.x * 256 + threadIdx.x; rr < n; rr += (long long)gridDim.x * 256) {
        const unsigned r = (unsigned)rr;
        const unsigned l = r / unx, x = r - l * unx;
        const unsigned z = l / uny, y = l - z * uny;
        long long q = ig_col_start((int)x, (int)y, (int)z, g);
        ptr[r] = (int32_t)q;
        idx[q++] = (int32_t)r;
        if ((int)x < g.nx - 1) idx[q++] = (int32_t)r + 1;
        if ((int)y < g.ny - 1) idx[q++] = (int32_t)r + g.nx;
        if ((int)z < g.nz - 1) idx[q++] = (int32_t)r + g.nx * g.ny;
        if (rr == n - 1) ptr[n] = (int32_t)nnzL;
    }
}

// ---------------------------------------------------------------------------------------------
// a consumer lane: the x-line (y, z)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ig_consumer(const IgArgs &A, unsigned char *lds, const int ty, const int tz)
{
    typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
    // a patch is SHEARED: its lane (y', z') is the line y = 16 ty + y' - z', z = 16 tz + z' -- the line (y+1, z-1) is the lane below, and
    // every line a patch needs from another patch belongs to one with a smaller ticket (ty - 1 or tz - 1): with upright patches the
    // neighbours in y would wait for each other column by column
    const int yl = t & 15, zl = t >> 4;
    const int y = ty * 16 + yl - zl, z = tz * 16 + zl;
    const int nx = A.g.nx;
    const bool active = y >= 0 && y < A.g.ny && z < A.g.nz;
    const int sk = 2 * yl + zl;
    const bool has2 = active && y < A.g.ny - 1, has3 = active && z < A.g.nz - 1;
    // where the three neighbours' records are read: a lane of this patch (d steps back), a pair of the courier (this step's slot), nobody
    unsigned srcA = kIgZero, srcB = kIgZero, srcC = kIgZero;
    int dA = 0, dB = 0, dC = 0;
    if (active && y > 0) { if (yl > 0) { srcA = (unsigned)(t - 1); dA = 1; } else srcA = (unsigned)(kIgLanes + zl); }
    if (active && z > 0) {
        if (zl > 0) { if (yl > 0) { srcB = (unsigned)(t - 17); dB = 2; } else srcB = (unsigned)(kIgLanes + 16 + (zl - 1)); }
        else srcB = (unsigned)(kIgLanes + 32 + yl);
    }
    if (active && z > 0 && y + 1 < A.g.ny) { if (zl > 0) { srcC = (unsigned)(t - 16); dC = 1; } else srcC = (unsigned)(kIgLanes + 48 + yl); }
    const unsigned aA = srcA * 8u, aB = srcB * 8u, aC = srcC * 8u, aMe = (unsigned)t * 8u;
    // A: the upper part of row (x, y, z) starts at line start + entries left of the diagonal of row 0 + x * (entries of an inner row)
    const int edge = active ? ((y == 0) + (y == A.g.ny - 1) + (z == 0) + (z == A.g.nz - 1)) : 0;
    const unsigned lenb = (unsigned)(7 - edge) * 8u;
    const long long ls = active ? grid_row_start(0, y, z, A.g) + (y > 0 ? 1 : 0) + (z > 0 ? 1 : 0) : 0;
    const unsigned ua0 = (unsigned)ls * 8u;
    const unsigned cub = (unsigned)(2 + (has2 ? 1 : 0) + (has3 ? 1 : 0)) * 8u;
    const unsigned ul0 = active ? (unsigned)ig_col_start(0, y, z, A.g) * 8u : 0u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.aval), 0, (int)A.abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(A.lval, 0, (int)A.lbytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

#ifdef IG_X_NOLOAD
#define IG_NOLOAD_(o) (o) = OOB
#else
#define IG_NOLOAD_(o) (void)0
#endif
    double ring[kIgRA][4];
#define IG_LOAD(slot_, k_)                                                                                   \
    do {                                                                                                     \
        unsigned o_ = (active && (unsigned)(k_) < (unsigned)nx) ? ua0 + (unsigned)(k_) * lenb : OOB;         \
        IG_NOLOAD_(o_);                                                                                      \
        asm volatile("" : "+v"(o_));                                                                         \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                     \
            ring[slot_][j_] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ra, o_ + 8u * (unsigned)j_, 0, 0)); \
    } while (0)
#pragma unroll
    for (int u = 0; u < kIgRA; ++u) IG_LOAD(u, u - sk);

    double e1p = 0.0, e2p = 0.0, e3p = 0.0, e1sqp = 0.0, e2sqp = 0.0, e3sqp = 0.0, qp = 0.0;
    bool bad = false;
    for (int sb = 0; sb < A.S; sb += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = sb + u;
            const int k = s - sk;
            const bool valid = active && (unsigned)k < (unsigned)nx;
            const bool has1 = valid && k < nx - 1;
            const double w0 = ring[u % kIgRA][0], w1 = ring[u % kIgRA][1], w2 = ring[u % kIgRA][2], w3 = ring[u % kIgRA][3];
            const double a0 = w0;
            const double a1 = has1 ? w1 : 0.0;
            const double a2 = has2 ? (has1 ? w2 : w1) : 0.0;
            const double a3m = has2 ? (has1 ? w3 : w2) : (has1 ? w2 : w1);
            const double a3 = has3 ? a3m : 0.0;
            ST_BARRIER();
            const unsigned oA = (unsigned)((u - dA) & 7) * kIgSlotB + aA;
            const unsigned oB = (unsigned)((u - dB) & 7) * kIgSlotB + aB;
            const unsigned oC = (unsigned)((u - dC) & 7) * kIgSlotB + aC;
            const double inE2 = st_lds(lds, IG_E2P * kIgWordB + oA), inQ = st_lds(lds, IG_QP * kIgWordB + oA), inF1 = st_lds(lds, IG_F1 * kIgWordB + oA);
            const double inE3 = st_lds(lds, IG_E3P * kIgWordB + oB), inF2 = st_lds(lds, IG_F2 * kIgWordB + oB);
            const double inF3 = st_lds(lds, IG_F3 * kIgWordB + oC);
            // D[j] over the columns that touched it, ascending (IChol.hpp:139), then the diagonal (IChol.hpp:112-113)
            double D = 0.0;
            D -= inE3; D -= inF2; D -= inF3; D -= inE2; D -= inF1; D -= e1sqp;
            D += a0;
            const double p = sqrt(D);
            double e1 = a1 / p, e2 = a2 / p, e3 = a3 / p;                   // IChol.hpp:137
            double f1 = (0.0 - e2p * e1p) / p;                              // IChol.hpp:128 (one contribution each), :137
            double f2 = (0.0 - e3p * e1p) / p;
            double f3 = (0.0 - inQ) / p;
            e1 = has1 ? e1 : 0.0; e2 = (valid && has2) ? e2 : 0.0; e3 = (valid && has3) ? e3 : 0.0;
            f1 = valid ? f1 : 0.0; f2 = valid ? f2 : 0.0; f3 = valid ? f3 : 0.0;
            // the premise: the cut (dropping.hpp:17-29) keeps the diagonal and A's entries -- all of them strictly above every fill entry
            const double vmax = fmax(fabs(f1), fmax(fabs(f2), fabs(f3)));
            const bool keep = p > vmax && (!has1 || fabs(e1) > vmax) && (!has2 || fabs(e2) > vmax) && (!has3 || fabs(e3) > vmax);
            bad = bad || (valid && !keep);
            const double e1sq = e1 * e1, e2sq = e2 * e2, e3sq = e3 * e3, q = e3 * e2;
            const unsigned oM = (unsigned)u * kIgSlotB + aMe;
            *reinterpret_cast<double *>(lds + IG_E2P * kIgWordB + oM) = e2sqp;
            *reinterpret_cast<double *>(lds + IG_QP * kIgWordB + oM) = qp;
            *reinterpret_cast<double *>(lds + IG_F1 * kIgWordB + oM) = f1 * f1;
            *reinterpret_cast<double *>(lds + IG_F3 * kIgWordB + oM) = f3 * f3;
            *reinterpret_cast<double *>(lds + IG_E3P * kIgWordB + oM) = e3sqp;
            *reinterpret_cast<double *>(lds + IG_F2 * kIgWordB + oM) = f2 * f2;
            e1p = e1; e2p = e2; e3p = e3; e1sqp = e1sq; e2sqp = e2sq; e3sqp = e3sq; qp = q;
            // column j of L: the diagonal and A's entries, by ascending row
            {
                const unsigned pos = ul0 + (unsigned)k * cub;
                unsigned o0 = valid ? pos : OOB;
                unsigned o1 = has1 ? pos + 8u : OOB;
                unsigned o2 = (valid && has2) ? pos + 8u + (has1 ? 8u : 0u) : OOB;
                unsigned o3 = (valid && has3) ? pos + 8u + (has1 ? 8u : 0u) + (has2 ? 8u : 0u) : OOB;
                asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
#ifdef IG_X_NOSTORE
                o0 = o1 = o2 = o3 = OOB;
#endif
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, p), rl, o0, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, e1), rl, o1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, e2), rl, o2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, e3), rl, o3, 0, 0);
            }
            IG_LOAD(u % kIgRA, k + kIgRA);
        }
    }
#undef IG_LOAD
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (t & 63) == 0) atomicOr(&A.ctrl[2], 1);
}

// ---------------------------------------------------------------------------------------------
// the courier: lane p brings the record words of one (lane of this patch, neighbour in another patch) pair into the hand-off array
// one step before they are read, and exports the records of this patch's border lanes one step after they were written
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ig_courier(const IgArgs &A, unsigned char *lds, const int ty, const int tz, const int wg)
{
    const int p = threadIdx.x & 63;
    const int nx = A.g.nx;
    // ---- imports
    int cy = 0, cz = 0, sty = ty, stz = tz, se = 0, mshift = 1, w0 = 0, w1 = 0, w2 = 0;
    bool exists = false, wantC = false;
    if (p < 16) {                      // line (y-1, z) for the lanes (0, z'): lane (15, z') of the patch before in y
        cy = 0; cz = p; sty = ty - 1; se = p; w0 = IG_E2P; w1 = IG_QP; w2 = IG_F1;
        exists = ty > 0;
    } else if (p < 31) {               // line (y, z-1) for the lanes (0, z' >= 1): lane (15, z'-1) of the patch before in y
        cy = 0; cz = p - 15; sty = ty - 1; se = cz - 1; w0 = IG_E3P; w1 = w2 = IG_F2;
        exists = ty > 0;
    } else if (p >= 32 && p < 48) {    // line (y, z-1) for the lanes (y', 0): lane (y'-1, 15) of the next patch in y of the row below, (15, 15) of this column's
        cy = p - 32; cz = 0; stz = tz - 1; w0 = IG_E3P; w1 = w2 = IG_F2;
        if (cy == 0) se = 15; else { sty = ty + 1; se = 16 + cy - 1; }
        exists = tz > 0;
    } else if (p >= 48) {              // line (y+1, z-1) for the lanes (y', 0): lane (y', 15) of the next patch in y of the row below
        cy = p - 48; cz = 0; stz = tz - 1; sty = ty + 1; se = cy == 15 ? 15 : 16 + cy; mshift = 0; w0 = w1 = w2 = IG_F3;
        exists = tz > 0; wantC = true;
    }
    {
        const int y = ty * 16 + cy - cz, z = tz * 16 + cz;
#ifdef IG_X_NOCOURIER
        exists = false;
#endif
        exists = exists && y >= 0 && y < A.g.ny && z < A.g.nz && sty >= 0 && sty < A.nty && (p >= 16 || y > 0) && (!wantC || y + 1 < A.g.ny);
    }
    const int skc = 2 * cy + cz;
    const unsigned long long *rec0 = A.xch + ((size_t)(stz * A.nty + sty) * kIgExp + (size_t)se) * (size_t)(nx + 1) * 8;
    const unsigned dst = (unsigned)(kIgLanes + p) * 8u;
    // ---- exports
    int et = -1;
    if (p < 16) et = 16 * p + 15; else if (p < 31) et = 240 + (p - 16);
    int ske = 0;
    if (et >= 0) {
        const int yl = et & 15, zl = et >> 4;
        const int y = ty * 16 + yl - zl;
        ske = 2 * yl + zl;
        if (y < 0 || y >= A.g.ny || tz * 16 + zl >= A.g.nz) et = -1;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(A.xch + (size_t)wg * kIgExp * (size_t)(nx + 1) * 8, 0,
                                                                        (int)((unsigned)kIgExp * (unsigned)(nx + 1) * 64u), 0x00020000);
    const unsigned eoff = (unsigned)(p < kIgExp ? p : 0) * (unsigned)(nx + 1) * 64u;

    unsigned long long gq[kIgNP + 1][3];
#define IGC_ADDR(s_, w_) ((exists && (unsigned)((s_) - skc) < (unsigned)nx) ? rec0 + ((size_t)((s_) - skc + mshift) * 8 + (size_t)(w_)) : A.idle)
#define IGC_POLL(s_, slot_)                                                                                  \
    do { gq[slot_][0] = ld_agent_u64(IGC_ADDR(s_, w0)); gq[slot_][1] = ld_agent_u64(IGC_ADDR(s_, w1)); gq[slot_][2] = ld_agent_u64(IGC_ADDR(s_, w2)); } while (0)
    bool dead = false;
    // deliver what the lanes read at step s_ (slot s_ & 7), then poll for step s_ + kIgNP + 1
#define IGC_DELIVER(s_, i_)                                                                                  \
    do {                                                                                                     \
        const bool need = exists && (unsigned)((s_) - skc) < (unsigned)nx;                                   \
        unsigned long long v0 = gq[(i_) % (kIgNP + 1)][0], v1 = gq[(i_) % (kIgNP + 1)][1], v2 = gq[(i_) % (kIgNP + 1)][2]; \
        if (!dead) {                                                                                         \
            unsigned spins = 0;                                                                              \
            while (__builtin_amdgcn_ballot_w64(need && (v0 == kSentinel || v1 == kSentinel || v2 == kSentinel)) != 0) { \
                if (need && v0 == kSentinel) v0 = ld_agent_u64(IGC_ADDR(s_, w0));                            \
                if (need && v1 == kSentinel) v1 = ld_agent_u64(IGC_ADDR(s_, w1));                            \
                if (need && v2 == kSentinel) v2 = ld_agent_u64(IGC_ADDR(s_, w2));                            \
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                          \
                __builtin_amdgcn_s_sleep(1);                                                                 \
                if ((++spins & 255u) == 0) {                                                                 \
                    if (spins > kStSpinLimit) atomicExch(&A.ctrl[1], 1);                                     \
                    const int e = ld_agent_i32(&A.ctrl[1]);                                                  \
                    __builtin_amdgcn_s_waitcnt(0x0F70);                                                      \
                    if (spins > kStSpinLimit || e != 0) { dead = true; break; }                              \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
        if (!need) { v0 = 0; v1 = 0; v2 = 0; }                                                               \
        const unsigned o_ = (unsigned)((s_) & 7) * kIgSlotB + dst;                                           \
        *reinterpret_cast<unsigned long long *>(lds + (unsigned)w0 * kIgWordB + o_) = v0;                    \
        *reinterpret_cast<unsigned long long *>(lds + (unsigned)w1 * kIgWordB + o_) = v1;                    \
        *reinterpret_cast<unsigned long long *>(lds + (unsigned)w2 * kIgWordB + o_) = v2;                    \
        IGC_POLL((s_) + kIgNP + 1, (i_) % (kIgNP + 1));                                                      \
    } while (0)
#pragma unroll
    for (int g = 0; g <= kIgNP; ++g) IGC_POLL(g, g);
    IGC_DELIVER(0, 0);
    constexpr int UN = 4 * (kIgNP + 1);        // (a multiple of the ring and of nothing else that matters)
    for (int ib = 0; ib < A.S; ib += UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = ib + u;
            if (i < A.S) {
                ST_BARRIER();
                // the records of step i - 1 of the exported lanes
                const int m = i - 1 - ske;
                if (et >= 0 && i >= 1 && (unsigned)m <= (unsigned)nx) {
                    const unsigned o = (unsigned)((i - 1) & 7) * kIgSlotB + (unsigned)et * 8u;
                    unsigned long long w[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        w[j] = *reinterpret_cast<const unsigned long long *>(lds + (unsigned)j * kIgWordB + o);
                        w[j] = w[j] == kSentinel ? kCanonNaN : w[j];
                    }
                    // (buffer stores, sc1 = write-through: an inline-asm store would be a memory operation hipcc's wait counts do not know of --
                    // the polls behind it were then taken for complete one operation early)
                    const unsigned qo = eoff + (unsigned)m * 64u;
                    __builtin_amdgcn_raw_buffer_store_b128(v4u{(unsigned)w[0], (unsigned)(w[0] >> 32), (unsigned)w[1], (unsigned)(w[1] >> 32)}, rx, qo, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(v4u{(unsigned)w[2], (unsigned)(w[2] >> 32), (unsigned)w[3], (unsigned)(w[3] >> 32)}, rx, qo + 16u, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(v4u{(unsigned)w[4], (unsigned)(w[4] >> 32), (unsigned)w[5], (unsigned)(w[5] >> 32)}, rx, qo + 32u, 0, 16);
                }
                IGC_DELIVER(i + 1, u + 1);
            }
        }
    }
#undef IGC_DELIVER
#undef IGC_POLL
#undef IGC_ADDR
    if (dead && p == 0) atomicExch(&A.ctrl[1], 1);
}

__global__ void __launch_bounds__(kIgThreads)
k_icholt_grid(IgArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int s_wg;
    if (threadIdx.x == 0) s_wg = atomicAdd(&A.ctrl[0], 1);
    for (unsigned i = threadIdx.x; i < kIgLds / 8; i += kIgThreads) reinterpret_cast<double *>(lds)[i] = 0.0;
    __syncthreads();
    const int wg = s_wg;
    const int ty = wg % A.nty, tz = wg / A.nty;
    if (threadIdx.x < kIgLanes) ig_consumer(A, lds, ty, tz);
    else ig_courier(A, lds, ty, tz, wg);
}

}  // namespace

IcholtGridJob::~IcholtGridJob()
{
    if (pattern_done) (void)hipEventDestroy(pattern_done);
}

// Queues everything on st and returns: L's index arrays (closed form; `pattern_done` is recorded behind them -- the caller's sweep
// analysis needs nothing else and can run beside the kernel), the proof of the grid (k_grid_check), the exchange buffer's sentinels,
// the kernel, the read-back of its verdict.  *L owns the arrays at once.  false: a grid outside the kernel's limits, nothing was queued.
bool icholt_grid_launch(hipStream_t st, const DevMat &A, const GridDims &g, int32_t *ctrl, DevMat *L, IcholtGridJob *job)
{
    static const bool off = getenv("ILUPP_NO_ICHOLT_GRID") != nullptr;
    if (off) return false;
    const int64_t n = A.n;
    const int64_t nnzL = (A.nnz + n) / 2;
    const int nty = (g.ny + 15 + 15) / 16, ntz = (g.nz + 15) / 16;        // (sheared patches: y + z' runs to ny - 1 + 15)
    // buffer resources cover the value arrays
    if (A.nnz * 8 >= (1LL << 32) || nnzL >= (1LL << 31) || (int64_t)nty * ntz > (1 << 20)) return false;
    static bool attr = false;
    if (!attr) {
        ILUPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_icholt_grid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kIgLds));
        attr = true;
    }
    const int64_t xwords = (int64_t)nty * ntz * kIgExp * (g.nx + 1) * 8;
    ILUPP_HIP(job->xch.alloc(sizeof(unsigned long long) * (size_t)(xwords + 8)));
    unsigned long long *xp = job->xch.as<unsigned long long>();
    ILUPP_HIP(job->ev.create());
    ILUPP_HIP(hipEventCreateWithFlags(&job->pattern_done, hipEventDisableTiming));
    job->g = g;
    L->release();
    L->n = A.n; L->nnz = nnzL; L->is_csr = false; L->owns = true;
    ILUPP_HIP(pool_malloc(&L->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(pool_malloc(&L->idx, sizeof(int32_t) * (size_t)nnzL));
    ILUPP_HIP(pool_malloc(&L->val, sizeof(double) * (size_t)nnzL));
    ILUPP_HIP(hipMemsetAsync(ctrl, 0, sizeof(int32_t) * 16, st));
    hipLaunchKernelGGL(k_icholt_grid_pattern, dim3(2048), dim3(256), 0, st, A.n, g, L->ptr, L->idx, (long long)nnzL);
    ILUPP_HIP(hipEventRecord(job->pattern_done, st));
    grid_check_launch(st, A, g, ctrl + 8);
    fill_u64(st, xp, xwords, kSentinel);
    ILUPP_HIP(hipMemsetAsync(xp + xwords, 0, 64, st));
    IgArgs a;
    a.g = g; a.nty = nty; a.ntz = ntz;
    a.S = ((g.nx + kIgMaxSkew + 3) + 7) & ~7;
    a.aval = A.val; a.abytes = (unsigned)(A.nnz * 8);
    a.lval = L->val; a.lbytes = (unsigned)(nnzL * 8);
    a.xch = xp; a.idle = xp + xwords; a.ctrl = ctrl;
    ILUPP_HIP(hipEventRecord(job->ev.a, st));
    hipLaunchKernelGGL(k_icholt_grid, dim3((unsigned)(nty * ntz)), dim3(kIgThreads), kIgLds, st, a);
    ILUPP_HIP(hipGetLastError());
    ILUPP_HIP(hipEventRecord(job->ev.b, st));
    ILUPP_HIP(d2h_async(st, job->h, ctrl, sizeof(job->h)));
    ILUPP_HIP(hipMemsetAsync(ctrl, 0, sizeof(int32_t) * 16, st));
    return true;
}

// waits for st.  true: L (icholt_grid_launch) is ICholT(0, 0.0) of A -- a box grid, proven; every column kept A's pattern, verified.
// false: a violated premise, no grid after all, or a time-out: the caller drops L and takes the general way.
bool icholt_grid_finish(hipStream_t st, IcholtGridJob *job, float *kernel_ms)
{
    ILUPP_HIP(stream_sync(st));
    float ms = 0.f;
    ILUPP_HIP(hipEventElapsedTime(&ms, job->ev.a, job->ev.b));
    if (kernel_ms) *kernel_ms = ms;
    const int32_t *h = job->h;
    if (getenv("ILUPP_IG_DEBUG"))
        fprintf(stderr, "icholt_grid: %d x %d x %d tickets %d timeout %d premise %d grid %d kernel %.3f ms\n", job->g.nx, job->g.ny, job->g.nz, h[0], h[1], h[2], h[8], ms);
    return h[1] == 0 && h[2] == 0 && h[8] == 0;
}

}  // namespace ilupp
