// ilupp_amd/csrc/icholt_grid.hip -- ICholT(add_fill_in = 0, threshold = 0) of a box-grid stencil matrix (BASELINE config C4) as a
// SPECULATIVE static computation.
//
// The reference (IChol.hpp:78-155) builds column j of L from a working column w:
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
#include <mutex>

#include "st_common.h"

namespace ilupp {
namespace {

constexpr int kIgLanes = 256, kIgPairs = 64;
constexpr int kIgZero = kIgLanes + kIgPairs;              // a cell nobody writes: the source of a lane without that neighbour
constexpr int kIgRow = kIgLanes + kIgPairs + 8;           // doubles per (word, slot)
constexpr int kIgSlots = 4, kIgWords = 6;                 // a record is read at most two steps after it was written, an import lands one step early
constexpr unsigned kIgSlotB = kIgRow * 8u;
constexpr unsigned kIgWordB = kIgSlots * kIgSlotB;
constexpr unsigned kIgHand = kIgWords * kIgWordB;         // 62 976 bytes of hand-off records
constexpr unsigned kIgAst = kIgHand;                      // A's values of a step, staged by the loader waves: [4 slots][256 lanes][4]
constexpr unsigned kIgAstSlot = kIgLanes * 32u;
constexpr unsigned kIgSst = kIgAst + 4u * kIgAstSlot;     // L's values of a step, staged for the storer waves: [8 slots][256 lanes][4]
constexpr unsigned kIgLds = kIgSst + 8u * kIgAstSlot;     // 161 280 bytes (of the 163 840 a workgroup can have)
constexpr int kIgExp = 32;                                // exported lanes of a patch: y' = 15 (16), z' = 15 (15 more)
constexpr int kIgThreads = 576;                           // four consumer waves, the courier, two loader waves, two storer waves
#ifndef IG_NP
#define IG_NP 2
#endif
constexpr int kIgNP = IG_NP;                              // the imports of a step are polled for kIgNP + 1 steps ahead (256^3, final kernel: 2.60 ms with 0, 2.59 with 1, 2.55 with 2, 2.72 with 3, 2.84 with 5)
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
// what a lane (the x-line (y, z)) is, for the waves that work on it: the consumer, its loader, its storer
// ---------------------------------------------------------------------------------------------
struct IgLane {
    int yl, zl, y, z, sk;
    bool active, has2, has3;
    unsigned ua0, lenb;       // A: byte offset of the diagonal of row 0, bytes per inner row
    unsigned ul0, cub;        // L: byte offset of column 0, bytes per inner column
};
// a patch is SHEARED: its lane (y', z') is the line y = 16 ty + y' - z', z = 16 tz + z' -- the line (y+1, z-1) is the lane below, and
// every line a patch needs from another patch belongs to one with a smaller ticket (ty - 1 or tz - 1): with upright patches the
// neighbours in y would wait for each other column by column
__device__ __forceinline__ IgLane ig_lane(const IgArgs &A, const int ty, const int tz, const int t)
{
    IgLane l;
    l.yl = t & 15; l.zl = t >> 4;
    l.y = ty * 16 + l.yl - l.zl; l.z = tz * 16 + l.zl;
    l.active = l.y >= 0 && l.y < A.g.ny && l.z < A.g.nz;
    l.sk = 2 * l.yl + l.zl;
    l.has2 = l.active && l.y < A.g.ny - 1; l.has3 = l.active && l.z < A.g.nz - 1;
    // A: the upper part of row (x, y, z) starts at line start + entries left of the diagonal of row 0 + x * (entries of an inner row)
    const int edge = l.active ? ((l.y == 0) + (l.y == A.g.ny - 1) + (l.z == 0) + (l.z == A.g.nz - 1)) : 0;
    l.lenb = (unsigned)(7 - edge) * 8u;
    const long long ls = l.active ? grid_row_start(0, l.y, l.z, A.g) + (l.y > 0 ? 1 : 0) + (l.z > 0 ? 1 : 0) : 0;
    l.ua0 = (unsigned)ls * 8u;
    l.cub = (unsigned)(2 + (l.has2 ? 1 : 0) + (l.has3 ? 1 : 0)) * 8u;
    l.ul0 = l.active ? (unsigned)ig_col_start(0, l.y, l.z, A.g) * 8u : 0u;
    return l;
}

// ---------------------------------------------------------------------------------------------
// a consumer lane: the recurrence.  Its row of A comes from LDS (loader waves), its column of L goes to LDS (storer waves): the
// scattered 8-byte memory operations of a step -- eight per lane -- are issued by waves that have nothing else to do
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ig_consumer(const IgArgs &A, unsigned char *lds, const int ty, const int tz)
{
    const int t = threadIdx.x;
    const IgLane l = ig_lane(A, ty, tz, t);
    const int yl = l.yl, zl = l.zl, y = l.y, z = l.z, sk = l.sk;
    const bool active = l.active, has2 = l.has2, has3 = l.has3;
    const int nx = A.g.nx;
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
    const unsigned aSt = (unsigned)t * 32u;

    double e1p = 0.0, e2p = 0.0, e3p = 0.0, e1sqp = 0.0, e2sqp = 0.0, e3sqp = 0.0, qp = 0.0;
    bool bad = false;
    for (int sb = 0; sb < A.S; sb += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = sb + u;
            const int k = s - sk;
            const bool valid = active && (unsigned)k < (unsigned)nx;
            const bool has1 = valid && k < nx - 1;
            ST_BARRIER();
            const v2d wa = *reinterpret_cast<const v2d *>(lds + kIgAst + (unsigned)(u & 3) * kIgAstSlot + aSt);
            const v2d wb = *reinterpret_cast<const v2d *>(lds + kIgAst + (unsigned)(u & 3) * kIgAstSlot + aSt + 16u);
            const unsigned oA = (unsigned)((u - dA) & (kIgSlots - 1)) * kIgSlotB + aA;
            const unsigned oB = (unsigned)((u - dB) & (kIgSlots - 1)) * kIgSlotB + aB;
            const unsigned oC = (unsigned)((u - dC) & (kIgSlots - 1)) * kIgSlotB + aC;
            const double inE2 = st_lds(lds, IG_E2P * kIgWordB + oA), inQ = st_lds(lds, IG_QP * kIgWordB + oA), inF1 = st_lds(lds, IG_F1 * kIgWordB + oA);
            const double inE3 = st_lds(lds, IG_E3P * kIgWordB + oB), inF2 = st_lds(lds, IG_F2 * kIgWordB + oB);
            const double inF3 = st_lds(lds, IG_F3 * kIgWordB + oC);
            const double w0 = wa.x, w1 = wa.y, w2 = wb.x, w3 = wb.y;
            const double a0 = w0;
            const double a1 = has1 ? w1 : 0.0;
            const double a2 = has2 ? (has1 ? w2 : w1) : 0.0;
            const double a3m = has2 ? (has1 ? w3 : w2) : (has1 ? w2 : w1);
            const double a3 = has3 ? a3m : 0.0;
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
            const unsigned oM = (unsigned)(u & (kIgSlots - 1)) * kIgSlotB + aMe;
            *reinterpret_cast<double *>(lds + IG_E2P * kIgWordB + oM) = e2sqp;
            *reinterpret_cast<double *>(lds + IG_QP * kIgWordB + oM) = qp;
            *reinterpret_cast<double *>(lds + IG_F1 * kIgWordB + oM) = f1 * f1;
            *reinterpret_cast<double *>(lds + IG_F3 * kIgWordB + oM) = f3 * f3;
            *reinterpret_cast<double *>(lds + IG_E3P * kIgWordB + oM) = e3sqp;
            *reinterpret_cast<double *>(lds + IG_F2 * kIgWordB + oM) = f2 * f2;
            e1p = e1; e2p = e2; e3p = e3; e1sqp = e1sq; e2sqp = e2sq; e3sqp = e3sq; qp = q;
            // column j of L, for the storer waves
            {
                v2d sa, sb2;
                sa.x = p; sa.y = e1; sb2.x = e2; sb2.y = e3;
                *reinterpret_cast<v2d *>(lds + kIgSst + (unsigned)(u & 7) * kIgAstSlot + aSt) = sa;
                *reinterpret_cast<v2d *>(lds + kIgSst + (unsigned)(u & 7) * kIgAstSlot + aSt + 16u) = sb2;
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (t & 63) == 0) atomicOr(&A.ctrl[2], 1);
}

// ---------------------------------------------------------------------------------------------
// a loader wave.  The upper part of a lane's row is a 32-byte window of A.val, 56 bytes (a row) after the one before: two rows of a lane
// are fetched as four 16-byte pieces by four threads -- an instruction covers 16 lanes x (2 windows in 88 bytes) instead of 64 lanes x
// 8 bytes.  The lanes whose skew has the parity of the step take their turn together (128 lanes = eight instructions over the two
// loader waves); a pair of rows is requested four steps before its first row is due, waits in registers, and lands in the stage one
// step early.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ig_loader(const IgArgs &A, unsigned char *lds, const int ty, const int tz, const int lw)
{
    const int ln = threadIdx.x & 63;
    const int nx = A.g.nx;
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.aval), 0, (int)A.abytes, 0x00020000);
    const int piece = ln & 3;
    unsigned gua[8], glen[8], gst[8];
    int gsk[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = (lw * 4 + j) * 16 + (ln >> 2);
            const int zl = 2 * (q >> 4) + c, yl = q & 15;
            const IgLane g = ig_lane(A, ty, tz, 16 * zl + yl);
            gua[c * 4 + j] = g.ua0 + 16u * (unsigned)(piece & 1);
            glen[c * 4 + j] = g.lenb;
            gst[c * 4 + j] = (unsigned)(16 * zl + yl) * 32u + 16u * (unsigned)(piece & 1);
            gsk[c * 4 + j] = g.active ? g.sk : (1 << 28);          // (an idle lane never has a row)
        }
    }
#ifdef IG_X_NOLOAD
#define IGL_NOLOAD_(o) (o) = OOB
#else
#define IGL_NOLOAD_(o) (void)0
#endif
    v4u R[8][2];
#pragma unroll
    for (int e = 0; e < 8; ++e) { R[e][0] = v4u{0u, 0u, 0u, 0u}; R[e][1] = v4u{0u, 0u, 0u, 0u}; }
    // rows 2m, 2m+1 (m = (i_ + 1 - skew) / 2 + 2) of the lanes of class c_: requested in iteration i_
#define IGL_LOAD(c_, j_, gen_, i_)                                                                           \
    do {                                                                                                     \
        const int kk_ = (i_) + 5 - gsk[(c_) * 4 + (j_)] + (piece >> 1);                                      \
        unsigned o_ = (unsigned)kk_ < (unsigned)nx ? gua[(c_) * 4 + (j_)] + (unsigned)kk_ * glen[(c_) * 4 + (j_)] : OOB; \
        IGL_NOLOAD_(o_);                                                                                     \
        asm volatile("" : "+v"(o_));                                                                         \
        R[(c_) * 4 + (j_)][gen_] = __builtin_amdgcn_raw_buffer_load_b128(ra, o_, 0, 0);                      \
    } while (0)
    // ... and put where the consumers read them in the steps i_ + 1, i_ + 2 (rows (i_ + 1 - skew) + {0, 1})
#define IGL_WRITE(c_, j_, gen_, i_)                                                                          \
    do {                                                                                                     \
        const unsigned slot_ = (unsigned)(((i_) + 1 + (piece >> 1)) & 3);                                    \
        *reinterpret_cast<v4u *>(lds + kIgAst + slot_ * kIgAstSlot + gst[(c_) * 4 + (j_)]) = R[(c_) * 4 + (j_)][gen_]; \
    } while (0)
#define IGL_EVENT(c_, gen_, i_, wr_)                                                                         \
    do {                                                                                                     \
        if (wr_) { IGL_WRITE(c_, 0, gen_, i_); IGL_WRITE(c_, 1, gen_, i_); IGL_WRITE(c_, 2, gen_, i_); IGL_WRITE(c_, 3, gen_, i_); } \
        IGL_LOAD(c_, 0, gen_, i_); IGL_LOAD(c_, 1, gen_, i_); IGL_LOAD(c_, 2, gen_, i_); IGL_LOAD(c_, 3, gen_, i_); \
    } while (0)
    // iteration i serves the class (i + 1) & 1; its register generation alternates: ((i + 1 - class) / 2) & 1
    IGL_EVENT(0, 0, -5, false); IGL_EVENT(1, 0, -4, false); IGL_EVENT(0, 1, -3, false); IGL_EVENT(1, 1, -2, false);
    IGL_EVENT(0, 0, -1, true);
    for (int ib = 0; ib < A.S; ib += 4) {
        ST_BARRIER(); IGL_EVENT(1, 0, ib, true);
        ST_BARRIER(); IGL_EVENT(0, 1, ib + 1, true);
        ST_BARRIER(); IGL_EVENT(1, 1, ib + 2, true);
        ST_BARRIER(); IGL_EVENT(0, 0, ib + 3, true);
    }
#undef IGL_EVENT
#undef IGL_LOAD
#undef IGL_WRITE
#undef IGL_NOLOAD_
}

// ---------------------------------------------------------------------------------------------
// a storer wave.  A lane's columns are contiguous in L (four entries each for a lane inside the box): four finished columns are one
// 128-byte run, written as eight 16-byte pieces by eight threads -- an instruction covers 8 lanes x 128 bytes instead of 64 lanes x 8.
// The lanes whose column k = 3 (mod 4) was finished in the previous step form one of four classes of 64 lanes (skew mod 4); the two
// storer waves take eight of them per instruction.  What the groups do not cover -- lanes on the box' faces (shorter columns), the
// last columns of every lane (the final one has no entry below the diagonal in x) -- goes out entry by entry as before.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ig_storer(const IgArgs &A, unsigned char *lds, const int ty, const int tz, const int sw)
{
    typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
    const int ln = threadIdx.x & 63;
    const int nx = A.g.nx;
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(A.lval, 0, (int)A.lbytes, 0x00020000);
    const IgLane l0 = ig_lane(A, ty, tz, (2 * sw) * 64 + ln), l1 = ig_lane(A, ty, tz, (2 * sw + 1) * 64 + ln);
    const unsigned st0 = (unsigned)((2 * sw) * 64 + ln) * 32u, st1 = (unsigned)((2 * sw + 1) * 64 + ln) * 32u;
    const int kcut = ((nx - 1) / 4) * 4;                 // columns from here on are not part of a full group of four
    // the grouped path: for class c (= step & 3) and instruction j, this thread's lane and piece
    const int piece = ln & 7;
    unsigned gul[16], gst[16];
    int gsk[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = (sw * 4 + j) * 8 + (ln >> 3);
            const int zl = 2 * (q >> 3) + (c & 1);
            const int yl = 2 * (q & 7) + (((c - zl) & 3) >> 1);
            const IgLane g = ig_lane(A, ty, tz, 16 * zl + yl);
            const bool grouped = g.active && g.has2 && g.has3;
            gul[c * 4 + j] = g.ul0 + 16u * (unsigned)piece;
            gst[c * 4 + j] = (unsigned)(16 * zl + yl) * 32u + (unsigned)(piece & 1) * 16u;
            gsk[c * 4 + j] = grouped ? g.sk : (1 << 28);             // (a lane outside the grouped path never has a column to store here)
        }
    }
#ifdef IG_X_NOSTORE
#define IGS_NOSTORE_(o) (o) = OOB
#else
#define IGS_NOSTORE_(o) (void)0
#endif
    // the columns k-3 .. k of the lanes of class c_ whose column k (= 3 mod 4) was written in step s_
#define IGS_GROUP(c_, j_, s_)                                                                                \
    do {                                                                                                     \
        const int k_ = (s_) - gsk[(c_) * 4 + (j_)];                                                          \
        const bool ok_ = k_ >= 3 && k_ < nx - 1;                                                             \
        const int kk_ = k_ - 3 + (piece >> 1);                                                               \
        const unsigned slot_ = (unsigned)((kk_ + gsk[(c_) * 4 + (j_)]) & 7);                                 \
        const v4u v_ = *reinterpret_cast<const v4u *>(lds + kIgSst + slot_ * kIgAstSlot + gst[(c_) * 4 + (j_)]); \
        unsigned o_ = ok_ ? gul[(c_) * 4 + (j_)] + (unsigned)(k_ - 3) * 32u : OOB;                           \
        IGS_NOSTORE_(o_);                                                                                    \
        asm volatile("" : "+v"(o_));                                                                         \
        __builtin_amdgcn_raw_buffer_store_b128(v_, rl, o_, 0, 0);                                            \
    } while (0)
    // entry by entry: the column written in step s_ of a lane the groups do not serve, or beyond the last full group
#define IGS_STORE(l_, st_, s_)                                                                               \
    do {                                                                                                     \
        const int k_ = (s_) - (l_).sk;                                                                       \
        const bool valid_ = (l_).active && (unsigned)k_ < (unsigned)nx && (!((l_).has2 && (l_).has3) || k_ >= kcut); \
        if (__builtin_amdgcn_ballot_w64(valid_) != 0) {                                                      \
            const bool has1_ = valid_ && k_ < nx - 1;                                                        \
            const v2d a_ = *reinterpret_cast<const v2d *>(lds + kIgSst + (unsigned)((s_) & 7) * kIgAstSlot + (st_)); \
            const v2d b_ = *reinterpret_cast<const v2d *>(lds + kIgSst + (unsigned)((s_) & 7) * kIgAstSlot + (st_) + 16u); \
            const unsigned pos_ = (l_).ul0 + (unsigned)k_ * (l_).cub;                                        \
            unsigned o0_ = valid_ ? pos_ : OOB;                                                              \
            unsigned o1_ = has1_ ? pos_ + 8u : OOB;                                                          \
            unsigned o2_ = (valid_ && (l_).has2) ? pos_ + 8u + (has1_ ? 8u : 0u) : OOB;                      \
            unsigned o3_ = (valid_ && (l_).has3) ? pos_ + 8u + (has1_ ? 8u : 0u) + ((l_).has2 ? 8u : 0u) : OOB; \
            IGS_NOSTORE_(o0_); IGS_NOSTORE_(o1_); IGS_NOSTORE_(o2_); IGS_NOSTORE_(o3_);                      \
            asm volatile("" : "+v"(o0_), "+v"(o1_), "+v"(o2_), "+v"(o3_));                                   \
            /* (scalars first: __builtin_bit_cast of a vector ELEMENT expression takes the vector's first element whichever is named) */ \
            const double v0_ = a_.x, v1_ = a_.y, v2_ = b_.x, v3_ = b_.y;                                     \
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, v0_), rl, o0_, 0, 0);             \
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, v1_), rl, o1_, 0, 0);             \
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, v2_), rl, o2_, 0, 0);             \
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_, v3_), rl, o3_, 0, 0);             \
        }                                                                                                    \
    } while (0)
    for (int ib = 0; ib < A.S; ib += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = ib + u;
            ST_BARRIER();
            if (i >= 1) {
                // the step before: i - 1; its class (skew = step - 3 mod 4) is (u + 4 - 1 - 3) & 3 = u
                IGS_GROUP(u, 0, i - 1); IGS_GROUP(u, 1, i - 1); IGS_GROUP(u, 2, i - 1); IGS_GROUP(u, 3, i - 1);
                IGS_STORE(l0, st0, i - 1); IGS_STORE(l1, st1, i - 1);
            }
        }
    }
#undef IGS_GROUP
#undef IGS_STORE
#undef IGS_NOSTORE_
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
    unsigned spins = 0;
#define IGC_WAIT(v_, w_, s_)                                                                                 \
    do {                                                                                                     \
        while (!dead && __builtin_amdgcn_ballot_w64(need && v_ == kSentinel) != 0) {                         \
            if (need && v_ == kSentinel) v_ = ld_agent_u64(IGC_ADDR(s_, w_));                                \
            __builtin_amdgcn_s_waitcnt(0x0F70);                                                              \
            __builtin_amdgcn_s_sleep(1);                                                                     \
            if ((++spins & 255u) == 0) {                                                                     \
                if (spins > kStSpinLimit) atomicExch(&A.ctrl[1], 1);                                         \
                const int e = ld_agent_i32(&A.ctrl[1]);                                                      \
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                          \
                if (spins > kStSpinLimit || e != 0) { dead = true; break; }                                  \
            }                                                                                                \
        }                                                                                                    \
    } while (0)
    // deliver what the lanes read at step s_ (slot s_ & 7), then poll for step s_ + kIgNP + 1
#define IGC_DELIVER(s_, i_)                                                                                  \
    do {                                                                                                     \
        const bool need = exists && (unsigned)((s_) - skc) < (unsigned)nx;                                   \
        unsigned long long v0 = gq[(i_) % (kIgNP + 1)][0], v1 = gq[(i_) % (kIgNP + 1)][1], v2 = gq[(i_) % (kIgNP + 1)][2]; \
        /* (one loop per word: with the three words in ONE loop hipcc waits for every outstanding memory operation at the loop's  \
           header -- the exports just issued, the polls of the steps ahead --, i.e. once per step: 1.82 -> ... ms without memory ops) */ \
        IGC_WAIT(v0, w0, s_); IGC_WAIT(v1, w1, s_); IGC_WAIT(v2, w2, s_);                                    \
        if (!need) { v0 = 0; v1 = 0; v2 = 0; }                                                               \
        const unsigned o_ = (unsigned)((s_) & (kIgSlots - 1)) * kIgSlotB + dst;                                           \
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
            {
                ST_BARRIER();
                // the records of step i - 1 of the exported lanes.  UNCONDITIONAL stores (a lane or step without an export stores beyond
                // the buffer's end, which drops it): hipcc's wait for a poll of three steps ago counts the memory operations issued since, and
                // takes the smallest count over all paths -- with the exports under a branch it waited for the exports of the step before
                const int m = i - 1 - ske;
                const bool ex = et >= 0 && i >= 1 && (unsigned)m <= (unsigned)nx;
                const unsigned o = (unsigned)((i - 1) & (kIgSlots - 1)) * kIgSlotB + (unsigned)(et >= 0 ? et : 0) * 8u;
                unsigned long long w[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    w[j] = *reinterpret_cast<const unsigned long long *>(lds + (unsigned)j * kIgWordB + o);
                    w[j] = w[j] == kSentinel ? kCanonNaN : w[j];
                }
                // (buffer stores, sc1 = write-through: an inline-asm store would be a memory operation hipcc's wait counts do not know of --
                // the polls behind it were then taken for complete one operation early)
                unsigned qo = ex ? eoff + (unsigned)m * 64u : 0xffffff00u;
                asm volatile("" : "+v"(qo));
                __builtin_amdgcn_raw_buffer_store_b128(v4u{(unsigned)w[0], (unsigned)(w[0] >> 32), (unsigned)w[1], (unsigned)(w[1] >> 32)}, rx, qo, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(v4u{(unsigned)w[2], (unsigned)(w[2] >> 32), (unsigned)w[3], (unsigned)(w[3] >> 32)}, rx, qo + 16u, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(v4u{(unsigned)w[4], (unsigned)(w[4] >> 32), (unsigned)w[5], (unsigned)(w[5] >> 32)}, rx, qo + 32u, 0, 16);
                IGC_DELIVER(i + 1, u + 1);
            }
        }
    }
#undef IGC_DELIVER
#undef IGC_WAIT
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
    const int wv = threadIdx.x >> 6;
    if (wv < 4) ig_consumer(A, lds, ty, tz);
    else if (wv == 4) ig_courier(A, lds, ty, tz, wg);
    else if (wv < 7) ig_loader(A, lds, ty, tz, wv - 5);
    else ig_storer(A, lds, ty, tz, wv - 7);
}

}  // namespace

IcholtGridJob::~IcholtGridJob()
{
    // (a construction that unwinds between launch and finish: the kernel may still run on what this object gives back to the pool)
    if (launched_on && !finished) { (void)hipStreamSynchronize(launched_on); if (side_stream) (void)hipStreamSynchronize(side_stream); }
    if (pattern_done) (void)hipEventDestroy(pattern_done);
}

// Queues everything on st and returns: the caller's work that needs nothing of L (`pattern_free`: the sweeps' schedule from the grid's
// dimensions), L's index arrays (closed form), the caller's work on them (`after_pattern`: the general schedule pass), the proof of the grid (k_grid_check), the exchange buffer's sentinels,
// the kernel, the read-back of its verdict.  *L owns the arrays at once.  false: a grid outside the kernel's limits, nothing was queued.
bool icholt_grid_launch(hipStream_t st, hipStream_t side, const DevMat &A, const GridDims &g, int32_t *ctrl, DevMat *L, IcholtGridJob *job,
                        const std::function<void(hipStream_t)> &pattern_free, const std::function<void(hipStream_t)> &after_pattern)
{
    static const bool off = getenv("ILUPP_NO_ICHOLT_GRID") != nullptr;
    if (off) return false;
    const int64_t n = A.n;
    const int64_t nnzL = (A.nnz + n) / 2;
    const int nty = (g.ny + 15 + 15) / 16, ntz = (g.nz + 15) / 16;        // (sheared patches: y + z' runs to ny - 1 + 15)
    // buffer resources cover the value arrays
    if (A.nnz * 8 >= (1LL << 32) || nnzL >= (1LL << 31) || (int64_t)nty * ntz > (1 << 20)) return false;
    {
        static std::once_flag once[64];      // once per device
        int dev = 0;
        ILUPP_HIP(hipGetDevice(&dev));
        std::call_once(once[dev & 63], [] {
            ILUPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_icholt_grid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kIgLds));
        });
    }
    const int64_t xwords = (int64_t)nty * ntz * kIgExp * (g.nx + 1) * 8;
    ILUPP_HIP(job->xch.alloc(sizeof(unsigned long long) * (size_t)(xwords + 8)));
    unsigned long long *xp = job->xch.as<unsigned long long>();
    ILUPP_HIP(job->ev.create());
    ILUPP_HIP(hipEventCreateWithFlags(&job->pattern_done, hipEventDisableTiming));
    job->g = g;
    job->launched_on = st; job->side_stream = side;
    L->release();
    L->n = A.n; L->nnz = nnzL; L->is_csr = false; L->owns = true;
    ILUPP_HIP(pool_malloc(&L->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(pool_malloc(&L->idx, sizeof(int32_t) * (size_t)nnzL));
    ILUPP_HIP(pool_malloc(&L->val, sizeof(double) * (size_t)nnzL));
    ILUPP_HIP(hipMemsetAsync(ctrl, 0, sizeof(int32_t) * 16, st));
    // Before the kernel, on two streams: L's index arrays and what the caller makes of them (the sweeps' schedule) on the side stream;
    // the proof of the grid and the exchange buffer's sentinels on this one.  The kernel starts when both are through: anything that
    // streams beside it costs several times its own duration (the patches of the kernel hold every CU; measured: the schedule's
    // 0.15 ms became 0.9 ms, the proof's 0.12 ms 0.5 ms, and the kernel 0.2 ms longer).
    hipStream_t q = side ? side : st;
    if (pattern_free) pattern_free(q);
    if (!job->pattern_written)
        hipLaunchKernelGGL(k_icholt_grid_pattern, dim3(2048), dim3(256), 0, q, A.n, g, L->ptr, L->idx, (long long)nnzL);
    if (after_pattern) after_pattern(q);
    ILUPP_HIP(hipEventRecord(job->pattern_done, q));
    grid_check_launch(st, A, g, ctrl + 8);
    fill_u64(st, xp, xwords, kSentinel);
    ILUPP_HIP(hipMemsetAsync(xp + xwords, 0, 64, st));
    if (side) ILUPP_HIP(hipStreamWaitEvent(st, job->pattern_done, 0));
    IgArgs a;
    a.g = g; a.nty = nty; a.ntz = ntz;
    a.S = ((g.nx + kIgMaxSkew + 3) + 23) / 24 * 24;        // (a multiple of the consumers' 8 and of the courier's 4 (kIgNP + 1) unrolled steps)
    static_assert(24 % (4 * (kIgNP + 1)) == 0, "the courier's unrolled loop must divide the step count");
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
    job->finished = true;
    float ms = 0.f;
    ILUPP_HIP(hipEventElapsedTime(&ms, job->ev.a, job->ev.b));
    if (kernel_ms) *kernel_ms = ms;
    const int32_t *h = job->h;
    if (getenv("ILUPP_IG_DEBUG"))
        fprintf(stderr, "icholt_grid: %d x %d x %d tickets %d timeout %d premise %d grid %d kernel %.3f ms\n", job->g.nx, job->g.ny, job->g.nz, h[0], h[1], h[2], h[8], ms);
    return h[1] == 0 && h[2] == 0 && h[8] == 0;
}

}  // namespace ilupp
