// ilupp_amd/csrc/st_common.h -- what the static level-major kernels (st.hip, st_direct.hip) share: value markers, the barrier of a
// step, the descriptor of a value that comes from an earlier workgroup.
#pragma once

#include "common.h"

namespace ilupp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define ST_STREAM_STORE(v, p) __builtin_nontemporal_store(v, p)
static constexpr int kStH = 8;                 // steps of hand-off history kept in LDS = steps the streams are read ahead
#ifndef ST_PS
#define ST_PS 2
#endif
#ifndef ST_PF
#define ST_PF 2
#endif
static constexpr int kStPF = ST_PF, kStPS = ST_PS;   // steps ahead the courier polls the values of earlier workgroups: factor kernel, sweeps.  (A tile settles
                                                       // about (poll distance + 3) steps + one trip through memory behind the tile it reads from: round 4, 256^3: factor kernel 1.13 -> 1.03 ms with 2 instead of 8)
static constexpr int kStMaxSkew = 30000;
static constexpr unsigned kStSpinLimit = 1u << 21;
static constexpr int64_t kStMaxChunks = 1 << 21;      // record offsets are 32-bit byte offsets

struct __attribute__((aligned(8))) D2s { double v[2]; };

#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long st_bits(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double st_dbl(unsigned long long b) { return __longlong_as_double((long long)b); }
// a value that enters the records must not look like one of the two markers
__device__ __forceinline__ double st_clean(double x)
{
    const unsigned long long b = st_bits(x);
    return (b == kSentinel || b == kAbsent) ? st_dbl(kCanonNaN) : x;
}

// the barrier of a step: this wave's LDS writes of the previous step have landed, then everybody's have
// (NOT __syncthreads(): that would also drain the global loads in flight, i.e. the read-ahead)
#define ST_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ double st_lds(const unsigned char *base, unsigned off) { return *reinterpret_cast<const double *>(base + off); }
__device__ __forceinline__ int st_med3(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
#endif

#if defined(__HIPCC__)
// ---- box grids (grid.hip, icholt_grid.hip) ----
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

#endif

// ---- the wave-exchange kernels (st_wave.hip) -------------------------------------------------------------------------
// The lanes of a workgroup are a 16 x 16 patch of chains, lane = 16 z + y, a wave = 16 x 4 of them.  A dependency on the
// chain one to the left (lane - 1, same row of 16 lanes) or one below (lane - 16, same wave) whose value is ONE step old never
// leaves the wave's registers: DPP row_shr:1 / ds_bpermute.  Everything else that comes from the workgroup (other waves) is at
// least kWrLag steps old -- the skews are computed with that weight on such edges -- and is read from the hand-off array one
// step early, so that no LDS round trip and no barrier lies on the chain of a step.
static constexpr int kWrLag = 2;
enum { WR_NONE = 0, WR_C = 1, WR_B = 2, WR_A = 3 };      // class of a dependency: lane - 16 or its stand-in / lane - 1 or its stand-in / own chain
#if defined(__HIPCC__)
__device__ __forceinline__ bool wr_fast_edge(int t, int lane) { return (lane == t - 1 && (t & 15) != 0) || (lane == t - 16 && (t & 63) >= 16); }
// weight of an in-workgroup edge in the skew fixpoint (st_link_body): 1 for what stays in registers
__device__ __forceinline__ int wr_edge_lag(int t, int lane, bool wave_exchange) { return (wave_exchange && !wr_fast_edge(t, lane)) ? kWrLag : 1; }
// Classes of a lane's (at most three) dependencies, from its table entry (after the link pass: ST_DT holds the age of each value).
// cls[j]: WR_* of template position j; ring[j]: the value comes from the hand-off array (another wave, the courier).  The classes
// of the positions ascend C, B, A in a forward schedule (the order of the columns left of the diagonal on a mesh) and descend
// A, B, C in a backward one.  false: not a lane for these kernels.
__device__ __forceinline__ bool wr_classify(const int32_t *T, const int t, const bool bwd, int cls[3], bool ring[3])
{
    const int nd = T[ST_ND], cnt = T[ST_CNT];
    cls[0] = cls[1] = cls[2] = WR_NONE; ring[0] = ring[1] = ring[2] = false;
    if (cnt <= 0 || nd <= 0) return true;
    if (nd > 3) return false;
    int want[3] = {0, 0, 0};                           // by canonical index q (classes ascend with q): 0 = hand-off array, else the class the registers dictate
    bool rq[3] = {false, false, false};
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q < nd) {
            const int j = bwd ? nd - 1 - q : q;
            const int sw = T[ST_SRC + j], ty = sw & 3, lane = (sw >> 2) & 255, dt = T[ST_DT + j];
            if (ty == ST_OWN) want[q] = WR_A;
            else if (ty == ST_LOCAL && dt == 1 && lane == t - 1 && (t & 15) != 0) want[q] = WR_B;
            else if (ty == ST_LOCAL && dt == 1 && lane == t - 16 && (t & 63) >= 16) want[q] = WR_C;
            else if (ty == ST_GHOST || (ty == ST_LOCAL && dt >= kWrLag && dt <= kStH - 1)) { want[q] = 0; rq[q] = true; }
            else ok = false;
        }
    }
    // the hand-off values take the classes that are free, from the top
    int prev = 4;
    int cq[3] = {WR_NONE, WR_NONE, WR_NONE};
#pragma unroll
    for (int qq = 0; qq < 3; ++qq) {
        const int q = 2 - qq;
        if (q < nd) {
            int c = want[q];
            if (c == 0) c = prev > WR_B ? WR_B : WR_C;
            if (c >= prev) ok = false;
            // a hand-off value of class B has to sit in a lane whose lane - 1 is not a row-mate (the DPP move leaves exactly those
            // lanes alone); of class C: any lane (it is selected)
            if (rq[q] && c == WR_B && (t & 15) != 0) ok = false;
            cq[q] = c; prev = c;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < nd) { const int j = bwd ? nd - 1 - q : q; cls[j] = cq[q]; ring[j] = rq[q]; }
    return ok;
}
// where the coefficient of a class sits in a class-aligned record (format 1): accumulation order = slot order = ascending column
__device__ __forceinline__ int wr_slot_of(const int cls, const bool bwd) { return bwd ? 3 - cls : cls - 1; }
// what the class-aligned records and the kernels without per-step selects (st_wave.hip) need beyond the classes: a coefficient that
// does not exist is +0.0 and meets an unknown that is +0.0 too (the cell of zeros for a whole class; the lane's own unknown before
// its first row), so every entry of the template exists for every row of the lane but the own-chain one of the first row
__device__ __forceinline__ bool wx_lane_ok(const int32_t *T, const int t, const bool bwd)
{
    int cls[3]; bool ring[3];
    if (!wr_classify(T, t, bwd, cls, ring)) return false;
    const int nd = T[ST_ND], cnt = T[ST_CNT];
    if (cnt <= 0) return true;
    bool ok = true, own = false, hasB = false;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nd) {
            if ((T[ST_SRC + j] & 3) == ST_OWN) own = true;
            else if (T[ST_KLO + j] > 0 || T[ST_KHI + j] < cnt) ok = false;
            if (cls[j] == WR_B) hasB = true;
        }
    }
    if (!own && cnt > 1) ok = false;
    if (!hasB && (t & 15) != 0) ok = false;          // (the DPP move hands such a lane its neighbour's unknown whatever the template says)
    return ok;
}
#endif

// forward-lane fields of the lane table that only the direct-feed factor kernel (st_direct.hip) reads
enum { ST_P0 = 26, ST_DFL = 27, ST_Q = 28 };
// ST_DFL: entries right of the diagonal | own-chain entry left << 2 | own-chain entry right << 3 | entries per full row << 4
// ST_Q + j: which of the producer row's entries right of its diagonal is the transposed entry of dependency j (-1: none)

static constexpr int kSdHist = 4;     // hand-off slots of the direct-feed kernel: an in-workgroup dependency lies at most kSdHist-1 steps back

#if defined(__HIPCC__)
// the direct-feed kernel's lane fields, and its premise at lane level (dflags |= 1: a chain without its backward lane, |= 2: a
// lane whose entries are not produced where its template says for ALL of its rows, or whose own-chain entries are not r-1 / r+1)
__device__ __forceinline__ void sd_tab_lane(const int f, int32_t *__restrict__ ltabF, const int32_t *__restrict__ ltabB, const int32_t *__restrict__ uslot,
                                            const int32_t *__restrict__ Aptr, int32_t *__restrict__ dflags)
{
    int32_t *T = ltabF + (size_t)f * kStTab;
    const int cnt = T[ST_CNT], nd = T[ST_ND];
    T[ST_P0] = 0; T[ST_DFL] = 0; T[ST_Q] = -1; T[ST_Q + 1] = -1; T[ST_Q + 2] = -1;
    if (cnt <= 0) return;
    int bad = 0;
    const int su = uslot[f];
    if (su < 0) { atomicOr(dflags, 1); return; }
    const int32_t *TB = ltabB + (size_t)su * kStTab;
    const int ndU = TB[ST_ND];
    if (TB[ST_CNT] != cnt) bad = 1;
    int ownL = 0, ownU = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nd) {
            const int ty = T[ST_SRC + j] & 3;
            if (ty == ST_OWN) {
                // the own-chain entry: column r - 1, the last one left of the diagonal
                if (j != nd - 1 || T[ST_OFF + j] != -1) bad = 1;
                ownL = 1;
            } else {
                // every row of the lane has the entry, and its producer is where the template says
                if (T[ST_KLO + j] > 0 || T[ST_KHI + j] < cnt) bad = 1;
                if (ty == ST_LOCAL && (T[ST_DT + j] < 1 || T[ST_DT + j] > kSdHist - 1)) bad = 1;
            }
        }
        if (j < ndU) {
            const int ty = TB[ST_SRC + j] & 3;
            if (ty == ST_OWN) {
                if (j != 0 || TB[ST_OFF + j] != 1) bad = 1;
                ownU = 1;
            } else {
                if (TB[ST_KLO + j] > 0 || TB[ST_KHI + j] < cnt) bad = 1;
            }
        }
    }
    // the transposed entry of dependency j: the entry of the pivot row's right side whose offset is the opposite one
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int q = -1;
        if (j < nd) {
            const int os = T[ST_SRC + j] >> 2;
            const int pu = uslot[os];
            if (pu < 0) {
                bad = 1;
            } else {
                const int32_t *TP = ltabB + (size_t)pu * kStTab;
#pragma unroll
                for (int p = 0; p < 3; ++p) if (p < TP[ST_ND] && TP[ST_OFF + p] == -T[ST_OFF + j]) q = p;
            }
        }
        T[ST_Q + j] = q;
    }
    T[ST_P0] = Aptr[T[ST_FIRST]];
    T[ST_DFL] = ndU | (ownL << 2) | (ownU << 3) | ((nd + 1 + ndU) << 4);
    if (bad) atomicOr(dflags, 2);
}

#endif

#if defined(__HIPCC__)
// ---- what the sweep kernels of st.hip and st_wave.hip share ----------------------------------------------------------
struct StSArgs {
    const v2d *pk;                            // 2 x 64 x 16 B per chunk: {v0,v1}{v2,vdiag}, dependencies in accumulation order
    const int32_t *ltab, *wtab;
    int32_t n, nchY;                          // nchY: first spare chunk (one per wave) of ylm
    double *xlm;                              // level-major in the FORWARD sweep's order: the right-hand side, overwritten with the intermediate vector
    double *ylm;                              // level-major in the backward sweep's order: the result
    const int32_t *ysrc;                      // backward: where in xlm the lane's row 0 is (row k: - 64 k)
    int32_t xlm_chunks;                       // chunks of xlm (and of the backward sweep's records)
    const int32_t *xe, *xw;                   // the exchange between workgroups (PackedSweep::xe, xw, xch): all-sentinel before the sweep
    double *xch;
    int32_t *ticket, *err;
    double *nat;                              // (st_wave.hip, vector wave) the caller's vector in natural order: read by the forward sweep, written by the backward one
};

#ifndef ST_CSLEEP
#define ST_CSLEEP 1
#endif
#ifndef ST_SOLO
#define ST_SOLO (48 * 1024)
#endif
static constexpr int kStSoloLds = ST_SOLO;      // dynamic LDS nobody uses: > 80 KB per workgroup in total
#ifndef ST_RA
#define ST_RA 16
#endif
static constexpr int kStRA = ST_RA;            // steps the streams of a sweep are read ahead (a multiple of kStH)
static constexpr int kStRow = kThreads + 64;     // doubles per slot of the hand-off array: the lanes, then the courier's pairs
static constexpr int kStWgThreads = kThreads + 64;

struct StPair { int idx0, stride, sk, klo, khi; };   // the value of step s is xch[idx0 + s * stride]; the lane's skew; the k = s - sk that have it

// where a lane's dependencies come from: the LDS read address of each (the value of `dt` steps ago of lane `u`, in the copy 8
// slots up: slot index = step % 8 + 8 - dt stays inside [1, 15] with the step's immediate), and for those of earlier workgroups
// where in the exchange the producer lane's value of step s = 0 would be and how far apart steps are
__device__ __forceinline__ void st_lane_sources(const int32_t *T, const int t, const int32_t *ltab, const int32_t *xe, const int32_t *xw,
                                                bool isg[3], int idx0[3], int stride[3], unsigned va[3], const int row = kStRow)
{
    const int nd = T[ST_ND], cnt = T[ST_CNT];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int sw = T[ST_SRC + j];
        const int ty = (j < nd && cnt > 0) ? (sw & 3) : ST_NONE;
        const int u = ty == ST_LOCAL ? ((sw >> 2) & 255) : t;
        const int dt = ty == ST_LOCAL ? T[ST_DT + j] : 1;          // (own previous row: lane t, one step back)
        va[j] = (unsigned)(((kStH - dt) * row + u) * 8);
        isg[j] = ty == ST_GHOST;
        idx0[j] = 0; stride[j] = 0;
        if (isg[j]) {
            // the producer lane's value of ITS step s' = k' + skew' with k' = k + T[ST_KAP + j], k = s - skew
            const int os = sw >> 2, pw = os >> 8;
            const int E = xw[pw * 4];
            stride[j] = E;
            idx0[j] = xw[pw * 4 + 3] + (T[ST_KAP + j] + ltab[(size_t)os * kStTab + ST_SKEW] - T[ST_SKEW] - xw[pw * 4 + 1]) * E + xe[os];
        }
    }
}

// The pairs of a workgroup, numbered: lane t's ghost dependency j gets the next free index p, its descriptor goes to pairs[p],
// and the lane reads it like any hand-off value: slot "this step", lane 256 + p.  Returns the dependency's LDS read address.
// (called by the 256 lanes of the schedule; s_cnt[4]: scratch; *s_total: the number of pairs)
__device__ __forceinline__ void st_number_pairs(const int32_t *T, const int t, const bool isg[3], const int idx0[3], const int stride[3],
                                                unsigned va[3], StPair *pairs, int *s_cnt, int *s_total, const int row = kStRow)
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
    if (t == 0) *s_total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (isg[j]) {
            const int p = before + __builtin_amdgcn_mbcnt_hi((unsigned)(bal[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[j], 0));
            if (p < 64) {
                StPair d;
                d.idx0 = idx0[j]; d.stride = stride[j]; d.sk = T[ST_SKEW];
                d.klo = max(T[ST_KLO + j], 0); d.khi = max(min(T[ST_KHI + j], T[ST_CNT]), d.klo);
                pairs[p] = d;
            }
            va[j] = (unsigned)((kStH * row + kThreads + min(p, 63)) * 8);
        }
        before += __popcll(bal[j]);
    }
}

#endif

#if defined(__HIPCC__)
// may the wave-exchange factor kernel (st_wave.hip: k_ilu0_wx) run this forward lane?  On top of wx_lane_ok: the transposed entry of
// every elimination exists and is an entry its owner hands on (a'B or a'C of the pivot row; the own chain's: a'A), and what comes
// from the workgroup is at most kSdHist - 1 steps old.  (after sd_tab_lane: ST_Q is set)
__device__ __forceinline__ bool wf_lane_ok(const int32_t *T, const int32_t *__restrict__ ltabB, const int32_t *__restrict__ uslot)
{
    const int nd = T[ST_ND], cnt = T[ST_CNT];
    if (cnt <= 0) return true;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nd) {
            const int sw = T[ST_SRC + j], ty = sw & 3, os = sw >> 2, q = T[ST_Q + j];
            if (ty == ST_OWN) {
                if (q != 0) ok = false;
            } else {
                const int pu = uslot[os];
                if (q < 0 || pu < 0) {
                    ok = false;
                } else {
                    int pc[3]; bool pr[3];
                    (void)wr_classify(ltabB + (size_t)pu * kStTab, pu & 255, true, pc, pr);
                    const int c = q == 0 ? pc[0] : (q == 1 ? pc[1] : pc[2]);
                    if (c != WR_B && c != WR_C) ok = false;
                }
                if (ty == ST_LOCAL && (T[ST_DT + j] < 1 || T[ST_DT + j] > kSdHist - 1)) ok = false;
            }
        }
    }
    return ok;
}
#endif

// st_direct.hip
bool st_direct_prepare(hipStream_t st, const DevMat &A, const Schedule &fwd, int32_t *dflags);
void st_direct_verify(hipStream_t st, const DevMat &A, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu, int32_t *dflags);
int ilu0_numeric_sd(hipStream_t st, const DevMat &A, PackedSweep *pl, PackedSweep *pu, int32_t *d_ctrl, float *kernel_ms,
                    hipEvent_t e0, hipEvent_t e1);
// st_wave.hip
int ilu0_numeric_wx(hipStream_t st, const DevMat &A, PackedSweep *pl, PackedSweep *pu, int32_t *d_ctrl, float *kernel_ms,
                    hipEvent_t e0, hipEvent_t e1);

}  // namespace ilupp
