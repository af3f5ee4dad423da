// ilupp_amd/csrc/common.h -- shared host/device helpers of the MI355X ILU engine (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
#include <utility>
#include <string>
#include <vector>

#include "../../include/ilupp_hip.h"

namespace ilupp {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const std::string &msg);
struct HipError { hipError_t code; const char *what; const char *file; int line; };

#define ILUPP_HIP(expr)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) throw ilupp::HipError{e_, #expr, __FILE__, __LINE__};          \
    } while (0)

// ---- pooled device memory ---------------------------------------------------------------------
// hipMalloc/hipFree of multi-GB buffers cost milliseconds and serialise the device; a factorisation
// that is repeated (time stepping, refactorisation, the benchmark loop) asks for the same sizes every
// time, so freed blocks are kept in a size-keyed pool (bounded) and handed out again.
hipError_t pool_malloc(void **p, size_t bytes);
hipError_t pool_free(void *p);
void pool_trim();
void pool_set_owner(int owner);      // the calling thread's owner lane (pool.h): kept blocks go back only to the owner that freed them ...
void pool_disown(int owner);         // ... until it has synchronised its stream and calls this
void pool_set_limit(size_t bytes);
size_t pool_cached_bytes();
size_t pool_live_blocks();
// a block that goes back to the pool when the scope ends (work arrays of a factorisation attempt: an ILUPP_HIP that throws in the
// middle of one would otherwise leak all of them)
struct PoolBlock {
    void *p = nullptr;
    PoolBlock() {}
    PoolBlock(const PoolBlock &) = delete;
    PoolBlock &operator=(const PoolBlock &) = delete;
    ~PoolBlock() { if (p) (void)pool_free(p); }
    hipError_t alloc(size_t bytes) { if (p) { (void)pool_free(p); p = nullptr; } return pool_malloc(&p, bytes); }
    void *release() { void *q = p; p = nullptr; return q; }                  // the caller takes the block over
    void swap(PoolBlock &o) { void *q = p; p = o.p; o.p = q; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
template <class T> inline hipError_t pool_malloc(T **p, size_t bytes) { return pool_malloc(reinterpret_cast<void **>(p), bytes); }

// two events that bracket a kernel for its time; destroyed on every way out of the scope (ILUPP_HIP throws)
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    EventPair() {}
    EventPair(const EventPair &) = delete;
    EventPair &operator=(const EventPair &) = delete;
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    hipError_t create() { hipError_t e = hipEventCreate(&a); return e != hipSuccess ? e : hipEventCreate(&b); }
};

// ---- device containers ----------------------------------------------------------------------
// A compressed sparse matrix resident in HBM.  `is_csr` is only the LABEL handed back to the
// caller (reference: matrix_sparse::orientation); kernels always see "major slices".
struct DevMat {
    int32_t n = 0;
    int64_t nnz = 0;
    int32_t *ptr = nullptr;   // n+1
    int32_t *idx = nullptr;   // nnz
    double *val = nullptr;    // nnz
    bool is_csr = true;
    bool owns = true;
    void release();
};

// Row-block schedule of one sweep direction: lane s of the persistent grid owns rows
// [start[b], start[b+1]) with b = s (forward) or nb-1-s (backward).
struct Schedule {
    int32_t nb = 0;             // number of row blocks
    int32_t B = 1;              // nominal rows per block (start[b] lies in [b*B, (b+1)*B])
    int32_t *start = nullptr;   // nb+1, ascending
    // placement of blocks on the persistent grid (schedule.hip); slot = workgroup*256 + lane
    bool fwd = true;
    int32_t nslots = 0;
    int32_t *slot2blk = nullptr;   // nslots, -1 = idle lane
    int32_t *blk2slot = nullptr;   // nb
    int32_t *sfirst = nullptr;     // nslots: first row of the slot in processing order
    int32_t *scount = nullptr;     // nslots: number of rows of the slot
    int32_t *exported = nullptr;   // nslots: 1 if some OTHER workgroup reads this slot's results (filled by make_desc / the program builder)
    int32_t *gtab = nullptr;       // nslots/256 x kGhosts: foreign producer slots a workgroup imports through ghost lanes (-1 = free)
    // 2-D tiling of the block grid (0 = identity placement): block b = (b % s2, b / s2), a workgroup owns ty x tz blocks
    int32_t tile_s2 = 0, tile_ty = 0, tile_tz = 0;
    bool chains = false;           // every block is exactly one chain of rows and the rows of every chain are alike (symbolic.hip: rows_alike)
    bool chains_pre = false;       // ... as far as the first pass' counts say; `ragged` (k_block_starts' verdict) arrives with the NEXT
    int32_t ragged = 1;            // stream_sync of the construction (ilu0_symbolic_and_schedule queues the read-back, finish_chains folds it in)
    void release();
};

// Descriptor owner fields >= kGhostBase name ghost lane (owner - kGhostBase) of the reading workgroup.
static constexpr int kGhosts = 64;
static constexpr int kGhostBase = (1 << 17) - kGhosts;

// One triangular factor renumbered in level order for the one-lane-per-row sweep (sptrsv_lvl.hip): rows sorted by
// (level, row), columns rewritten to positions, each row = off-diagonal entries in application order + diagonal.
struct LevelSweep {
    bool tried = false, valid = false;
    int32_t n = 0, nlevels = 0;
    int w = 8, block = 1024;                                    // window and block size of the sweep kernel
    int32_t *ptr = nullptr, *idx = nullptr, *perm = nullptr;    // n+1, nnz, n (position -> row)
    double *val = nullptr, *xp = nullptr;                       // nnz; n unknowns in position order (what the rows poll)
    void release();
};

// Level-major packed form of one triangular sweep (sptrsv_lm.hip); short-row factors only.
struct PackedSweep {
    bool built = false;         // structure (skews, chunk table, storage) exists
    bool valid = false;         // records written and verified: the packed kernel may run
    int kind = 0;               // SweepKind the records were ordered for
    int32_t nwg = 0;
    int32_t *skew = nullptr;    // nslots: step offset of each lane
    int32_t *wtab = nullptr;    // nwg*4 x {first chunk, first step, chunks, -}
    int32_t *flags = nullptr;   // [0] rejected, [1] total chunks, [2] longest wave
    void *pk = nullptr;         // chunks x 3072 bytes
    int32_t *uslot = nullptr;   // (backward sweep) slot of this schedule that owns the rows of each forward-schedule slot
    bool linked = false;        // uslot complete: the factor kernel can address this sweep's records
    double *ybuf = nullptr;     // (forward sweep) its unknowns in level-major order, 64 per chunk
    int32_t *ysrc = nullptr;    // (backward sweep) per slot: where its first row's right-hand side sits in the forward sweep's ybuf
    int64_t nchunks = 0;
    int32_t max_chunks = 0;
    // static form (st.hip): every lane's dependency structure is the same for all of its rows and sits in a lane
    // table; the records then hold values only (2 KB per chunk, absent entries = kAbsent)
    bool stat = false;
    bool wx = false;            // every lane of the schedule fits the wave-exchange kernels (st_wave.hip; st_common.h: wx_lane_ok)
    bool vec_ok = false;        // ... and no more than 16 lanes of a workgroup share a skew modulo 16 (st_wave.hip: the vector wave serves two instructions of 8 lanes per step)
    int fmt = 0;                // static records: 0 = by template position, kAbsent where there is no entry; 1 = class-aligned, +0.0 (st_wave.hip)
    int32_t *ltab = nullptr;    // nslots x kStTab ints
    double *dump = nullptr;     // where the stores of lanes without a row go
    double *xlm = nullptr;      // (forward sweep) the right-hand side in this sweep's level-major order, 64 per chunk
    int64_t y_chunks = 0;       // (forward sweep) chunks of the backward sweep, in whose order the intermediate vector is stored
    // exchange between workgroups of a static schedule, step-major: workgroup w's exported lanes (ordinal xe[slot], xw[4w] of
    // them rounded up to 16) store the value of step s at xch[xw[4w+3] + (s - xw[4w+1]) * xw[4w] + xe]
    int32_t *xe = nullptr, *xw = nullptr;
    double *xch = nullptr;
    int64_t xch_len = 0;
    // records of the TRANSPOSED factor for this sweep's schedule (static form; built on the first transposed apply, dropped
    // by a re-factorisation): forward schedule: U^T (diagonal u_rr), backward schedule: L^T (diagonal 1)
    unsigned char *pkT = nullptr;
    int fmtT = 0;               // format of the transposed records (0 / 1 as `fmt`)
    // static sweeps made from a pair of stored triangular factors (st_analyse_pair): the forward sweep divides by its stored
    // diagonal; the backward sweep accumulates in descending column order when `desc` (a row-stored lower factor used transposed)
    bool pair = false, desc = false;
    // (forward sweep of an ILU(0) whose row blocks were guessed, grid.hip) the proof's end on its side stream and its verdict word, ctrl[8]:
    // the factor kernel's own read-back waits for the one and takes the other along
    hipEvent_t join_ev = nullptr;
    bool join_before = false;           // wait for join_ev in front of the factor kernel (behind the launches that prepare it), not behind it
    int32_t join_verdict = 0;
    // (forward sweep of a static ILU(0)) what the factor kernel's launch is followed by, behind its read-back and in front of the wait for
    // it: the arming of the first apply (api.hip: arm_apply) -- `arm` runs with `arm_ctx`, the wait is for `arm_ev` instead of the stream
    void (*arm)(void *) = nullptr;
    void *arm_ctx = nullptr;
    hipEvent_t arm_ev = nullptr;
    mutable bool xch_armed = false;     // the exchange buffer is all-sentinel already (api.hip: arm_apply, behind the previous call's last wait): the sweep need not fill it
    void release();
};

// lane table of the static level-major kernels (st.hip), one entry of kStTab ints per slot
static constexpr int kStTab = 32;
enum { ST_FIRST = 0, ST_CNT = 1, ST_SKEW = 2, ST_ND = 3, ST_OFF = 4, ST_SRC = 7, ST_KLO = 10, ST_KAP = 13, ST_UP0 = 16, ST_DT = 17,
       ST_KHI = 20, ST_SCAT = 23 };
enum { ST_NONE = 0, ST_OWN = 1, ST_LOCAL = 2, ST_GHOST = 3 };     // low two bits of a source word; the producer slot above them

// Hand-off rings of the level-major factor kernel (their addresses are baked into the factor records, records_lm.hip):
// own U rows kFlmUF deep (slot = row index modulo kFlmUF; a wave runs at most kFlmUF-2 steps ahead of the slowest wave
// of its workgroup), ghost U rows kFlmGF deep
static constexpr int kFlmUF = 4;
static constexpr int kFlmGF = 4;
static constexpr int kFlmGhostBase = kFlmUF * 3 * 256;
static_assert(kFlmGhostBase + kFlmGF * 3 * 64 <= 4096, "ring addresses are 12-bit fields of the factor records");

// Level-major factor kernel (ilu0_lm.hip): what it needs beyond the two sweeps' structures
struct FactorLM {
    bool built = false;
    bool values_packed = false;     // the factor records already hold the values of the matrix about to be factored
    void *pkA = nullptr;            // chunks x 5120 bytes: rows of A (diagonal-aligned), program header, decoded dependencies
    int32_t *xbase = nullptr;       // nslots: first exchange row of a slot whose U rows other workgroups read, else -1 (+ the rows pass' records behind it)
    int64_t xbase_len = 0;          // ints of that allocation
    double *xch = nullptr;          // exchange rows x 4 doubles (write-through, sentinel = not yet)
    long long *xcount = nullptr;    // device: doubles of xch in use
    bool stat = false;              // static form (st.hip): pkA holds 4 KB chunks {a0..a6, mask}
    bool direct = false;            // static form fed from A's CSR values (st_direct.hip): no pkA at all
    bool wxf = false;               // ... by the wave-exchange factor kernel (st_wave.hip: k_ilu0_wx), which writes format-1 records
    // (box grids, grid.hip) the sizes the analysis used to wait for were PREDICTED from the dimensions and everything behind them launched
    // at once; what the device found (flags of both schedules, exchange totals) lands here with the construction's last read-back and
    // must equal the prediction, or the construction is redone the waiting way (api.hip: ilu0_factor)
    bool spec = false;
    int32_t chk_hl[12] = {0}, chk_hu[12] = {0}, chk_xtot[4] = {0}, pred[4] = {0};      // pred: chunks, longest wave, exchange offset of the last workgroup, its size
    void release();
};

// ILU(0) update program (schedule.hip)
struct Ilu0Program {
    int32_t *prow = nullptr;    // n+1 word offsets
    int32_t *prog = nullptr;    // nwords
    int64_t nwords = 0;
    int32_t max_words = 0;      // longest record
    int32_t max_ulen = 0;       // longest U row (diagonal included)
    void release();
};

// Sentinel that marks "not yet computed" in solve vectors (data-is-flag hand-off, see sptrsv.hip).
// A quiet NaN with a payload no arithmetic produces (hardware NaNs are canonical 0x7FF8000000000000).
static constexpr unsigned long long kSentinel = 0x7FF85EEDC0DE0001ull;
static constexpr unsigned long long kCanonNaN = 0x7FF8000000000000ull;
// "no such entry" in the value records and hand-off entries of the static level-major kernels (st.hip)
static constexpr unsigned long long kAbsent = 0x7FF85EEDC0DE0002ull;

// persistent-grid geometry: one 256-thread workgroup per CU
static constexpr int kThreads = 256;

// ---- device helpers -------------------------------------------------------------------------
#if defined(__HIPCC__)
// agent-scope relaxed accesses: global_load/store ... sc1 (bypass the per-CU L1, coherent across XCDs)
__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent_f64(const double *p)
{
    return __longlong_as_double((long long)ld_agent_u64(reinterpret_cast<const unsigned long long *>(p)));
}
__device__ __forceinline__ void st_agent_f64(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_u64(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent_i32(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_i32(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a 32-byte record with two 16-byte agent-scope accesses instead of four 8-byte ones (the dataflow kernels of ICholT / ILUC are
// bound by the number of small memory operations; a record is only read after the counter that publishes it)
struct Rec32 { unsigned long long w[4]; };
__device__ __forceinline__ Rec32 ld_agent_rec32(const unsigned long long *p)
{
    typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
    v2u a, b;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    Rec32 r; r.w[0] = a.x; r.w[1] = a.y; r.w[2] = b.x; r.w[3] = b.y;
    return r;
}
__device__ __forceinline__ void st_agent_rec32(unsigned long long *p, unsigned long long w0, unsigned long long w1, unsigned long long w2,
                                               unsigned long long w3)
{
    typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
    v2u a, b;
    a.x = w0; a.y = w1; b.x = w2; b.y = w3;
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" :: "v"(p), "v"(a), "v"(b) : "memory");
}
// block that owns row c:  start[b] <= c < start[b+1]
__device__ __forceinline__ int block_of(int c, int B, int nb, const int32_t *__restrict__ start)
{
    int b = c / B;
    if (b >= nb) b = nb - 1;
    while (c < start[b]) --b;
    while (c >= start[b + 1]) ++b;
    return b;
}

// The (at most 8) column indices of a short row in registers: two 16-byte loads instead of a load per entry.
// Entries beyond `len` are set to INT_MAX so that comparisons need no length checks.
struct __attribute__((aligned(4))) I4a { int v[4]; };
struct Row8 { int c[8]; };
__device__ __forceinline__ Row8 load_row8(const int32_t *__restrict__ idx, int q0, int len, int64_t nnz)
{
    Row8 r;
    if ((int64_t)q0 + 8 <= nnz) {
        const I4a lo = *reinterpret_cast<const I4a *>(idx + q0);
        const I4a hi = *reinterpret_cast<const I4a *>(idx + q0 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { r.c[i] = lo.v[i]; r.c[4 + i] = hi.v[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) r.c[i] = (int64_t)q0 + i < nnz ? idx[q0 + i] : 0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r.c[i] = i < len ? r.c[i] : 0x7fffffff;
    return r;
}
#define ROW8_AT(R, I) ((I) == 0 ? (R).c[0] : (I) == 1 ? (R).c[1] : (I) == 2 ? (R).c[2] : (I) == 3 ? (R).c[3] : \
                       (I) == 4 ? (R).c[4] : (I) == 5 ? (R).c[5] : (I) == 6 ? (R).c[6] : (R).c[7])

// compiler-only ordering point: keeps payload loads below the poll that guards them
__device__ __forceinline__ void order_after_poll() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
// every store of this wave has left the CU before the flag that publishes them
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// ---- kernels' host entry points (one per .hip file) -------------------------------------------

// symbolic.hip
int ilu0_symbolic_and_schedule(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, int32_t *first_missing_diag,
                               int max_lanes, Schedule *fwd, Schedule *bwd, int32_t *max_row_len);
void ilu0_write_patterns(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U);
void finish_chains(Schedule *fwd, Schedule *bwd);
const char *wx_factor_kernel_name();     // st_wave.hip: the factor kernel ilu0_numeric_wx launches, as a profiler names it
bool wx_vec_on();            // st_wave.hip: the sweeps move the caller's vector through their vector wave (no level-major copies, no k_st_vec)
// grid.hip: the first analysis pass for lexicographic box-grid stencil matrices (guess from row 0, proof on a side stream)
struct GridDims { int32_t nx, ny, nz; };
bool grid_guess(int32_t n, int64_t nnz, const int32_t *head /* ptr[0], ptr[1], idx[0..7] */, GridDims *g);
bool grid_shape_recall(int32_t n, int64_t nnz, GridDims *g, const void *idx);
void grid_shape_remember(int32_t n, int64_t nnz, const GridDims &g, const void *idx);
void grid_shape_forget(int32_t n, int64_t nnz, const void *idx);
void grid_check_launch(hipStream_t side, const DevMat &A, const GridDims &g, int32_t *d_bad);
bool grid_llt_schedule(hipStream_t st, int32_t n, const GridDims &g, int max_lanes, Schedule *bwd);
bool llt_grid_rows(hipStream_t st, const DevMat &Lc, const GridDims &g, DevMat *T);      // row-major copy of a column-major factor with the grid's lower pattern
void make_desc_llt_grid(hipStream_t st, const DevMat &M, const Schedule &sch, const GridDims &g, int32_t **desc, bool with_pattern = false);
void grid_schedules(hipStream_t st, const DevMat &A, const GridDims &g, DevMat *L, DevMat *U, Schedule *fwd, Schedule *bwd,
                    int32_t *max_row_len, int max_wgs);
// slot tables, lane templates and the link between the two schedules of a box grid, from its dimensions (one launch; the schedules'
// slot arrays must be allocated: build_slot_tables(..., false))
// (true: skews, hand-off ages and the waves' chunk ranges are made too -- k_st_link_pair's job)
bool grid_lane_tables(hipStream_t st, const GridDims &gd, const Schedule &fwd, const Schedule &bwd, int32_t *ltabF, int32_t *ltabB,
                      int32_t *flagsF, int32_t *flagsB, int32_t *uslot, int32_t *skewF, int32_t *skewB, int32_t *wtabF, int32_t *wtabB, bool wx);
// chunks of a schedule (all waves), chunks of its longest wave, exchange entries before the last workgroup and of the last workgroup, for
// a box grid placed in 16 x 16 patches with the wave-exchange skews (what k_st_link / k_st_scan / k_st_xch_pair find); false: not predictable
bool grid_predict_sizes(const GridDims &g, int ty, int tz, int64_t *nchunks, int32_t *maxch, int64_t *xoff_last, int32_t *xsz_last);
// ... and everything the static analysis makes from them (first chunks of the waves, positions in the other schedule's records, direct-feed
// fields, backward right-hand-side map, exchange layout) for 16 x 16 patches, in closed form: one launch instead of k_st_scan_pair,
// k_st_clear, k_st_inv_ysrc, k_st_scat and k_st_xch_pair
void grid_scat_tables(hipStream_t st, const GridDims &gd, const Schedule &fwd, int32_t *ltabF, int32_t *ltabB, int32_t *wtabF, int32_t *wtabB,
                      int32_t *ysrc, int32_t *rtab, int32_t *xeF, int32_t *xeB, int32_t *xwF, int32_t *xwB, int32_t *flagsF, int32_t *flagsB,
                      int32_t *tot);
// a verdict computed on another stream that a read-back of the analysis takes along: the stream waits for `ev`, then *host = *dev
struct SideJoin { hipEvent_t ev; const int32_t *dev; int32_t *host; bool done; };
int ilu0_csr_ptrs(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U);
int csr_ptrs_from_counts(hipStream_t st, int32_t n, int32_t *counts, DevMat *M);
int st_make_csr(hipStream_t st, int32_t n, const PackedSweep &pl, const PackedSweep &pu, DevMat *L, DevMat *U);
int count_cuts_and_schedule(hipStream_t st, int32_t n, const int32_t *ptr, const int32_t *idx,
                            int max_lanes, Schedule *fwd, Schedule *bwd, int32_t *max_row_len);
void transpose_storage(hipStream_t st, const DevMat &A, DevMat *T, const GridDims *llt_grid = nullptr);
// The order vector_dense<T>::quicksort(index_list&, left, right) (sparse_implementation.h:471-505) leaves keys and their list in: an
// unstable quicksort with the first element as pivot -- where keys are equal, the permutation that comes out IS the reference's answer,
// so the same partition scheme runs here (host; the smaller part by recursion, the larger by the loop: the two parts are disjoint, the
// order in which they are sorted does not matter, and the stack stays logarithmic).
template <class T> void ref_quicksort(T *key, int32_t *list, long left, long right)
{
    while (left < right) {
        const T pivot = key[left];
        long i = left, j = right;
        while (i <= j) {
            while (key[i] < pivot) i++;
            while (key[j] > pivot) j--;
            if (i <= j) { std::swap(key[i], key[j]); std::swap(list[i], list[j]); i++; j--; }
        }
        if (j - left < right - i) { ref_quicksort(key, list, left, j); left = i; }
        else { ref_quicksort(key, list, i, right); right = j; }
    }
}
void mwm_release_stage();   // ml_order.hip: the pinned staging buffer of the matching
// krylov.hip: BiCGstab with SPLIT preconditioning on device vectors (iterative_solvers_implementation.h:385-530 from the zero vector)
int bicgstab_split(hipStream_t st, int32_t n, const std::function<int(const double *, double *)> &op, double *r, double *y, int32_t min_iter,
                   int32_t max_iter, double rtol, double atol, int32_t *it_out, double *rel_out, double *res_out);
void fill_u64(hipStream_t st, unsigned long long *p, int64_t count, unsigned long long v);
int device_cu_count();
int32_t min_row_len(hipStream_t st, int32_t n, const int32_t *ptr, const int32_t *idx, int diag_at);
void iota_i32(hipStream_t st, int32_t *p, int64_t count);
hipError_t d2h_async(hipStream_t st, void *host_dst, const void *dev_src, size_t bytes);   // small read-back, visible after stream_sync()
struct D2HItem { void *dst; const void *src; size_t bytes; };
hipError_t d2h_async_many(hipStream_t st, const D2HItem *items, int n);   // up to 8 read-backs with one launch
hipError_t stream_sync(hipStream_t st);
hipError_t event_sync(hipStream_t st, hipEvent_t ev);      // ... for an event behind all pending read-backs of st (what is queued behind it runs on)
void d2h_cancel_all();

// schedule.hip
void choose_tiling(hipStream_t st, int32_t n, const int32_t *ptr, const int32_t *idx, Schedule *sch, bool fwd, int max_wgs);
void choose_tiling_pair(hipStream_t st, const int32_t *ptr, const int32_t *idx, Schedule *fwd, Schedule *bwd, int max_wgs);
void build_slot_tables(hipStream_t st, Schedule *sch, bool fwd, bool fill = true);      // fill = false: the arrays only (grid.hip fills them)
void make_desc(hipStream_t st, const DevMat &M, const Schedule &sch, int32_t **desc, bool fill_import_table = true);
bool build_ilu0_program(hipStream_t st, const DevMat &A, const DevMat &U, const Schedule &sch, Ilu0Program *P);
bool build_ilu0_program_f3(hipStream_t st, const DevMat &A, const DevMat &U, const Schedule &sch, int32_t **prog_out);
int ilu0_numeric_lc(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const Schedule &fwd, const int32_t *prog_f3,
                    int32_t *d_ctrl, float *kernel_ms, const PackedSweep *pl = nullptr, const PackedSweep *pu = nullptr);

// ilu0.hip
int ilu0_numeric_program(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const Schedule &fwd,
                         const Ilu0Program &P, int32_t max_row_len, int32_t *d_ctrl, float *kernel_ms);
void ilu0_unit_diagonal(hipStream_t st, DevMat *L);
int ilu0_numeric(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const Schedule &fwd,
                 int32_t max_row_len, int32_t *d_done, int32_t *d_ctrl, float *kernel_ms);

// ichol.hip
int triangular_part(hipStream_t st, const DevMat &A, bool lower, DevMat *T, int32_t *first_missing_diag);
int ichol0_numeric(hipStream_t st, DevMat *L, const Schedule &fwd, int32_t max_row_len, int32_t *d_done, int32_t *d_ctrl,
                   float *kernel_ms);

int icholt_factor(hipStream_t st, const DevMat &Atri, int32_t add_fill_in, double threshold, DevMat *L, float *kernel_ms);
// icholt_df.hip (returns 1 when the matrix is outside its capacities: the caller runs the sequential kernel)
// ICholT(0, 0.0) of a box grid as a speculative static computation (icholt_grid.hip): launch queues everything and hands L's index
// arrays over at once (event pattern_done), finish waits and says whether the premise held
struct IcholtGridJob {
    PoolBlock xch;
    EventPair ev;
    hipEvent_t pattern_done = nullptr;
    hipStream_t launched_on = nullptr, side_stream = nullptr;
    bool finished = false;
    bool pattern_written = false;      // the caller's pattern_free work wrote L's index arrays too (make_desc_llt_grid)
    GridDims g = {0, 0, 0};
    int32_t h[16] = {0};
    IcholtGridJob() {}
    IcholtGridJob(const IcholtGridJob &) = delete;
    IcholtGridJob &operator=(const IcholtGridJob &) = delete;
    ~IcholtGridJob();
};
bool icholt_grid_launch(hipStream_t st, hipStream_t side, const DevMat &A, const GridDims &g, int32_t *ctrl, DevMat *L, IcholtGridJob *job,
                        const std::function<void(hipStream_t)> &pattern_free, const std::function<void(hipStream_t)> &after_pattern);
bool icholt_grid_finish(hipStream_t st, IcholtGridJob *job, float *kernel_ms);
int icholt_factor_df(hipStream_t st, const DevMat &Atri, int32_t add_fill_in, double threshold, DevMat *L, float *kernel_ms);
// iluc_df.hip: Crout ILU on the major-order view; L by columns (arrays = CSR of L^T, 1 first), U by rows (pivot first)
int iluc_factor(hipStream_t st, const DevMat &Av, int32_t max_fill_in, double threshold, DevMat *L, DevMat *U, int32_t *err_row,
                float *kernel_ms);

// piluc_df.hip: one level of the multilevel preconditioner without pivoting (reference partialILUC, ILUCDP.hpp:1405-2231)
struct PilucParams {
    double min_pivot = 1e-2;              // MIN_PIVOT
    bool small_pivot_terminates = true;   // SMALL_PIVOT_TERMINATES
    double min_elim_factor = 0.0;         // MIN_ELIM_FACTOR
    double threshold_shift_schur = 0.0;   // THRESHOLD_SHIFT_SCHUR
    int32_t max_fill_in = 0;              // 0: MAX_FILLIN_IS_INF; else fill_in
    int32_t rules = 4;                    // PILUC_DROP_*: which rules make up the weight of a row (default: error propagation)
    int32_t combine = 0;                  // COMBINE_FACTOR
    bool scale_invdiag = false;           // SCALE_WEIGHT_INVDIAG
    double wgt[7] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};   // WEIGHT_STANDARD_DROP, _DROP2, WEIGHT_ERR_PROP_DROP, _DROP2, WEIGHT_PIVOT_DROP, WEIGHT_INVERSE_DROP, WEIGHT_WEIGHTED_DROP
    double init_weights_lu = 1.0;         // INIT_WEIGHTS_LU
    double neutral = 0.0, min_weight = 1.0;      // NEUTRAL_ELEMENT, MIN_WEIGHT
    // the factorisation with pivoting (pilucdp.hip)
    double piv_tol = 0.0;                 // piv_tol
    int32_t permute_rows = 0, total_piv = 0;     // PERMUTE_ROWS, TOTAL_PIV
    bool begin_total_piv = true;          // BEGIN_TOTAL_PIV
    int32_t final_row_crit = -1;          // FINAL_ROW_CRIT
    double move_level_factor = 2.0, row_u_max = 1.5;     // MOVE_LEVEL_FACTOR, ROW_U_MAX
    // preconditioner_implementation.h:1376-1382 (EXTERNAL_FINAL_ROW off)
    bool pivoting() const { return !((permute_rows == 0 || permute_rows == 1) && (!begin_total_piv || total_piv == 0) && piv_tol == 0.0); }
};
enum { PILUC_DROP_STANDARD = 1, PILUC_DROP_STANDARD2 = 2, PILUC_DROP_ERR_PROP = 4, PILUC_DROP_ERR_PROP2 = 8, PILUC_DROP_PIVOT = 16,
       PILUC_DROP_INVERSE = 32, PILUC_DROP_WEIGHTED = 64, PILUC_DROP_WEIGHTED2 = 128 };   // = ILUPP_DROP_*
int piluc_level(hipStream_t st, const DevMat &Av, const PilucParams &P, bool force_finish, double tau, DevMat *L, DevMat *U, double **Dinv, DevMat *Anew,
                int32_t *kterm, float *kernel_ms);

// pilucdp.hip: one level WITH pivoting (reference partialILUCDP, ILUCDP.hpp:268-1404): a sequential algorithm -- every step picks its
// column by the values of the step and its row by the fill so far -- run by one wave; pc2 / pr2 (device, n entries): the column / row
// taken at step k
// pilucdp.hip: the chains of several constructions that run side by side (one host thread and stream each) are launched TOGETHER, one
// workgroup per chain: a thread inside a batch hands its launch to the batch and waits until every other live thread of the batch
// has handed in its own (or has finished)
struct ChainBatch;
ChainBatch *chain_batch_create(int workers);
void chain_batch_destroy(ChainBatch *b);
void chain_batch_enter(ChainBatch *b);       // the calling thread is a worker of b from now on ...
void chain_batch_leave(ChainBatch *b);       // ... until here (it will hand in no more launches)
// pilucdp.hip: partialILUC as a sequential chain with its working vectors in LDS -- for what the dataflow kernel cannot run (dropping rules
// that are recurrences over all steps) or runs badly (long working rows: its largest class).  +1: outside the chain kernel's capacity (LDS: working
// rows of 2 048 entries; with mem_ok the chain goes on with its vectors in global memory: 32 768).
int piluc_chain_level(hipStream_t st, const DevMat &Arow, const PilucParams &P, bool force_finish, double tau, DevMat *L, DevMat *U, double **Dinv_out,
                      DevMat *Anew, int32_t *kterm, float *kernel_ms, bool mem_ok);
int pilucdp_level(hipStream_t st, const DevMat &Arow, const PilucParams &P, bool force_finish, double tau, int32_t bp, int32_t bpr, int32_t epr,
                  DevMat *L, DevMat *U, double **Dinv, DevMat *Anew, int32_t *pc2, int32_t *pr2, float *kernel_ms);

// pilucdp.hip: the entries with |x| > 0 of every segment (compress()), their indices through `map` minus `shift`, every segment sorted by the new index
int seg_compress_sort(hipStream_t st, int32_t nseg, const int32_t *ptr, const int32_t *idx, const double *val, const int32_t *map, int32_t shift,
                      bool is_csr, DevMat *M);
// ilutp.hip: ILUTP2 (ILUTP.hpp:13-140) on the major-order view A (rows): L by rows (1 last, permuted numbering), U by rows in the permuted numbering
// (pivot first) and the same with original column indices, perm (device)
int ilutp_factor(hipStream_t st, const DevMat &A, int32_t max_fill_in, double threshold, double piv_tol, int32_t bp, double mem_factor,
                 DevMat *L, DevMat *Up, DevMat *Uorig, int32_t *perm_out, int32_t *zero_pivots, float *kernel_ms);
// ilucp.hip: ILUCP4 (ILUC.hpp:212-370) on the major-order view C: L by columns, U by rows (pivot first, original column indices), perm (device)
int ilucp_factor(hipStream_t st, const DevMat &C, int32_t max_fill_in, double threshold, double piv_tol, int32_t rp, double mem_factor,
                 DevMat *L, DevMat *U, int32_t *perm_out, int32_t *zero_pivots, float *kernel_ms);

// ml.hip: the multilevel preconditioner built from such levels (reference preconditioner_implementation.h:1350-1665, :433-488)
enum { ML_PRE_NORMALIZE_COLUMNS = 1, ML_PRE_NORMALIZE_ROWS = 2, ML_PRE_PQ_ORDERING = 3, ML_PRE_MAX_WEIGHTED_MATCHING_ORDERING = 4,
       ML_PRE_DD_SYMM_MOVE_CORNER_ORDERING_IM = 5, ML_PRE_UNIT_OR_ZERO_DIAGONAL_SCALING = 6, ML_PRE_SPARSE_FIRST_ORDERING = 7, ML_PRE_SYMM_PQ = 8 };    // = ILUPP_PRE_* of include/ilupp_hip.h
struct MlParams {
    double threshold = 0.0;
    int n_pre = 0;
    int pre[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double pq_threshold = 0.0;            // PQ_THRESHOLD
    int max_levels = 100;                 // MAX_LEVELS
    int32_t min_ml_size = 0;              // MIN_ML_SIZE
    PilucParams pil;
    double vary_threshold_factor = 1.0;   // VARY_THRESHOLD_FACTOR
    bool use_final_threshold = false;     // USE_FINAL_THRESHOLD
    double final_threshold = 0.0;         // FINAL_THRESHOLD
};
struct MlLevelDev {
    int32_t n = 0;
    DevMat L, U;                          // Precond_left (by columns, the 1 first), Precond_right (by rows, the 1 first)
    double *D = nullptr;                  // Precond_middle
    double *Dl = nullptr, *Dr = nullptr;  // D_l, D_r
    int32_t *pr = nullptr, *pc = nullptr, *ipr = nullptr, *ipc = nullptr;   // permutation_rows / _columns and their inverses
    void release();
};
int ml_build(hipStream_t st, const DevMat &A, const MlParams &IP, std::vector<MlLevelDev> *levels, float *kernel_ms);
void ml_scale_perm(hipStream_t st, int32_t n, const double *x, const double *D, const int32_t *perm, double *out);
void ml_take(hipStream_t st, int32_t n, double *w, const double *D, double *x);
void ml_scale(hipStream_t st, int32_t n, const double *x, const double *D, double *out);
void ml_perm_scale(hipStream_t st, int32_t n, const double *w, const int32_t *perm, const double *D, double *x);

// ilut.hip
int ilut_factor(hipStream_t st, const DevMat &A, int32_t max_fill_in, double threshold, DevMat *L, DevMat *U,
                int32_t *err_row, float *kernel_ms);

// ilut_wp.hip (returns 1 when a row fits no capacity class, not even the largest: the matrix is too wide for the memory budget)
// (the U rows are RECORDS while the kernel runs: a row's length word, columns and values lie together -- 124 bytes for a budget of 10: one
// 128-byte line per fetched row instead of three or four; Uri / Urv / Ulen point into that slab, a row apart by si ints / sv doubles / sl ints)
struct UrowLayout { int32_t si, sv, sl; };
int ilut_rows_wp(hipStream_t st, const DevMat &A, int32_t p, double threshold,
                 int32_t *Lri, double *Lrv, int32_t *Llen, int32_t *Uri, double *Urv, int32_t *Ulen, UrowLayout ul, int32_t *ctrl, float *kernel_ms);

// sptrsv.hip
enum SweepKind { SWEEP_FWD_LAST_ASC = 0, SWEEP_BWD_FIRST_ASC = 1, SWEEP_BWD_FIRST_DESC = 2 };
int sptrsv(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, const int32_t *desc,
           int32_t max_row_len, double *rhs_and_reset, double *out, int32_t *d_ticket, int32_t *d_err);

int sptrsv_rows(hipStream_t st, SweepKind kind, const DevMat &M, double *rhs_and_reset, double *out, int32_t *d_ticket, int32_t *d_err);
// sptrsv_small.hip: one workgroup, the unknowns in LDS (the small, dense levels of a multilevel preconditioner); rhs is left as it is
static constexpr int32_t kSmallSweepMax = 4096;      // round 3 (one entry per trip): 20 against 27 ms per apply for a 7-level object of n = 700, no gain from n = 3 000 on; round 4 (up to four entries per trip): see DESIGN 4e
int sptrsv_small(hipStream_t st, SweepKind kind, const DevMat &M, const double *rhs, double *out, int32_t *err);

// sptrsv_lvl.hip
bool lvl_order(hipStream_t st, int mode, int32_t n, int64_t nnz, const int32_t *ptr, const int32_t *idx, const Schedule &sch,
               int32_t **perm_out, int32_t *nlevels);
bool lvl_build(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, LevelSweep *ls, const int32_t *perm_given = nullptr,
               int32_t nlevels_given = 0);
int ilu0_numeric_lvl(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const int32_t *perm, int32_t *d_ctrl, float *kernel_ms);
int sptrsv_lvl(hipStream_t st, const LevelSweep &ls, double *rhs_and_reset, double *out, int32_t *d_ticket, int32_t *d_err);

// sptrsv_lm.hip
bool lm_prepare(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, const int32_t *desc,
                int32_t max_row_len, PackedSweep *ps);
void lm_pack(hipStream_t st, SweepKind kind, const DevMat &M, const Schedule &sch, const int32_t *desc, PackedSweep *ps, int what);
void lm_link_factor(hipStream_t st, const Schedule &fwd, const Schedule &bwd, PackedSweep *ps_u);
bool lm_finish(hipStream_t st, PackedSweep *ps);
int sptrsv_lm(hipStream_t st, const PackedSweep &ps, const Schedule &sch, int32_t n, const double *rhs, double *out,
              int32_t *d_ticket, int32_t *d_err, double *ypk_out = nullptr, const double *ypk_in = nullptr,
              const int32_t *ysrc = nullptr);
void lm_link_y(hipStream_t st, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu);

// st.hip
bool st_analyse_ilu0(hipStream_t st, const DevMat &A, const Schedule &fwd, const Schedule &bwd, PackedSweep *pl,
                     PackedSweep *pu, FactorLM *f, SideJoin *join = nullptr, const GridDims *grid = nullptr);
int ilu0_numeric_st(hipStream_t st, const DevMat &A, const Schedule &fwd, PackedSweep *pl, PackedSweep *pu, FactorLM *f,
                    int32_t *d_ctrl, float *kernel_ms, hipEvent_t e0, hipEvent_t e1);
int sptrsv_st(hipStream_t st, const PackedSweep &ps, const Schedule &sch, int32_t n, const double *rhs, double *out,
              int32_t *d_ticket, int32_t *d_err, double *ypk_out = nullptr, const double *ypk_in = nullptr,
              const int32_t *ysrc = nullptr);
void st_unpack(hipStream_t st, const DevMat &M, const Schedule &sch, const PackedSweep &ps);
bool ichol0_numeric_st(hipStream_t st, DevMat *L, const Schedule &fwd, int32_t *d_ctrl, float *kernel_ms, int *rc_out);
// static sweeps for a pair of stored factors: Lrow = row-major lower, diagonal last; Urow = row-major upper, diagonal first; both
// with at most 3 entries per row besides the diagonal (false: the structure does not fit; pl, pu released)
bool st_analyse_pair(hipStream_t st, int32_t n, const DevMat &Lrow, const DevMat &Urow, const Schedule &fwd, const Schedule &bwd,
                     PackedSweep *pl, PackedSweep *pu, bool bwd_desc);
// static form, transposed apply: records of U^T / L^T from those of L and U (false: the pattern is not symmetric enough)
bool st_build_transposed(hipStream_t st, const Schedule &fwd, int32_t n, const FactorLM &f, PackedSweep *pl, PackedSweep *pu,
                         int64_t offdiagL, int64_t offdiagU);
void st_drop_transposed(PackedSweep *pl, PackedSweep *pu);
void st_vec_to_lm(hipStream_t st, const PackedSweep &ps, const double *nat, double *lm);
void st_vec_from_lm(hipStream_t st, const PackedSweep &ps, double *nat);
// st_wave.hip
void wx_convert_records(hipStream_t st, PackedSweep *pl, PackedSweep *pu, int to_fmt);
void wx_convert_transposed(hipStream_t st, PackedSweep *pl, PackedSweep *pu);      // pkT of both: format 0 -> 1 (U^T forward, L^T backward in descending order)
int sptrsv_wx(hipStream_t st, const PackedSweep &ps, int32_t n, const double *rhs, double *out, int32_t *d_ticket, int32_t *d_err,
              double *ypk_out, const double *ypk_in, const int32_t *ysrc);
int sptrsv_st_T(hipStream_t st, const PackedSweep &ps, int32_t n, const double *rhs, double *out, int32_t *d_ticket, int32_t *d_err,
                double *lml, const int32_t *ysrc);

}  // namespace ilupp
