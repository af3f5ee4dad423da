// ilupp_amd/csrc/ilu0.hip -- ILU(0) for gfx950: symbolic split + persistent dataflow numeric kernel.
//
// Replaces ILU0.hpp:69-106 of the reference (compute_ilu0 :26-66, sparse_vec_update :8-23).
// Arithmetic follows the reference operation for operation (row-wise IKJ, ascending k, merge update
// `u_ij -= l_ik * u_kj` as separate multiply and subtract: the library is built with
// -ffp-contract=off), so factor VALUES are bit-identical to the CPU, not merely within 1e-12.
//
// Layout in HBM (all int32 / fp64):
//   L  CSR, strictly-lower entries of the row in ascending column order, then the unit diagonal LAST
//   U  CSR, the diagonal FIRST, then the strictly-upper entries ascending
// The symbolic pass writes both patterns (and L's unit diagonal) once; the numeric pass streams A's
// values in, keeps the working row in LDS, and writes L/U values once.  Algorithmic traffic per row
// of the 7-point problem: 88 B in, 104 B out.
//
// Numeric kernel = one launch, no per-level launches and no grid barrier:
//   * a lane owns a contiguous block of rows (symbolic.hip) and walks them in order;
//   * row r may use row k<r only after done[k] is set: the U row of k is stored write-through (sc1),
//     the storing wave drains its stores (s_waitcnt vmcnt(0)), then sets done[k] (sc1); consumers
//     poll done[k] with sc1 loads and read the U row with sc1 loads (MI355X has 8 XCDs with private,
//     mutually non-coherent L2s: cdna_hip_programming.md guideline 16, form R1);
//   * lanes never block inside divergent code: every lane retries its pending dependency once per
//     wave iteration, so a lane may wait on another lane of its own wave;
//   * workgroup ids come from an atomic ticket, so every block a lane can wait on belongs to a
//     workgroup that has already started (forward progress without assuming dispatch order).
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ilupp {

// ---------------------------------------------------------------------------------------------
// symbolic
// ---------------------------------------------------------------------------------------------
// the unit diagonal of L (ILU0.hpp:93), for the kernels that write the eliminations only
__global__ void k_unit_diag(int32_t n, const int32_t *__restrict__ Lptr, double *__restrict__ Lval)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) Lval[Lptr[r + 1] - 1] = 1.0;
}
void ilu0_unit_diagonal(hipStream_t st, DevMat *L)
{
    hipLaunchKernelGGL(k_unit_diag, dim3((unsigned)((L->n + 255) / 256)), dim3(256), 0, st, L->n, L->ptr, L->val);
}

// ---------------------------------------------------------------------------------------------
// numeric: persistent dataflow kernel
// ---------------------------------------------------------------------------------------------
static constexpr unsigned kSpinLimit = 1u << 22;
// The limit counts idle trips of a wave during which NO wave of the grid made progress: a wave that progresses bumps ctrl[2] at most once
// per millisecond, a wave that idles looks at it every 8 192 trips (a long chain elsewhere is not a hang).
__device__ __forceinline__ void ilu0_heartbeat(int32_t *ctrl, long long &beat_at)
{
    const long long now = wall_clock64();
    if (now - beat_at > 100000) {
        beat_at = now;
        if ((threadIdx.x & 63) == 0) atomicAdd(&ctrl[2], 1);
    }
}
__device__ __forceinline__ bool ilu0_progress_elsewhere(const int32_t *ctrl, int &beat_seen)
{
    const int f = ld_agent_i32(&ctrl[2]);
    const bool moved = f != beat_seen;
    beat_seen = f;
    return moved;
}

// ctrl[0] = workgroup ticket, ctrl[1] = error word
template <int MAXLEN, bool GLOBAL_W>
__global__ void __launch_bounds__(kThreads)
k_ilu0_numeric(const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval,
               const int32_t *__restrict__ Lptr, double *__restrict__ Lval,
               const int32_t *__restrict__ Uptr, const int32_t *__restrict__ Uidx, double *Uval,
               int32_t nb, const int32_t *__restrict__ bstart, int32_t *done, int32_t *ctrl,
               double *wscratch, int32_t wstride)
{
    extern __shared__ __attribute__((aligned(16))) double wlds[];
    __shared__ unsigned wg_ticket;
    if (threadIdx.x == 0) wg_ticket = (unsigned)atomicAdd(&ctrl[0], 1);
    __syncthreads();
    const int tid = threadIdx.x;
    const int64_t slot = (int64_t)wg_ticket * kThreads + tid;

#define W(q) (GLOBAL_W ? wscratch[(size_t)slot * wstride + (q)] : wlds[(q) * kThreads + tid])

    int r = 0, rend = 0;
    if (slot < nb) { r = bstart[slot]; rend = bstart[slot + 1]; }
    bool active = r < rend;
    bool need_init = true;
    int a0 = 0, len = 0, cl = 0, p = 0;
    unsigned spins = 0;
    long long beat_at = 0;
    int beat_seen = 0;

    for (;;) {
        if (!__any(active)) break;
        bool progressed = false;
        if (active) {
            if (need_init) {
                a0 = Aptr[r];
                len = Aptr[r + 1] - a0;
                cl = Lptr[r + 1] - Lptr[r] - 1;
                for (int q = 0; q < len; ++q) W(q) = Aval[a0 + q];     // U[i,:] = A[i,:]   (ILU0.hpp:36-37)
                p = 0;
                need_init = false;
                progressed = true;
            }
            while (p < cl) {                                           // for k < i in row  (ILU0.hpp:47-62)
                const int k = Aidx[a0 + p];
                if (ld_agent_i32(&done[k]) == 0) break;                // row k not finished: retry next round
                order_after_poll();
                const int u0 = Uptr[k], u1 = Uptr[k + 1];
                const double piv = ld_agent_f64(&Uval[u0]);            // diag_U[k]: first entry of U row k
                const double l_ik = W(p) / piv;                        // ILU0.hpp:52
                int pp = p + 1;
                for (int j = u0 + 1; j < u1; ++j) {                    // sparse_vec_update (ILU0.hpp:8-23)
                    const int m = Uidx[j];
                    while (pp < len && Aidx[a0 + pp] < m) ++pp;
                    if (pp >= len) break;
                    if (Aidx[a0 + pp] == m) {
                        const double u_kj = ld_agent_f64(&Uval[j]);
                        const double prod = l_ik * u_kj;
                        W(pp) = W(pp) - prod;
                        ++pp;
                    }
                }
                W(p) = l_ik;                                           // ILU0.hpp:61
                ++p;
                progressed = true;
            }
            if (p == cl) {
                // split (ILU0.hpp:85-98): L gets the multipliers (unit diagonal written by the symbolic
                // pass), U the rest, diagonal first.  U is published write-through for other CUs.
                const int l0 = Lptr[r];
                for (int q = 0; q < cl; ++q) Lval[l0 + q] = W(q);
                const int ub = Uptr[r];
                for (int q = cl; q < len; ++q) st_agent_f64(&Uval[ub + q - cl], W(q));
                drain_stores();
                st_agent_i32(&done[r], 1);
                ++r;
                need_init = true;
                active = r < rend;
                progressed = true;
            }
        }
        if (__any(progressed)) {
            spins = 0;
            ilu0_heartbeat(ctrl, beat_at);
        } else {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 8191u) == 0u && ilu0_progress_elsewhere(ctrl, beat_seen)) spins = 0;
            if (spins > kSpinLimit) {
                if ((tid & 63) == 0) atomicExch(&ctrl[1], 1);
                break;
            }
        }
    }
#undef W
}


// ---------------------------------------------------------------------------------------------
// numeric, program-driven: the production kernel
// ---------------------------------------------------------------------------------------------
// Same dataflow as above, but
//  * the eliminations of a row come from the precomputed update program (schedule.hip): no merge
//    loop, no index loads, every address known up front;
//  * a finished U row is handed to consumers of the SAME workgroup through an LDS ring
//    (ring slot = kloc mod D, seqlock-style tag), which costs ~100 cycles instead of a trip through
//    the memory fabric; this covers the lane's own previous row and its neighbours in the block grid;
//  * every U value is its own flag for consumers in OTHER workgroups: U.val starts as all-sentinel,
//    a value is stored once with a write-through (sc1) store, readers poll with sc1 loads until it
//    is not the sentinel.  No flag array, no store drain, no fence.
// LDS per workgroup: working rows maxlen*256*8 B, ring D*maxu*256*8 B, tags D*256*4 B.
template <int D>
__global__ void __launch_bounds__(kThreads)
k_ilu0_numeric_prog(const double *__restrict__ Aval, const int32_t *__restrict__ Aptr,
                    const int32_t *__restrict__ Lptr, double *__restrict__ Lval,
                    const int32_t *__restrict__ Uptr, double *Uval,
                    const int32_t *__restrict__ prow, const int32_t *__restrict__ prog,
                    int32_t nslots_used, const int32_t *__restrict__ sfirst, const int32_t *__restrict__ scount,
                    int32_t maxlen, int32_t maxu, int32_t *ctrl)
{
    // all LDS in the dynamic region so its base stays 16-byte aligned (cdna_hip_programming.md G17)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    double *w = lds;                                                  // [maxlen][256]
    volatile double *uring = lds + (size_t)maxlen * kThreads;        // [D][maxu][256]
    volatile int *tag = reinterpret_cast<volatile int *>(lds + (size_t)(maxlen + D * maxu) * kThreads);   // [D][256]
    volatile unsigned *wg_ticket = reinterpret_cast<volatile unsigned *>(tag + D * kThreads);
    if (tid == 0) *wg_ticket = (unsigned)atomicAdd(&ctrl[0], 1);
    for (int s = 0; s < D; ++s) tag[s * kThreads + tid] = -1;
    __syncthreads();
    const unsigned wg = *wg_ticket;
    const int myslot = (int)(wg * kThreads + tid);

    int cnt = 0, r = 0;
    if (myslot < nslots_used) { cnt = scount[myslot]; r = sfirst[myslot]; }
    bool active = cnt > 0;
    int rloc = 0;
    int a0 = 0, l0 = 0, u0 = 0, po = 0;
    if (active) { a0 = Aptr[r]; l0 = Lptr[r]; u0 = Uptr[r]; po = prow[r]; }
    bool need_init = true;
    int len = 0, cl = 0, e = 0, pe = 0;
    unsigned spins = 0;
    long long beat_at = 0;
    int beat_seen = 0;

#define W(q) w[(q) * kThreads + tid]
    for (;;) {
        if (!__any(active)) break;
        bool progressed = false;
        if (active) {
            if (need_init) {
                const int hdr = prog[po];
                len = hdr & 0xffff;
                cl = (int)((unsigned)hdr >> 16);
                for (int q = 0; q < len; ++q) W(q) = Aval[a0 + q];
                e = 0;
                pe = po + 1;
                need_init = false;
                progressed = true;
            }
            while (e < cl) {
                const int piv_pos = prog[pe];
                const int kloc = prog[pe + 1];
                const unsigned meta = (unsigned)prog[pe + 2];
                const int nm = (int)(meta & 255u);
                const unsigned oslot = meta >> 8;
                bool from_memory = true;
                if ((oslot >> 8) == wg) {
                    const int lane = (int)(oslot & 255u);
                    const int s = kloc & (D - 1);
                    const int t1 = tag[s * kThreads + lane];
                    if (t1 == kloc) {
                        const double piv = uring[(size_t)(s * maxu) * kThreads + lane];
                        const double l_ik = W(e) / piv;
                        for (int m = 0; m < nm; ++m) {
                            const unsigned mw = (unsigned)prog[pe + 3 + m];
                            const int off = (int)(mw & 0xffffu), pp = (int)(mw >> 16);
                            const double u_kj = uring[(size_t)(s * maxu + off) * kThreads + lane];
                            const double prod = l_ik * u_kj;
                            W(pp) = W(pp) - prod;
                        }
                        const int t2 = tag[s * kThreads + lane];
                        if (t2 != kloc) { need_init = true; break; }   // ring slot recycled under us: redo the row
                        W(e) = l_ik;
                        from_memory = false;
                    } else if (t1 < kloc) {
                        break;                                           // producer not there yet
                    }                                                    // else: slot already recycled -> memory
                }
                if (from_memory) {
                    const double piv = ld_agent_f64(&Uval[piv_pos]);
                    if ((unsigned long long)__double_as_longlong(piv) == kSentinel) break;
                    bool ok = true;
                    for (int m = 0; m < nm; ++m) {
                        const int off = (int)((unsigned)prog[pe + 3 + m] & 0xffffu);
                        const unsigned long long b = ld_agent_u64(reinterpret_cast<const unsigned long long *>(&Uval[piv_pos + off]));
                        ok = ok && (b != kSentinel);
                    }
                    if (!ok) break;
                    const double l_ik = W(e) / piv;
                    for (int m = 0; m < nm; ++m) {
                        const unsigned mw = (unsigned)prog[pe + 3 + m];
                        const int off = (int)(mw & 0xffffu), pp = (int)(mw >> 16);
                        const double u_kj = ld_agent_f64(&Uval[piv_pos + off]);   // final once not the sentinel
                        const double prod = l_ik * u_kj;
                        W(pp) = W(pp) - prod;
                    }
                    W(e) = l_ik;
                }
                pe += 3 + nm;
                ++e;
                progressed = true;
            }
            if (!need_init && e == cl) {
                for (int q = 0; q < cl; ++q) Lval[l0 + q] = W(q);
                const int s = rloc & (D - 1);
                tag[s * kThreads + tid] = -1;
                for (int q = cl; q < len; ++q) {
                    double v = W(q);
                    if ((unsigned long long)__double_as_longlong(v) == kSentinel) v = __longlong_as_double((long long)kCanonNaN);
                    uring[(size_t)(s * maxu + (q - cl)) * kThreads + tid] = v;
                    st_agent_f64(&Uval[u0 + q - cl], v);
                }
                tag[s * kThreads + tid] = rloc;
                a0 += len;
                l0 += cl + 1;
                u0 += len - cl;
                po = pe;
                ++rloc;
                ++r;
                need_init = true;
                active = rloc < cnt;
                progressed = true;
            } else if (need_init) {
                progressed = true;   // row restart counts as activity
            }
        }
        if (__any(progressed)) {
            spins = 0;
            ilu0_heartbeat(ctrl, beat_at);
        } else {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 8191u) == 0u && ilu0_progress_elsewhere(ctrl, beat_seen)) spins = 0;
            if (spins > kSpinLimit) {
                if ((tid & 63) == 0) atomicExch(&ctrl[1], 1);
                break;
            }
        }
    }
#undef W
}

int ilu0_numeric_program(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const Schedule &fwd,
                         const Ilu0Program &P, int32_t max_row_len, int32_t *d_ctrl, float *kernel_ms)
{
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    fill_u64(st, reinterpret_cast<unsigned long long *>(U->val), U->nnz, kSentinel);
    const unsigned grid = (unsigned)(fwd.nslots / kThreads);
    int maxlen = max_row_len < 1 ? 1 : max_row_len;
    int maxu = P.max_ulen < 1 ? 1 : P.max_ulen;
    // ring depth: as deep as LDS allows (a consumer that finds its slot recycled falls back to memory)
    const size_t budget = 144 * 1024;
    int D = 4;
    auto need = [&](int d) { return (size_t)(maxlen + d * maxu) * kThreads * sizeof(double) + (size_t)d * kThreads * sizeof(int) + 16; };
    while (D > 1 && need(D) > budget) D >>= 1;
    if (need(D) > budget) return ILUPP_ERR_UNSUPPORTED;
    const size_t ldsb = need(D);
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    ILUPP_HIP(hipEventRecord(e0, st));
#define LAUNCHP(DD)                                                                                              \
    do {                                                                                                         \
        ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_numeric_prog<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); \
        hipLaunchKernelGGL((k_ilu0_numeric_prog<DD>), dim3(grid), dim3(kThreads), ldsb, st, A.val, A.ptr, L->ptr, L->val, \
                           U->ptr, U->val, P.prow, P.prog, fwd.nslots, fwd.sfirst, fwd.scount, maxlen, maxu, d_ctrl); \
    } while (0)
    if (D == 4) LAUNCHP(4); else if (D == 2) LAUNCHP(2); else LAUNCHP(1);
#undef LAUNCHP
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4];
    ILUPP_HIP(hipMemcpyAsync(ctrl, d_ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    ILUPP_HIP(hipEventDestroy(e0));
    ILUPP_HIP(hipEventDestroy(e1));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}


// ---------------------------------------------------------------------------------------------
// numeric, loader/consumer: the production kernel for short-row matrices (F3 program)
// ---------------------------------------------------------------------------------------------
// Same roles as the solve kernel (sptrsv.hip): waves 4-7 stream each consumer lane's fixed-size update
// records and A values into per-lane LDS rings a few rows ahead; waves 0-3 read LDS only and need two
// dependent LDS round trips per row (round 1: hand-shake words + record + A values; round 2: the ring
// entries of every pivot and matched U value the row needs).  A finished U row is published as one
// 16-byte {tag,value} LDS entry per value (each entry validates itself: no seqlock) and stored
// write-through to HBM, where consumers of other workgroups poll it (data-is-flag on U.val).
struct __attribute__((aligned(8))) D2f { double v[2]; };
struct __attribute__((aligned(16))) I4f { int v[4]; };
typedef int v4i_f __attribute__((ext_vector_type(4)));

static constexpr int kPR = 6, kPQ = 2, kPNQ = 2;   // program ring (8-word records; not a power of two) / quantum / quanta per round
static constexpr int kAW = 32, kAQ = 8, kANQ = 3;  // A-value ring / quantum / quanta per loader round
static constexpr int kUD = 2;               // depth of the U-row hand-off ring (rows per lane)
static constexpr size_t kIluLcLds = (size_t)kThreads * (kPR * 8 * 4 + kAW * 8 + kUD * 4 * 16 + 5 * 4) + 16;

// the working row lives in seven NAMED scalars w0..w6 (an indexable aggregate gets demoted to scratch memory),
// diagonal-aligned: w3 is the diagonal, w0..w2 the (right-aligned) eliminations, w4..w6 the upper entries
#define SEL7(i) ((i) == 0 ? w0 : (i) == 1 ? w1 : (i) == 2 ? w2 : (i) == 3 ? w3 : (i) == 4 ? w4 : (i) == 5 ? w5 : w6)
#define PUT7(i, nv)                                                                                         \
    do {                                                                                                    \
        const int i_ = (i); const double nv_ = (nv);                                                        \
        w0 = i_ == 0 ? nv_ : w0; w1 = i_ == 1 ? nv_ : w1; w2 = i_ == 2 ? nv_ : w2; w3 = i_ == 3 ? nv_ : w3; \
        w4 = i_ == 4 ? nv_ : w4; w5 = i_ == 5 ? nv_ : w5; w6 = i_ == 6 ? nv_ : w6;                          \
    } while (0)

typedef double v2d_f __attribute__((ext_vector_type(2)));
struct PackedOut {                    // where the factor kernel drops the values of the level-major sweep records (null: nowhere)
    v2d_f *pkL; const int32_t *wtabL; const int32_t *skewL;
    v2d_f *pkU; const int32_t *wtabU; const int32_t *skewU; const int32_t *uslot;
};

__global__ void __launch_bounds__(2 * kThreads)
k_ilu0_numeric_lc(const double *__restrict__ Aval, long nnzA, const int32_t *__restrict__ Aptr,
                  const int32_t *__restrict__ Lptr, double *__restrict__ Lval,
                  const int32_t *__restrict__ Uptr, double *Uval, long nnzU,
                  const int32_t *__restrict__ prog, int32_t n,
                  int32_t nslots_used, const int32_t *__restrict__ sfirst, const int32_t *__restrict__ scount,
                  const int32_t *__restrict__ exported, int32_t *ctrl, PackedOut po)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x & (kThreads - 1);
    const bool is_loader = threadIdx.x >= kThreads;
    double *sA = reinterpret_cast<double *>(smem);                                 // [kAW][256]
    v4i_f *ur = reinterpret_cast<v4i_f *>(sA + kAW * kThreads);                    // [kUD][4][256] {tag,-,lo,hi}
    int *sprog = reinterpret_cast<int *>(ur + kUD * 4 * kThreads);                 // [kPR*8][256]
    int *p_avail = sprog + kPR * 8 * kThreads;                                     // [256] each
    int *a_avail = p_avail + kThreads;
    int *p_cons = a_avail + kThreads;
    int *a_cons = p_cons + kThreads;
    int *fin = a_cons + kThreads;
    unsigned *wg_ticket = reinterpret_cast<unsigned *>(fin + kThreads);
    if (threadIdx.x == 0) *wg_ticket = (unsigned)atomicAdd(&ctrl[0], 1);
    __syncthreads();
    const unsigned wg = *wg_ticket;
    const unsigned myslot = wg * kThreads + tid;

#define PW(row, k) sprog[((((row) % kPR) * 8) + (k)) * kThreads + tid]
#define RA(i) sA[((i) & (kAW - 1)) * kThreads + tid]

    int cnt = 0, r0 = 0;
    bool exports = true;          // U rows read by another workgroup must leave the XCD (write-through); the rest may stay in L2
    if ((int)myslot < nslots_used) { cnt = scount[myslot]; r0 = sfirst[myslot]; exports = exported[myslot] != 0; }
    int a00 = 0, l00 = 0, u00 = 0;
    if (cnt > 0) { a00 = Aptr[r0]; l00 = Lptr[r0]; u00 = Uptr[r0]; }
    if (!is_loader) {
#pragma unroll
        for (int s = 0; s < kUD * 4; ++s) { v4i_f e; e.x = -1; e.y = 0; e.z = 0; e.w = 0; ur[s * kThreads + tid] = e; }
        p_avail[tid] = r0; a_avail[tid] = a00;
        p_cons[tid] = r0;  a_cons[tid] = a00;
        fin[tid] = cnt > 0 ? 0 : 1;
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    if (is_loader) {
        // ------------------------------------------------------------------ loader
        // every round: issue ALL quanta the rings have room for, one wait, drop them into LDS, publish
        int p_next = r0, a_next = a00;
        bool live = cnt > 0;
        unsigned idle = 0;
        for (;;) {
            if (!__any(live)) break;
            asm volatile("" ::: "memory");
            bool did = false;
            if (live) {
                if (fin[tid]) {
                    live = false;
                } else {
                    const int pc = p_cons[tid], ac = a_cons[tid];
                    bool tp[kPNQ], ta[kANQ];
                    I4f vp[kPNQ][kPQ * 2];
                    D2f va[kANQ][kAQ / 2];
                    long ba[kANQ][kAQ / 2];
#pragma unroll
                    for (int u = 0; u < kPNQ; ++u) {
                        const int pn = p_next + u * kPQ;
                        tp[u] = (pn + kPQ <= pc + kPR);
                        if (tp[u]) {
#pragma unroll
                            for (int q = 0; q < kPQ; ++q) {
                                const long row = (pn + q < n) ? (pn + q) : (n - 1);
                                vp[u][2 * q] = *reinterpret_cast<const I4f *>(prog + row * 8);
                                vp[u][2 * q + 1] = *reinterpret_cast<const I4f *>(prog + row * 8 + 4);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kANQ; ++u) {
                        const int an = a_next + u * kAQ;
                        ta[u] = (an + kAQ <= ac + kAW);
                        if (ta[u]) {
#pragma unroll
                            for (int q = 0; q < kAQ / 2; ++q) {
                                long b = (long)an + 2 * q;
                                b = b < 0 ? 0 : (b > nnzA - 2 ? nnzA - 2 : b);
                                ba[u][q] = b;
                                va[u][q] = *reinterpret_cast<const D2f *>(Aval + b);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kPNQ; ++u) {
                        if (tp[u]) {
#pragma unroll
                            for (int q = 0; q < kPQ; ++q)
#pragma unroll
                                for (int k = 0; k < 8; ++k) PW(p_next + q, k) = vp[u][2 * q + (k >> 2)].v[k & 3];
                            p_next += kPQ;
                            did = true;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kANQ; ++u) {
                        if (ta[u]) {
#pragma unroll
                            for (int q = 0; q < kAQ / 2; ++q)
#pragma unroll
                                for (int k = 0; k < 2; ++k) RA((int)ba[u][q] + k) = va[u][q].v[k];
                            a_next += kAQ;
                            did = true;
                        }
                    }
                    asm volatile("" ::: "memory");
                    p_avail[tid] = p_next;
                    a_avail[tid] = a_next;
                }
            }
            if (__any(did)) {
                idle = 0;
            } else {
                __builtin_amdgcn_s_sleep(2);
                if (++idle > kSpinLimit) break;
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    int r = r0, rloc = 0;
    int a0 = a00, l0 = l00, u0 = u00;
    bool active = cnt > 0;
    // level-major records of the two sweeps (sptrsv_lm.hip): this lane's rows sit 192 sixteen-byte units apart,
    // ascending in the forward sweep's storage and descending in the backward sweep's
    long lp = 0, up = 0;
    if (po.pkL && cnt > 0) {
        const int w = (int)(myslot >> 6);
        lp = ((long)po.wtabL[w * 4] + (po.skewL[myslot] - po.wtabL[w * 4 + 1])) * 192 + 64 + (tid & 63);
        const int su = po.uslot[myslot];
        const int wu = su >> 6;
        up = ((long)po.wtabU[wu * 4] + (cnt - 1 + po.skewU[su] - po.wtabU[wu * 4 + 1])) * 192 + 64 + (su & 63);
    }
    int phase = 0;
    int pw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pw[k] = 0;
    int len = 0, cl = 0, nmt = 0;
    unsigned spins = 0;
    long long beat_at = 0;
    int beat_seen = 0;

    for (;;) {
        if (!__any(active)) break;
        asm volatile("" ::: "memory");
        bool progressed = false;
        if (active) {
            if (phase == 0) {
                // round 1: hand-shake words + the row's record
                const int pa = p_avail[tid], aa = a_avail[tid];
                asm volatile("" ::: "memory");      // hand-shake words are read BEFORE the data they guard
#pragma unroll
                for (int k = 0; k < 8; ++k) pw[k] = PW(r, k);
                asm volatile("" ::: "memory");      // ... and the data BEFORE the words that release its slots
                if (r < pa) {
                    len = pw[0] & 15; cl = (pw[0] >> 4) & 3; nmt = (pw[0] >> 6) & 7;
                    p_cons[tid] = r;
                    a_cons[tid] = a0;
                    if (a0 + len <= aa) { phase = 1; progressed = true; }
                }
            }
            if (phase == 1) {
                // round 2: the aligned working row and every pivot / matched value the row needs
                const int ab = a0 + cl - 3;
                double w0 = RA(ab), w1 = RA(ab + 1), w2 = RA(ab + 2), w3 = RA(ab + 3), w4 = RA(ab + 4), w5 = RA(ab + 5), w6 = RA(ab + 6);
                int dlane[3], dkl[3]; bool dinwg[3];
                v4i_f pe[3], me[5];
#pragma unroll
                for (int sl = 0; sl < 3; ++sl) {
                    const unsigned kd = (unsigned)pw[2 + 2 * sl];
                    const unsigned oslot = kd >> 15;
                    dkl[sl] = (int)(kd & 0x7fffu);
                    dlane[sl] = (int)(oslot & 255u);
                    dinwg[sl] = (oslot >> 8) == wg;
                    pe[sl] = ur[((dkl[sl] & (kUD - 1)) * 4) * kThreads + dlane[sl]];
                }
                int me_s[5], me_off[5], me_pp[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    const unsigned mw = m < 3 ? ((unsigned)pw[0] >> (9 + 7 * m)) & 127u : ((unsigned)pw[1] >> (7 * (m - 3))) & 127u;
                    me_s[m] = (int)(mw & 3u); me_off[m] = (int)((mw >> 2) & 3u); me_pp[m] = (int)((mw >> 4) & 7u);
                    const int ln = me_s[m] == 0 ? dlane[0] : (me_s[m] == 1 ? dlane[1] : dlane[2]);
                    const int kl = me_s[m] == 0 ? dkl[0] : (me_s[m] == 1 ? dkl[1] : dkl[2]);
                    me[m] = ur[((kl & (kUD - 1)) * 4 + me_off[m]) * kThreads + ln];
                }
                const int s0 = 3 - cl;            // first used dep slot
                bool ready = true, use_mem = false;
                double piv[3] = {1.0, 1.0, 1.0}, um[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                bool dmem[3] = {false, false, false};
#pragma unroll
                for (int sl = 0; sl < 3; ++sl) {
                    if (sl >= s0) {
                        if (dinwg[sl] && pe[sl].x <= dkl[sl]) {
                            if (pe[sl].x == dkl[sl]) piv[sl] = __hiloint2double(pe[sl].w, pe[sl].z);
                            else ready = false;                       // producer not there yet
                        } else { dmem[sl] = true; use_mem = true; }     // other workgroup, or ring slot recycled
                    }
                }
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    if (m < nmt) {
                        const int sl = me_s[m];
                        const bool mm = sl == 0 ? dmem[0] : (sl == 1 ? dmem[1] : dmem[2]);
                        const int kl = sl == 0 ? dkl[0] : (sl == 1 ? dkl[1] : dkl[2]);
                        if (!mm) {
                            if (me[m].x == kl) um[m] = __hiloint2double(me[m].w, me[m].z);
                            else if (me[m].x < kl) ready = false;
                            else { use_mem = true; }                    // recycled between the two reads: take it from memory
                        }
                    }
                }
                if (ready && use_mem) {
                    // values that live only in HBM: write-through stored by their producer, each its own flag
                    unsigned long long bp[3] = {0, 0, 0}, bm[5] = {0, 0, 0, 0, 0};
                    bool needm[5];
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl)
                        if (sl >= s0 && dmem[sl]) bp[sl] = ld_agent_u64(reinterpret_cast<const unsigned long long *>(Uval + pw[3 + 2 * sl]));
#pragma unroll
                    for (int m = 0; m < 5; ++m) {
                        const int sl = me_s[m];
                        const bool mm = sl == 0 ? dmem[0] : (sl == 1 ? dmem[1] : dmem[2]);
                        const int kl = sl == 0 ? dkl[0] : (sl == 1 ? dkl[1] : dkl[2]);
                        needm[m] = (m < nmt) && (mm || me[m].x > kl);
                        const int pv = sl == 0 ? pw[3] : (sl == 1 ? pw[5] : pw[7]);
                        if (needm[m]) bm[m] = ld_agent_u64(reinterpret_cast<const unsigned long long *>(Uval + pv + me_off[m]));
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl)
                        if (sl >= s0 && dmem[sl]) { if (bp[sl] == kSentinel) ready = false; else piv[sl] = __longlong_as_double((long long)bp[sl]); }
#pragma unroll
                    for (int m = 0; m < 5; ++m)
                        if (needm[m]) { if (bm[m] == kSentinel) ready = false; else um[m] = __longlong_as_double((long long)bm[m]); }
                }
                if (ready) {
                    // eliminations in ascending k (= ascending slot), matches of each in ascending column
                    // (reference merge order, ILU0.hpp:8-23, :47-62); every position is a fixed register
                    if ((pw[0] >> 30) & 1) {
                        // simple row: match m belongs to slot s0+m and lands on the diagonal
                        const double u0v = s0 == 0 ? um[0] : 0.0;
                        const double u1v = s0 == 0 ? um[1] : (s0 == 1 ? um[0] : 0.0);
                        const double u2v = s0 == 0 ? um[2] : (s0 == 1 ? um[1] : (s0 == 2 ? um[0] : 0.0));
                        if (0 >= s0) { const double l = w0 / piv[0]; const double pr = l * u0v; w3 = w3 - pr; w0 = l; }
                        if (1 >= s0) { const double l = w1 / piv[1]; const double pr = l * u1v; w3 = w3 - pr; w1 = l; }
                        if (2 >= s0) { const double l = w2 / piv[2]; const double pr = l * u2v; w3 = w3 - pr; w2 = l; }
                    } else {
#define ELIM(SL, WS)                                                                                       \
                    if (SL >= s0) {                                                                         \
                        const double l_ik = WS / piv[SL];                                                   \
                        _Pragma("unroll") for (int m = 0; m < 5; ++m) {                                     \
                            if (m < nmt && me_s[m] == SL) {                                                 \
                                const double prod = l_ik * um[m];                                           \
                                if (me_pp[m] == 3) { w3 = w3 - prod; }                                      \
                                else { const double nv = SEL7(me_pp[m]) - prod; PUT7(me_pp[m], nv); }       \
                            }                                                                               \
                        }                                                                                   \
                        WS = l_ik;                                                                          \
                    }
                    ELIM(0, w0)
                    ELIM(1, w1)
                    ELIM(2, w2)
#undef ELIM
                    }
                    if (0 >= s0) Lval[l0 + 0 - s0] = w0;
                    if (1 >= s0) Lval[l0 + 1 - s0] = w1;
                    if (2 >= s0) Lval[l0 + 2 - s0] = w2;
                    const int ulen = len - cl;
                    // never store the sentinel's bit pattern
                    if ((unsigned long long)__double_as_longlong(w3) == kSentinel) w3 = __longlong_as_double((long long)kCanonNaN);
                    if ((unsigned long long)__double_as_longlong(w4) == kSentinel) w4 = __longlong_as_double((long long)kCanonNaN);
                    if ((unsigned long long)__double_as_longlong(w5) == kSentinel) w5 = __longlong_as_double((long long)kCanonNaN);
                    if ((unsigned long long)__double_as_longlong(w6) == kSentinel) w6 = __longlong_as_double((long long)kCanonNaN);
                    if (po.pkL) {
                        // L record: eliminations in stored order, unit diagonal; U record: strictly-upper entries, then the pivot
                        v2d_f la, lb, ua, ub;
                        la.x = s0 == 0 ? w0 : (s0 == 1 ? w1 : w2); la.y = s0 == 0 ? w1 : w2;
                        lb.x = w2; lb.y = 1.0;
                        ua.x = w4; ua.y = w5; ub.x = w6; ub.y = w3;
                        po.pkL[lp] = la; po.pkL[lp + 64] = lb;
                        po.pkU[up] = ua; po.pkU[up + 64] = ub;
                        lp += 192; up -= 192;
                    }
                    const int sr = rloc & (kUD - 1);
#define PUBLISH(Q, WQ)                                                                                     \
                    if (Q < ulen) {                                                                         \
                        const double v = WQ;                                                                \
                        v4i_f e; e.x = rloc; e.y = 0; e.z = __double2loint(v); e.w = __double2hiint(v);      \
                        ur[(sr * 4 + Q) * kThreads + tid] = e;                                              \
                        if (exports) st_agent_f64(&Uval[u0 + Q], v); else Uval[u0 + Q] = v;                 \
                    }
                    PUBLISH(0, w3)
                    PUBLISH(1, w4)
                    PUBLISH(2, w5)
                    PUBLISH(3, w6)
#undef PUBLISH
                    asm volatile("" ::: "memory");
                    a0 += len; l0 += cl + 1; u0 += ulen;
                    ++r; ++rloc;
                    phase = 0;
                    active = rloc < cnt;
                    if (!active) fin[tid] = 1;
                    progressed = true;
                }
            }
        }
        if (__any(progressed)) {
            spins = 0;
            ilu0_heartbeat(ctrl, beat_at);
        } else {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 8191u) == 0u && ilu0_progress_elsewhere(ctrl, beat_seen)) spins = 0;
            if (spins > kSpinLimit) {
                if ((tid & 63) == 0) atomicExch(&ctrl[1], 1);
                fin[tid] = 1;
                break;
            }
        }
    }
#undef PW
#undef RA
#undef SEL7
#undef PUT7
}

int ilu0_numeric_lc(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const Schedule &fwd, const int32_t *prog_f3,
                    int32_t *d_ctrl, float *kernel_ms, const PackedSweep *pl, const PackedSweep *pu)
{
    PackedOut po = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (pl && pu && pl->built && pu->built && pu->uslot) {
        po.pkL = reinterpret_cast<v2d_f *>(pl->pk); po.wtabL = pl->wtab; po.skewL = pl->skew;
        po.pkU = reinterpret_cast<v2d_f *>(pu->pk); po.wtabU = pu->wtab; po.skewU = pu->skew; po.uslot = pu->uslot;
    }
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    fill_u64(st, reinterpret_cast<unsigned long long *>(U->val), U->nnz, kSentinel);
    const unsigned grid = (unsigned)(fwd.nslots / kThreads);
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    ILUPP_HIP(hipEventRecord(e0, st));
    ILUPP_HIP(hipFuncSetAttribute((const void *)k_ilu0_numeric_lc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kIluLcLds));
    hipLaunchKernelGGL(k_ilu0_numeric_lc, dim3(grid), dim3(2 * kThreads), kIluLcLds, st, A.val, (long)A.nnz, A.ptr,
                       L->ptr, L->val, U->ptr, U->val, (long)U->nnz, prog_f3, A.n, fwd.nslots, fwd.sfirst, fwd.scount, fwd.exported, d_ctrl, po);
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4];
    ILUPP_HIP(hipMemcpyAsync(ctrl, d_ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    ILUPP_HIP(hipEventDestroy(e0));
    ILUPP_HIP(hipEventDestroy(e1));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

int ilu0_numeric(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, const Schedule &fwd,
                 int32_t max_row_len, int32_t *d_done, int32_t *d_ctrl, float *kernel_ms)
{
    const int32_t n = A.n;
    ILUPP_HIP(hipMemsetAsync(d_done, 0, sizeof(int32_t) * (size_t)n, st));
    ILUPP_HIP(hipMemsetAsync(d_ctrl, 0, 16, st));
    const unsigned grid = (unsigned)((fwd.nb + kThreads - 1) / kThreads);
    hipEvent_t e0, e1;
    ILUPP_HIP(hipEventCreate(&e0));
    ILUPP_HIP(hipEventCreate(&e1));
    double *wscratch = nullptr;
    ILUPP_HIP(hipEventRecord(e0, st));
#define LAUNCH(ML, GW, LDSB)                                                                              \
    hipLaunchKernelGGL((k_ilu0_numeric<ML, GW>), dim3(grid), dim3(kThreads), (LDSB), st, A.ptr, A.idx, A.val, \
                       L->ptr, L->val, U->ptr, U->idx, U->val, fwd.nb, fwd.start, d_done, d_ctrl, wscratch, max_row_len)
    if (max_row_len <= 8) LAUNCH(8, false, 8 * kThreads * sizeof(double));
    else if (max_row_len <= 16) LAUNCH(16, false, 16 * kThreads * sizeof(double));
    else if (max_row_len <= 32) LAUNCH(32, false, 32 * kThreads * sizeof(double));
    else if (max_row_len <= 64) LAUNCH(64, false, 64 * kThreads * sizeof(double));
    else {
        ILUPP_HIP(pool_malloc(&wscratch, sizeof(double) * (size_t)grid * kThreads * (size_t)max_row_len));
        LAUNCH(1, true, 0);
    }
#undef LAUNCH
    ILUPP_HIP(hipEventRecord(e1, st));
    ILUPP_HIP(hipGetLastError());
    int32_t ctrl[4];
    ILUPP_HIP(hipMemcpyAsync(ctrl, d_ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (kernel_ms) ILUPP_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    ILUPP_HIP(hipEventDestroy(e0));
    ILUPP_HIP(hipEventDestroy(e1));
    if (wscratch) ILUPP_HIP(pool_free(wscratch));
    if (ctrl[1] != 0) return ILUPP_ERR_TIMEOUT;
    return ILUPP_OK;
}

}  // namespace ilupp
