// ilupp_amd/csrc/symbolic.hip -- pattern analysis shared by all sweeps (gfx950).
//
//  * row-block schedule: the persistent sweeps (ilu0.hip, sptrsv.hip) give every lane a contiguous
//    block of rows that it processes strictly in order, so a dependency on the lane's own previous
//    row never leaves the lane.  Blocks are cut only where row r does NOT depend on row r-1
//    (forward) / row r-1 does NOT depend on row r (backward): a chain r-1 -> r -> r+1 ... is serial
//    work anyway, and cutting inside it would serialise two lanes.  On a lexicographic grid the
//    chains are the x-lines.
//  * transposed storage for the scatter-form solves of the reference
//    (sparse_implementation.h:4055-4065, :4075-4084), built with a stable radix sort.
//
// Everything here is streaming integer work: coalesced loads, HBM-bound, no MFMA.
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ilupp {

// ---------------------------------------------------------------------------------------------
// small device-to-host read-backs (flags, counts) without a copy command: a one-block kernel writes the words into a
// pinned, device-mapped staging buffer of the stream; stream_sync() hands them to their destinations.  A
// hipMemcpyAsync to pageable memory costs 40-50 us of idle stream per read-back on this stack (copy command + staging),
// and the ILU(0) analysis has half a dozen of them.
// ---------------------------------------------------------------------------------------------
namespace {
struct Staging { char *host = nullptr; char *dev = nullptr; size_t used = 0; struct Item { void *dst; size_t off, bytes; }; std::vector<Item> pending; };
std::mutex g_stage_mu;
std::unordered_map<hipStream_t, Staging> g_stage;
constexpr size_t kStageBytes = 262144;        // (the two block samples of an ILU(0) analysis are 64 KiB each: beyond the stage a read-back is a
                                              //  hipMemcpyAsync to pageable memory, which leaves the stream idle for 40 us)
}
__global__ void k_copy_words(const int *__restrict__ src, int *__restrict__ dst, int nwords)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwords) dst[i] = src[i];
}
hipError_t d2h_async(hipStream_t st, void *host_dst, const void *dev_src, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_stage_mu);
    Staging &s = g_stage[st];
    if (!s.host) {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, kStageBytes, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            s.host = static_cast<char *>(h); s.dev = static_cast<char *>(d);
        } else {
            (void)hipGetLastError();
        }
    }
    if (!s.host || (bytes & 3) != 0 || (reinterpret_cast<uintptr_t>(dev_src) & 3) != 0 || s.used + bytes > kStageBytes)
        return hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, st);
    const int nwords = (int)(bytes / 4);
    hipLaunchKernelGGL(k_copy_words, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, static_cast<const int *>(dev_src),
                       reinterpret_cast<int *>(s.dev + s.used), nwords);
    s.pending.push_back({host_dst, s.used, bytes});
    s.used += (bytes + 15) & ~(size_t)15;
    return hipGetLastError();
}
// several read-backs with ONE launch (a launch costs about 4 us of stream time; an analysis has a dozen read-backs)
struct CopyList { const int *src[8]; int off[8]; int words[8]; int n; };
__global__ void k_copy_words_many(CopyList L, int *__restrict__ dst)
{
    for (int e = 0; e < L.n; ++e)
        for (int i = threadIdx.x; i < L.words[e]; i += blockDim.x) dst[L.off[e] + i] = L.src[e][i];
}
hipError_t d2h_async_many(hipStream_t st, const D2HItem *items, int n)
{
    bool fits = n <= 8;
    {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        Staging &s = g_stage[st];
        size_t need = 0;
        for (int e = 0; e < n && fits; ++e) {
            if ((items[e].bytes & 3) != 0 || (reinterpret_cast<uintptr_t>(items[e].src) & 3) != 0) fits = false;
            need += (items[e].bytes + 15) & ~(size_t)15;
        }
        if (!s.host || s.used + need > kStageBytes) fits = false;
        if (fits) {
            CopyList L;
            L.n = n;
            for (int e = 0; e < n; ++e) {
                L.src[e] = static_cast<const int *>(items[e].src); L.off[e] = (int)(s.used / 4); L.words[e] = (int)(items[e].bytes / 4);
                s.pending.push_back({items[e].dst, s.used, items[e].bytes});
                s.used += (items[e].bytes + 15) & ~(size_t)15;
            }
            hipLaunchKernelGGL(k_copy_words_many, dim3(1), dim3(64), 0, st, L, reinterpret_cast<int *>(s.dev));
            return hipGetLastError();
        }
    }
    for (int e = 0; e < n; ++e) {
        const hipError_t err = d2h_async(st, items[e].dst, items[e].src, items[e].bytes);
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}
// after a failure: destinations of read-backs still pending may be gone (stack variables of the frames that threw)
void d2h_cancel_all()
{
    std::lock_guard<std::mutex> lk(g_stage_mu);
    for (auto &kv : g_stage) { kv.second.pending.clear(); kv.second.used = 0; }
}
hipError_t stream_sync(hipStream_t st)
{
    const hipError_t e = hipStreamSynchronize(st);
    std::lock_guard<std::mutex> lk(g_stage_mu);
    auto it = g_stage.find(st);
    if (it != g_stage.end()) {
        Staging &s = it->second;
        if (e == hipSuccess) for (const auto &p : s.pending) memcpy(p.dst, s.host + p.off, p.bytes);
        s.pending.clear();
        s.used = 0;
    }
    return e;
}

// wait for an event recorded on `st` BEHIND its read-backs instead of for the whole stream: what was queued behind the event (the arming
// of the next apply, api.hip) runs on while the host goes back to its caller.  Every pending read-back of `st` must lie before `ev`.
hipError_t event_sync(hipStream_t st, hipEvent_t ev)
{
    const hipError_t e = hipEventSynchronize(ev);
    std::lock_guard<std::mutex> lk(g_stage_mu);
    auto it = g_stage.find(st);
    if (it != g_stage.end()) {
        Staging &s = it->second;
        if (e == hipSuccess) for (const auto &p : s.pending) memcpy(p.dst, s.host + p.off, p.bytes);
        s.pending.clear();
        s.used = 0;
    }
    return e;
}

__global__ void k_iota_i32(int32_t *p, int64_t count)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) p[i] = (int32_t)i;
}
void iota_i32(hipStream_t st, int32_t *p, int64_t count)
{
    unsigned gb = (unsigned)((count + 255) / 256);
    if (gb > 4096) gb = 4096;
    if (gb < 1) gb = 1;
    hipLaunchKernelGGL(k_iota_i32, dim3(gb), dim3(256), 0, st, p, count);
}

__global__ void k_min_row_len(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int diag_at, int32_t *out)
{
    int m = 0x7fffffff;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const int lo = ptr[r], hi = ptr[r + 1];
        int len = hi - lo;
        // the entry at the diagonal's POSITION is not the diagonal (it was dropped): counts like an empty slice
        if (len > 0 && diag_at == 1 && idx[lo] != r) len = 0;
        if (len > 0 && diag_at == 2 && idx[hi - 1] != r) len = 0;
        m = min(m, len);
    }
    for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMin(out, m);
}
// length of the shortest major slice; 0 also when a slice does not have its diagonal where the solves take it from
// (diag_at: 1 first, 2 last, 0 not checked).  0 = only the guarded, purely positional row-parallel sweep may run on the factor.
int32_t min_row_len(hipStream_t st, int32_t n, const int32_t *ptr, const int32_t *idx, int diag_at)
{
    int32_t *d = nullptr, h = 0x7fffffff;
    ILUPP_HIP(pool_malloc(&d, sizeof(int32_t)));
    ILUPP_HIP(hipMemcpyAsync(d, &h, sizeof(int32_t), hipMemcpyHostToDevice, st));
    unsigned gb = (unsigned)((n + 255) / 256);
    if (gb > 1024) gb = 1024;
    hipLaunchKernelGGL(k_min_row_len, dim3(gb), dim3(256), 0, st, n, ptr, idx, diag_at, d);
    ILUPP_HIP(hipMemcpyAsync(&h, d, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    ILUPP_HIP(pool_free(d));
    return h;
}

int device_cu_count()
{
    int dev = 0;
    hipDeviceProp_t prop;
    ILUPP_HIP(hipGetDevice(&dev));
    ILUPP_HIP(hipGetDeviceProperties(&prop, dev));
    return prop.multiProcessorCount;
}

__global__ void k_fill_u64(unsigned long long *p, int64_t count, unsigned long long v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) p[i] = v;
}

void fill_u64(hipStream_t st, unsigned long long *p, int64_t count, unsigned long long v)
{
    if (count <= 0) return;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)blocks), dim3(256), 0, st, p, count, v);
}

// first position in [lo,hi) with idx[pos] >= key
__device__ __forceinline__ int lower_bound_dev(const int32_t *idx, int lo, int hi, int key)
{
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (idx[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// cutf[r] = 1 iff row r has no entry in column r-1  (a forward block may start at r)
// cutb[r] = 1 iff row r-1 has no entry in column r  (a backward block may end just above r-1 ... i.e. cut between r-1 and r)
// stats[0] += #forward cuts, stats[1] += #backward cuts, stats[2] = max row length
__global__ void k_row_cuts(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                           uint8_t *__restrict__ cutf, uint8_t *__restrict__ cutb, int32_t *__restrict__ stats)
{
    // grid-stride with per-thread partial sums and ONE set of atomics per workgroup: three words shared by
    // a quarter of a million waves would serialise in L2
    int nf = 0, nbk = 0, mx = 0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const int lo = ptr[r], hi = ptr[r + 1];
        mx = max(mx, hi - lo);
        const int pos = lower_bound_dev(idx, lo, hi, r - 1);
        bool has_prev = false, has_next = false;
        for (int q = pos; q < hi && q < pos + 3; ++q) {
            const int c = idx[q];
            has_prev |= (c == r - 1);
            has_next |= (c == r + 1);
        }
        const int cf = (r == 0) ? 1 : (has_prev ? 0 : 1);
        cutf[r] = (uint8_t)cf;
        nf += cf;
        if (r + 1 < n) { const int cb = has_next ? 0 : 1; cutb[r + 1] = (uint8_t)cb; nbk += cb; }
        if (r == 0) { cutb[0] = 1; nbk += 1; }
    }
    __shared__ int red[3];
    if (threadIdx.x < 3) red[threadIdx.x] = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) { nf += __shfl_xor(nf, off); nbk += __shfl_xor(nbk, off); mx = max(mx, __shfl_xor(mx, off)); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&red[0], nf); atomicAdd(&red[1], nbk); atomicMax(&red[2], mx); }
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(&stats[0], red[0]); atomicAdd(&stats[1], red[1]); atomicMax(&stats[2], red[2]); }
}

// start[b] = first allowed cut in the nominal cell [b*B, (b+1)*B), or the cell's end if the cell
// lies inside one long chain (the chain is then split at a cell border: correct, merely serial).
__global__ void k_block_starts(int32_t n, int32_t B, int32_t nb, const uint8_t *__restrict__ cut, int32_t *__restrict__ start,
                               int32_t *__restrict__ ragged)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    if (b == nb) { start[nb] = n; return; }
    const int64_t lo = (int64_t)b * B;
    // *ragged: some nominal cell does not begin with the start of a chain (with as many chains as cells -- the host knows both
    // numbers -- every block is then exactly one chain)
    if (ragged && !cut[lo]) *ragged = 1;
    if (b == 0) { start[0] = 0; return; }
    int64_t hi = lo + B;
    if (hi > n) hi = n;
    int64_t s = hi;
    for (int64_t p = lo; p < hi; ++p)
        if (cut[p]) { s = p; break; }
    start[b] = (int32_t)s;
}

static void make_schedule(hipStream_t st, int32_t n, const uint8_t *cut, int32_t ncuts, int max_lanes, Schedule *sch, int32_t *ragged = nullptr)
{
    // rows per lane: at least n/max_lanes, and about one chain when chains are long
    int64_t B = ((int64_t)n + max_lanes - 1) / max_lanes;
    int64_t chain = ncuts > 0 ? ((int64_t)n + ncuts - 1) / ncuts : n;
    if (chain > B) B = chain;
    // no chains to speak of (rows rarely depend on their predecessor: random matrices): one row per lane.  A lane that owns
    // a block of unrelated rows makes each of them wait for everything before it in the block (ILU(0) numeric on a random
    // matrix with n = 1e6: 48.7 ms with 15 rows per lane, 3.3 ms with one)
    if (chain <= 4) B = 1;
    if (B < 1) B = 1;
    int64_t nb = ((int64_t)n + B - 1) / B;
    sch->nb = (int32_t)nb;
    sch->B = (int32_t)B;
    ILUPP_HIP(pool_malloc(&sch->start, sizeof(int32_t) * (size_t)(nb + 1)));
    hipLaunchKernelGGL(k_block_starts, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0, st,
                       n, (int32_t)B, (int32_t)nb, cut, sch->start, ragged);
}

// ILU(0) analysis, first pass over A's pattern: everything k_row_cuts finds plus the row counts of L (lrow[r] =
// strictly-lower entries + unit diagonal, ILU0.hpp:93) and the first row without a diagonal entry.
// stats[3] = min row with no diagonal (INT_MAX when none)
// side entries of a row: everything but the diagonal and the two links of its chain (columns r - 1, r + 1).  Rows r - 1 and r of one
// chain are ALIKE when their side entries have the same offsets from the diagonal, row r - 1 has column r, and both fit 8 entries.
// (the light form of st_direct.hip's statement about lanes: with every chain alike and every lane exactly one chain, the lane
// templates that k_st_template takes from three sampled rows hold for all rows)
// cur / prv: the (at most 8) columns of rows r and r - 1, padded with INT_MAX; side(r) = cl entries left of column r - 1 and the
// entries from place us on; the same for row r - 1 relative to its own diagonal.  Bit masks instead of loops: the pass reads
// 0.6 GB at HBM speed, which leaves about 250 lane-instructions per row.
__device__ __forceinline__ bool rows_alike(const Row8 &cur, int len, int cl, int us, const Row8 &prv, int plen, int pcl, int pus, bool phn)
{
    if (len > 8 || plen > 8 || !phn) return false;                      // (!phn: (r, r-1) stored without (r-1, r))
    if (cl != pcl || len - us != plen - pus) return false;
    if (us == pus) {
        // same places in both rows (the rule inside a chain): entry by entry, row r - 1 shifted by one column
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) bad |= (cur.c[i] != prv.c[i] + 1 ? 1u : 0u) << i;
        const unsigned side = ((1u << cl) - 1u) | (((1u << len) - 1u) & ~((1u << us) - 1u));
        return (bad & side) == 0;
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) ok = ok && (i >= cl || cur.c[i] == prv.c[i] + 1);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int a = us + i, b = pus + i;
        if (a < len) ok = ok && ROW8_AT(cur, a > 7 ? 7 : a) == ROW8_AT(prv, b > 7 ? 7 : b) + 1;
    }
    return ok;
}

__global__ void k_row_cuts_counts(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int64_t nnz,
                                  uint8_t *__restrict__ cutf, uint8_t *__restrict__ cutb, int32_t *__restrict__ lrow,
                                  int32_t *__restrict__ stats)
{
    int nf = 0, nbk = 0, mx = 0, miss = 0x7fffffff, nl = 0, nun = 0;
    for (int r0 = blockIdx.x * blockDim.x; r0 < n; r0 += gridDim.x * blockDim.x) {
        const int r = r0 + (int)threadIdx.x;
        const bool live = r < n;
        const int lo = live ? ptr[r] : 0, hi = live ? ptr[r + 1] : 0;
        mx = max(mx, hi - lo);
        int cl = 0;
        bool has_prev = false, has_next = false, has_diag = false;
        Row8 row;
#pragma unroll
        for (int i = 0; i < 8; ++i) row.c[i] = 0x7fffffff;
        if (live && hi - lo <= 8) {
            row = load_row8(idx, lo, hi - lo, nnz);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                cl += row.c[i] < r ? 1 : 0;
                has_prev |= row.c[i] == r - 1; has_next |= row.c[i] == r + 1; has_diag |= row.c[i] == r;
            }
        } else {
            for (int q = lo; q < hi; ++q) {
                const int c = idx[q];
                cl += c < r ? 1 : 0;
                has_prev |= c == r - 1; has_next |= c == r + 1; has_diag |= c == r;
            }
        }
        // the row before, from the lane before (the first lane of a wave reads it); its three counts travel packed
        // (DPP wave shift: one VALU instruction each; __shfl_up goes through the LDS crossbar)
        const int cls = cl - (has_prev ? 1 : 0);                              // side entries left of the diagonal block
        const int us = cl + 1 + (has_next ? 1 : 0);                           // first side entry right of it
        Row8 prv;
#pragma unroll
        for (int i = 0; i < 8; ++i) prv.c[i] = __builtin_amdgcn_update_dpp(0, row.c[i], 0x138, 0xf, 0xf, false);
        int pk = __builtin_amdgcn_update_dpp(0, cls | (us << 4) | ((has_next ? 1 : 0) << 8) | ((hi - lo) << 9), 0x138, 0xf, 0xf, false);
        if ((threadIdx.x & 63) == 0 && live && r > 0) {
            const int plo = ptr[r - 1];
            const int plen = lo - plo;
            int pcl = 0, php = 0, phn = 0;
            if (plen <= 8) {
                prv = load_row8(idx, plo, plen, nnz);
#pragma unroll
                for (int i = 0; i < 8; ++i) { pcl += prv.c[i] < r - 1 ? 1 : 0; php |= prv.c[i] == r - 2; phn |= prv.c[i] == r; }
            }
            pk = (pcl - php) | ((pcl + 1 + phn) << 4) | (phn << 8) | (plen << 9);
        }
        if (live) {
            const int pcls = pk & 15, pus = (pk >> 4) & 15, plen = pk >> 9;
            const bool phn = (pk >> 8) & 1;
            if (has_prev && !rows_alike(row, hi - lo, cls, us, prv, plen, pcls, pus, phn)) ++nun;
            // a chain starts here: the row before must not point at this one either
            if (!has_prev && r > 0 && plen <= 8 && phn) ++nun;
            if (lrow) lrow[r] = cl + 1;
            nl += cl + 1;
            if (!has_diag) miss = min(miss, r);
            const int cf = (r == 0) ? 1 : (has_prev ? 0 : 1);
            cutf[r] = (uint8_t)cf;
            nf += cf;
            if (r + 1 < n) { const int cb = has_next ? 0 : 1; cutb[r + 1] = (uint8_t)cb; nbk += cb; }
            if (r == 0) { cutb[0] = 1; nbk += 1; }
        }
    }
    __shared__ int red[6];
    if (threadIdx.x < 3) red[threadIdx.x] = 0;
    if (threadIdx.x == 3) red[3] = 0x7fffffff;
    if (threadIdx.x == 4) red[4] = 0;
    if (threadIdx.x == 5) red[5] = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        nf += __shfl_xor(nf, off); nbk += __shfl_xor(nbk, off); mx = max(mx, __shfl_xor(mx, off)); miss = min(miss, __shfl_xor(miss, off)); nl += __shfl_xor(nl, off);
        nun += __shfl_xor(nun, off);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&red[0], nf); atomicAdd(&red[1], nbk); atomicMax(&red[2], mx); atomicMin(&red[3], miss); atomicAdd(&red[4], nl); atomicAdd(&red[5], nun); }
    __syncthreads();
    // per-block partial results; k_reduce_stats folds them (same-address atomics cost ~10 ns each on this chip: four per
    // block were 0.16 of this kernel's 0.26 ms at 4096 blocks, and 1.2 ms at 32768)
    if (threadIdx.x == 0) { int *o = stats + 8 + 8 * blockIdx.x; o[0] = red[0]; o[1] = red[1]; o[2] = red[2]; o[3] = red[3]; o[4] = red[4]; o[5] = red[5] > 0 ? 1 : 0; }
}

__global__ void k_reduce_stats(int nblocks, int32_t *stats)
{
    int nf = 0, nbk = 0, mx = 0, miss = 0x7fffffff, nun = 0;
    long long nl = 0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
        const int *o = stats + 8 + 8 * b;
        nf += o[0]; nbk += o[1]; mx = max(mx, o[2]); miss = min(miss, o[3]); nl += o[4]; nun += o[5];
    }
    __shared__ int red[5];
    __shared__ unsigned long long rl;
    if (threadIdx.x < 3) red[threadIdx.x] = 0;
    if (threadIdx.x == 3) red[3] = 0x7fffffff;
    if (threadIdx.x == 4) { rl = 0; red[4] = 0; }
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        nf += __shfl_xor(nf, off); nbk += __shfl_xor(nbk, off); mx = max(mx, __shfl_xor(mx, off)); miss = min(miss, __shfl_xor(miss, off));
        nl += __shfl_xor(nl, off); nun += __shfl_xor(nun, off);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&red[0], nf); atomicAdd(&red[1], nbk); atomicMax(&red[2], mx); atomicMin(&red[3], miss); atomicAdd(&rl, (unsigned long long)nl); atomicAdd(&red[4], nun); }
    __syncthreads();
    // stats[4..5]: entries of L (strictly lower + unit diagonal), 64 bits; stats[6]: blocks of rows with a chain that is not alike
    if (threadIdx.x == 0) { stats[0] = red[0]; stats[1] = red[1]; stats[2] = red[2]; stats[3] = red[3]; *reinterpret_cast<unsigned long long *>(stats + 4) = rl; stats[6] = red[4]; stats[7] = 0; }
}

// after the stream_sync that follows ilu0_symbolic_and_schedule
void finish_chains(Schedule *fwd, Schedule *bwd) { fwd->chains = bwd->chains = fwd->chains_pre && fwd->ragged == 0; }

// L/U patterns from the row pointers of L alone: Uptr[r] = Aptr[r] - (Lptr[r] - r)
__global__ void k_ilu0_pattern2(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int64_t nnz,
                                const int32_t *__restrict__ Lptr, int32_t *__restrict__ Uptr,
                                int32_t *__restrict__ Lidx, int32_t *__restrict__ Uidx)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int lo = ptr[r], hi = ptr[r + 1];
    int l = Lptr[r], u = lo - (l - r);
    Uptr[r] = u;
    if (r == n - 1) Uptr[n] = hi - (Lptr[n] - n);
    if (hi - lo <= 8) {
        const Row8 row = load_row8(idx, lo, hi - lo, nnz);
        int cl = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) cl += row.c[i] < r ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < hi - lo) { if (i < cl) Lidx[l + i] = row.c[i]; else Uidx[u + i - cl] = row.c[i]; }
        }
        Lidx[l + cl] = r;
        return;
    }
    for (int q = lo; q < hi; ++q) {
        const int c = idx[q];
        if (c < r) Lidx[l++] = c; else Uidx[u++] = c;
    }
    Lidx[l] = r;
}

// ILU(0): symbolic factorisation (L/U patterns, ILU0.hpp:85-98) and both sweep schedules with ONE pass over
// the pattern for the counts, one scan and one host round trip
int ilu0_symbolic_and_schedule(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U, int32_t *first_missing_diag,
                               int max_lanes, Schedule *fwd, Schedule *bwd, int32_t *max_row_len)
{
    // (the CSR row pointers and index arrays of L and U are NOT made here: the static level-major kernels never read them --
    // st_make_csr builds them from the records when somebody asks for the factors -- and the other generations call
    // ilu0_csr_ptrs below once the static analysis has declined the matrix)
    const int32_t n = A.n;
    uint8_t *cutf = nullptr, *cutb = nullptr;
    int32_t *stats = nullptr;
    ILUPP_HIP(pool_malloc(&cutf, (size_t)n + 1));
    ILUPP_HIP(pool_malloc(&cutb, (size_t)n + 1));
    unsigned gb = (unsigned)((n + 255) / 256);
    static const unsigned gb_max = []() { const char *e = getenv("ILUPP_RCC_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? (unsigned)v : 16384u; }();
    if (gb > gb_max) gb = gb_max;
    ILUPP_HIP(pool_malloc(&stats, sizeof(int32_t) * (8 + 8 * (size_t)gb)));
    hipLaunchKernelGGL(k_row_cuts_counts, dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, (int64_t)A.nnz, cutf, cutb,
                       static_cast<int32_t *>(nullptr), stats);
    hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(1024), 0, st, (int)gb, stats);
    L->n = U->n = n; L->is_csr = U->is_csr = true; L->owns = U->owns = true;
    int32_t h[8];
    ILUPP_HIP(d2h_async(st, h, stats, sizeof(h)));
    ILUPP_HIP(stream_sync(st));
    if (max_row_len) *max_row_len = h[2];
    if (first_missing_diag) *first_missing_diag = (h[3] == 0x7fffffff) ? -1 : h[3];
    long long nnzl = 0;
    memcpy(&nnzl, h + 4, sizeof(nnzl));
    L->nnz = nnzl;
    U->nnz = A.nnz - (nnzl - n);
    if (h[3] != 0x7fffffff) { ILUPP_HIP(pool_free(cutf)); ILUPP_HIP(pool_free(cutb)); ILUPP_HIP(pool_free(stats)); return ILUPP_ERR_NO_DIAGONAL; }
    make_schedule(st, n, cutf, h[0], max_lanes, fwd, stats + 7);
    make_schedule(st, n, cutb, h[1], max_lanes, bwd, stats + 7);
    // every lane one whole chain, all chains alike (st_direct.hip's premise, in its light form).  The blocks' verdict is not waited for
    // here (a host round trip is 10-20 us of idle stream): it arrives with the caller's next stream_sync, finish_chains() folds it in
    fwd->chains = bwd->chains = false;
    fwd->chains_pre = bwd->chains_pre = h[6] == 0 && h[0] == fwd->nb && h[1] == bwd->nb;
    fwd->ragged = 1;
    ILUPP_HIP(d2h_async(st, &fwd->ragged, stats + 7, sizeof(fwd->ragged)));
    ILUPP_HIP(pool_free(cutf));
    ILUPP_HIP(pool_free(cutb));
    ILUPP_HIP(pool_free(stats));
    return ILUPP_OK;
}

// row pointers of L (strictly-lower entries + unit diagonal, ILU0.hpp:93) and U, and room for their entries
__global__ void k_ilu0_lrow(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t *__restrict__ lrow)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int cl = 0;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) cl += idx[q] < r ? 1 : 0;
    lrow[r] = cl + 1;
}
int csr_ptrs_from_counts(hipStream_t st, int32_t n, int32_t *lrow, DevMat *L)
{
    ILUPP_HIP(pool_malloc(&L->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(hipMemsetAsync(L->ptr, 0, sizeof(int32_t), st));
    size_t tmp_bytes = 0;
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, lrow, L->ptr + 1, n, st));
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, lrow, L->ptr + 1, n, st));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(tmp));
    ILUPP_HIP(pool_malloc(&L->idx, sizeof(int32_t) * (size_t)(L->nnz > 0 ? L->nnz : 1)));
    ILUPP_HIP(pool_malloc(&L->val, sizeof(double) * (size_t)(L->nnz > 0 ? L->nnz : 1)));
    return ILUPP_OK;
}
int ilu0_csr_ptrs(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U)
{
    const int32_t n = A.n;
    int32_t *lrow = nullptr;
    ILUPP_HIP(pool_malloc(&lrow, sizeof(int32_t) * (size_t)n));
    hipLaunchKernelGGL(k_ilu0_lrow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, A.ptr, A.idx, lrow);
    int rc = csr_ptrs_from_counts(st, n, lrow, L);
    ILUPP_HIP(pool_free(lrow));
    if (rc) return rc;
    ILUPP_HIP(pool_malloc(&U->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(pool_malloc(&U->idx, sizeof(int32_t) * (size_t)(U->nnz > 0 ? U->nnz : 1)));
    ILUPP_HIP(pool_malloc(&U->val, sizeof(double) * (size_t)(U->nnz > 0 ? U->nnz : 1)));
    return ILUPP_OK;
}

// the split of A's pattern into the patterns of L (unit diagonal appended last) and U (diagonal first), ILU0.hpp:85-98;
// needs L->ptr from ilu0_symbolic_and_schedule
void ilu0_write_patterns(hipStream_t st, const DevMat &A, DevMat *L, DevMat *U)
{
    const int32_t n = A.n;
    hipLaunchKernelGGL(k_ilu0_pattern2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, A.ptr, A.idx, (int64_t)A.nnz,
                       L->ptr, U->ptr, L->idx, U->idx);
}

int count_cuts_and_schedule(hipStream_t st, int32_t n, const int32_t *ptr, const int32_t *idx,
                            int max_lanes, Schedule *fwd, Schedule *bwd, int32_t *max_row_len)
{
    uint8_t *cutf = nullptr, *cutb = nullptr;
    int32_t *stats = nullptr;
    ILUPP_HIP(pool_malloc(&cutf, (size_t)n + 1));
    ILUPP_HIP(pool_malloc(&cutb, (size_t)n + 1));
    ILUPP_HIP(pool_malloc(&stats, sizeof(int32_t) * 4));
    ILUPP_HIP(hipMemsetAsync(stats, 0, sizeof(int32_t) * 4, st));
    {
        unsigned gb = (unsigned)((n + 255) / 256);
        if (gb > 2048) gb = 2048;
        hipLaunchKernelGGL(k_row_cuts, dim3(gb), dim3(256), 0, st, n, ptr, idx, cutf, cutb, stats);
    }
    int32_t h[4];
    ILUPP_HIP(d2h_async(st, h, stats, sizeof(h)));
    ILUPP_HIP(stream_sync(st));
    if (max_row_len) *max_row_len = h[2];
    if (fwd) make_schedule(st, n, cutf, h[0], max_lanes, fwd);
    if (bwd) make_schedule(st, n, cutb, h[1], max_lanes, bwd);
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(cutf));
    ILUPP_HIP(pool_free(cutb));
    ILUPP_HIP(pool_free(stats));
    return ILUPP_OK;
}

// ---------------------------------------------------------------------------------------------
// transposed storage: (ptr, idx, val) of A  ->  the same arrays of A^T, entries of each new major
// slice in ascending order of the old major index (stable sort by column).
// ---------------------------------------------------------------------------------------------
__global__ void k_expand_rows(int32_t n, const int32_t *__restrict__ ptr, int32_t *__restrict__ rowid, int32_t *__restrict__ seq)
{
    // one wave per 64 rows would be enough; rows are short, so thread-per-row with a short loop
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) { rowid[q] = r; seq[q] = q; }
}

__global__ void k_gather_transposed(int64_t nnz, const int32_t *__restrict__ perm, const int32_t *__restrict__ rowid,
                                    const double *__restrict__ val, int32_t *__restrict__ tidx, double *__restrict__ tval)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < nnz; e += stride) {
        const int p = perm[e];
        tidx[e] = rowid[p];
        tval[e] = val[p];
    }
}

__global__ void k_ptr_from_sorted(int32_t n, int64_t nnz, const int32_t *__restrict__ sorted_cols, int32_t *__restrict__ tptr)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n) return;
    // first position with sorted_cols[pos] >= c
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (sorted_cols[mid] < c) lo = mid + 1; else hi = mid;
    }
    tptr[c] = (int32_t)lo;
}

// Short slices (stencil-like factors: the first apply of an LL^T object, the transposed apply of an ILU(0) object): count, scan, scatter
// with one atomic per entry, then every new slice sorts its few entries in registers -- the same arrays as the sort by column below
// (a (row, column) pair occurs once: the order inside a slice is determined), without three radix passes over all entries
// (256^3, ICholT(0, 0): 2.8 -> 0.6 ms of the first apply).  A slice of more than kTrMax entries sends the matrix the other way.
static constexpr int kTrMax = 8;
__global__ void k_tr_count(int64_t nnz, const int32_t *__restrict__ idx, int32_t *__restrict__ cnt)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < nnz; e += stride) atomicAdd(&cnt[idx[e]], 1);
}
__global__ void k_tr_scatter(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
                             const int32_t *__restrict__ tptr, int32_t *__restrict__ fill, int32_t *__restrict__ tidx, double *__restrict__ tval)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int c = idx[q];
        const int pos = tptr[c] + atomicAdd(&fill[c], 1);
        tidx[pos] = r; tval[pos] = val[q];
    }
}
__global__ void k_tr_sort(int32_t n, const int32_t *__restrict__ tptr, int32_t *__restrict__ tidx, double *__restrict__ tval, int32_t *__restrict__ toolong)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const int p0 = tptr[c], len = tptr[c + 1] - p0;
    if (len > kTrMax) { atomicOr(toolong, 1); return; }
    int ki[kTrMax]; double kv[kTrMax];
#pragma unroll
    for (int a = 0; a < kTrMax; ++a) { ki[a] = a < len ? tidx[p0 + a] : 0x7fffffff; kv[a] = a < len ? tval[p0 + a] : 0.0; }
    // (a fixed network of compare-exchanges: the arrays stay in registers)
#pragma unroll
    for (int a = 0; a < kTrMax; ++a)
#pragma unroll
        for (int b = 0; b + 1 < kTrMax - a; ++b)
            if (ki[b] > ki[b + 1]) { const int ti = ki[b]; ki[b] = ki[b + 1]; ki[b + 1] = ti; const double tv = kv[b]; kv[b] = kv[b + 1]; kv[b + 1] = tv; }
#pragma unroll
    for (int a = 0; a < kTrMax; ++a) if (a < len) { tidx[p0 + a] = ki[a]; tval[p0 + a] = kv[a]; }
}
static bool transpose_short(hipStream_t st, const DevMat &A, DevMat *T)
{
    const int32_t n = A.n;
    const int64_t nnz = A.nnz;
    int32_t *cnt = nullptr, *flag = nullptr;
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&cnt, sizeof(int32_t) * ((size_t)n + 1)));
    ILUPP_HIP(pool_malloc(&flag, 16));
    ILUPP_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * ((size_t)n + 1), st));
    ILUPP_HIP(hipMemsetAsync(flag, 0, 16, st));
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_tr_count, dim3((unsigned)blocks), dim3(256), 0, st, nnz, A.idx, cnt);
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, T->ptr, n + 1, st));
    ILUPP_HIP(pool_malloc(&tmp, tb > 0 ? tb : 16));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cnt, T->ptr, n + 1, st));
    ILUPP_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)n, st));
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    hipLaunchKernelGGL(k_tr_scatter, g, b, 0, st, n, A.ptr, A.idx, A.val, T->ptr, cnt, T->idx, T->val);
    hipLaunchKernelGGL(k_tr_sort, g, b, 0, st, n, T->ptr, T->idx, T->val, flag);
    int32_t h = 0;
    ILUPP_HIP(d2h_async(st, &h, flag, sizeof(h)));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(tmp)); ILUPP_HIP(pool_free(cnt)); ILUPP_HIP(pool_free(flag));
    return h == 0;
}

void transpose_storage(hipStream_t st, const DevMat &A, DevMat *T, const GridDims *llt_grid)
{
    const int32_t n = A.n;
    const int64_t nnz = A.nnz;
    T->n = n; T->nnz = nnz; T->is_csr = !A.is_csr; T->owns = true;
    ILUPP_HIP(pool_malloc(&T->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(pool_malloc(&T->idx, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
    ILUPP_HIP(pool_malloc(&T->val, sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
    static const bool no_short = getenv("ILUPP_NO_SHORT_TRANSPOSE") != nullptr;
    // (a column-major LL^T factor with the grid's lower pattern: where every entry sits follows from the dimensions, grid.hip)
    if (llt_grid && getenv("ILUPP_NO_LLT_GRID_ROWS") == nullptr && llt_grid_rows(st, A, *llt_grid, T)) return;
    if (!no_short && n >= 4096 && nnz > 0 && nnz <= 8 * (int64_t)n && transpose_short(st, A, T)) return;
    int32_t *rowid, *seq, *keys_out, *perm;
    const size_t eb = sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1);
    ILUPP_HIP(pool_malloc(&rowid, eb));
    ILUPP_HIP(pool_malloc(&seq, eb));
    ILUPP_HIP(pool_malloc(&keys_out, eb));
    ILUPP_HIP(pool_malloc(&perm, eb));
    hipLaunchKernelGGL(k_expand_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, A.ptr, rowid, seq);
    size_t tmp_bytes = 0;
    int end_bit = 1;
    while ((1ll << end_bit) < (long long)n && end_bit < 31) ++end_bit;
    ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, A.idx, keys_out, seq, perm, (int)nnz, 0, end_bit, st));
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, A.idx, keys_out, seq, perm, (int)nnz, 0, end_bit, st));
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_gather_transposed, dim3((unsigned)blocks), dim3(256), 0, st, nnz, perm, rowid, A.val, T->idx, T->val);
    hipLaunchKernelGGL(k_ptr_from_sorted, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, n, nnz, keys_out, T->ptr);
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(tmp));
    ILUPP_HIP(pool_free(rowid)); ILUPP_HIP(pool_free(seq)); ILUPP_HIP(pool_free(keys_out)); ILUPP_HIP(pool_free(perm));
}

}  // namespace ilupp
