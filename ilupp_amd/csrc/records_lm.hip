// ilupp_amd/csrc/records_lm.hip -- analysis for the level-major kernels straight from A's pattern.
//
// For a matrix whose ILU(0) rows are short (<= 3 eliminations, <= 4 entries in a U row: the 5-/7-point
// stencils) everything the level-major factor kernel (ilu0_lm.hip) and the two level-major sweeps
// (sptrsv_lm.hip) need is written here in two passes over A's pattern, one per direction, in the order the
// kernels will read it: no descriptor arrays, no update program, no intermediate CSR stream.
//
//   forward pass, per (chunk, lane) = row r of the forward schedule
//       L-sweep record, pattern half   {d0,d1,d2,valid}            (sptrsv_lm.hip)
//       factor record, pattern part    program header w0,w1 (schedule.hip "F3" semantics: reference merge order
//                                      ILU0.hpp:8-23) and {dep0,dep1,dep2,hdr}   (ilu0_lm.hip)
//   backward pass, per (chunk, lane) = row r of the backward schedule
//       U-sweep record, pattern half   {d0,d1,d2,valid}
//
// Both passes check what the kernels rely on (row shapes, step order inside a workgroup, ticket order between
// workgroups); the caller falls back to the CSR machinery when a flag comes back set.
#include <stdio.h>
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ilupp {

typedef int v4i __attribute__((ext_vector_type(4)));

static constexpr int kNoDep = -1;
static constexpr int kOwnPrev = -3;
static constexpr int kMaxSkewA = 30000;
#ifndef REC_ROWS
#define REC_ROWS 1
#endif
static constexpr int kRecRows = REC_ROWS;     // groups of 8 rows a block of the record kernels walks per lane (measured: 4 is slower, 3.05 vs 2.57 ms analysis)
struct __attribute__((aligned(8))) D2r { double v[2]; };

// Import table from one triangle of A (tri = +1: columns below the diagonal, -1: above); see k_ghost_table
__global__ void __launch_bounds__(kThreads)
k_ghost_table_tri(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t B, int32_t nb,
                  const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot, int tri,
                  const int32_t *__restrict__ sfirst, const int32_t *__restrict__ scount, int32_t *__restrict__ gtab)
{
    __shared__ int tab[kGhosts];
    const int wg = blockIdx.x, t = threadIdx.x;
    if (t < kGhosts) tab[t] = -1;
    __syncthreads();
    const int slot = wg * kThreads + t;
    const int cnt = scount[slot], first = sfirst[slot];
    for (int sample = 0; sample < 3 && cnt > 0; ++sample) {
        const int k = sample == 0 ? 0 : (sample == 1 ? cnt / 2 : cnt - 1);
        if (sample > 0 && k == (sample == 1 ? 0 : cnt / 2)) continue;
        const int r = first + tri * k;
        for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
            const int c = idx[q];
            if (tri > 0 ? c >= r : c <= r) continue;
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

// skews as in k_lm_skew (sptrsv_lm.hip), the sampled constraints taken from A's triangle instead of descriptors
__global__ void __launch_bounds__(kThreads)
k_lm_skew_tri(const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t B, int32_t nb,
              const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot, int tri,
              const int32_t *__restrict__ sfirst, const int32_t *__restrict__ scount,
              int32_t *__restrict__ skew, int32_t *__restrict__ wtab, int32_t *__restrict__ flags)
{
    __shared__ int s[kThreads];
    const int wg = blockIdx.x, t = threadIdx.x;
    const int slot = wg * kThreads + t;
    const int cnt = scount[slot], first = sfirst[slot];
    int cb[12], cd[12], nc = 0;
    for (int sample = 0; sample < 3 && cnt > 0; ++sample) {
        const int k = sample == 0 ? 0 : (sample == 1 ? cnt / 2 : cnt - 1);
        if ((sample == 1 && k == 0) || (sample == 2 && (k == 0 || k == cnt / 2))) continue;
        const int r = first + tri * k;
        const int q0 = ptr[r], q1 = ptr[r + 1];
        for (int q = q0; q < q1 && q < q0 + 8; ++q) {
            const int c = idx[q];
            if (tri > 0 ? c >= r : c <= r) continue;
            const int b = block_of(c, B, nb, start);
            const int os = blk2slot[b];
            if ((os >> 8) != wg || (os & 255) == t) continue;
            const int kl = tri > 0 ? c - start[b] : start[b + 1] - 1 - c;
            if (nc < 12) { cb[nc] = os & 255; cd[nc] = kl + 1 - k; ++nc; }
        }
    }
    s[t] = 0;
    __syncthreads();
    for (int it = 0; it < 600; ++it) {
        int v = s[t];
        for (int j = 0; j < nc; ++j) { const int c = s[cb[j]] + cd[j]; v = c > v ? c : v; }
        v = v > kMaxSkewA ? kMaxSkewA : v;
        const int changed = v != s[t];
        __syncthreads();
        s[t] = v;
        if (!__syncthreads_or(changed)) break;
    }
    const int sk = s[t];
    skew[slot] = sk;
    if (sk >= kMaxSkewA) atomicOr(&flags[0], 1);
    int lo = cnt > 0 ? sk : 0x7fffffff, hi = cnt > 0 ? sk + cnt : -0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
    if ((t & 63) == 0) {
        int32_t *w = wtab + (size_t)(wg * 4 + (t >> 6)) * 4;
        const int nch = hi > lo ? hi - lo : 0;
        w[0] = 0; w[1] = nch > 0 ? lo : 0; w[2] = nch; w[3] = 0;
    }
}

__global__ void __launch_bounds__(kThreads)
k_lm_scan_a(int32_t nwaves, int32_t *__restrict__ wtab, int32_t *__restrict__ flags)
{
    __shared__ int part[kThreads];
    __shared__ int pmax[kThreads];
    const int t = threadIdx.x;
    const int per = (nwaves + kThreads - 1) / kThreads;
    int sum = 0, mx = 0;
    for (int i = t * per; i < (t + 1) * per && i < nwaves; ++i) { const int c = wtab[(size_t)i * 4 + 2]; sum += c; mx = c > mx ? c : mx; }
    part[t] = sum; pmax[t] = mx;
    __syncthreads();
    if (t == 0) {
        int run = 0, m = 0;
        for (int i = 0; i < kThreads; ++i) { const int c = part[i]; part[i] = run; run += c; m = pmax[i] > m ? pmax[i] : m; }
        flags[1] = run; flags[2] = m;
    }
    __syncthreads();
    int run = part[t];
    for (int i = t * per; i < (t + 1) * per && i < nwaves; ++i) { wtab[(size_t)i * 4] = run; run += wtab[(size_t)i * 4 + 2]; }
}

// ---------------------------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------------------------
// flags[0] |= 2 : a row or dependency the level-major kernels cannot take; flags[4] |= 1 : step structure unfit
// for the factor kernel (an in-workgroup dependency that is not exactly one step back)
// A block owns the 64 lanes of one wave x 8 consecutive rows of each; inside a wave 8 lanes x 8 rows, so a load
// instruction touches 8 short contiguous segments of A and a store instruction lands on diagonals of 8 neighbouring
// places (the skew of neighbouring lanes differs by one step).  k_pad_records marks the places no row lands on.
__global__ void __launch_bounds__(512)
k_fwd_records(const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, const double *__restrict__ Aval, int64_t nnz, int32_t B, int32_t nb,
              const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot,
              const int32_t *__restrict__ wtab, const int32_t *__restrict__ skew, const int32_t *__restrict__ sfirst,
              const int32_t *__restrict__ scount, const int32_t *__restrict__ gtab, int32_t *__restrict__ exported,
              v4i *__restrict__ pkL, v4i *__restrict__ pkA, int32_t *__restrict__ flags)
{
    const int w = blockIdx.x;
    const int L = (threadIdx.x >> 6) * 8 + (threadIdx.x & 7);
    const int wg = w >> 2;
    const int slot = wg * kThreads + (w & 3) * 64 + L;
    // kRecRows consecutive groups of 8 rows per lane and block: the pieces of A a lane reads (28 + 56 bytes per row) are
    // then long enough that the 128-byte lines at their ends are fetched once, not once per neighbouring block
    for (int jj = 0; jj < kRecRows; ++jj) {
    const int k = (blockIdx.y * kRecRows + jj) * 8 + ((threadIdx.x >> 3) & 7);
    if (k >= scount[slot]) continue;
    const int r = sfirst[slot] + k;
    const int tau = k + skew[slot];
    const int base = wtab[(size_t)w * 4], c = tau - wtab[(size_t)w * 4 + 1];
    v4i lrec; lrec.x = kNoDep; lrec.y = kNoDep; lrec.z = kNoDep; lrec.w = 0;
    v4i dec; dec.x = 0; dec.y = 0; dec.z = 0; dec.w = 0;
    int w0 = 0, w1 = 0;
    {
        const int a0 = Aptr[r], a1 = Aptr[r + 1];
        const int len = a1 - a0;
        const Row8 own = load_row8(Aidx, a0, len > 8 ? 8 : len, nnz);
        int cl = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) cl += own.c[i] < r ? 1 : 0;
        int bad = (len > 7 || cl > 3 || len - cl > 4 || len - cl < 1) ? 1 : 0;
        int badf = 0;
        unsigned long long mbits = 0;
        int nmt = 0, kinds = 0;
        int sd[3] = {kNoDep, kNoDep, kNoDep}, word[3] = {0, 0, 0};
        if (!bad) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
#ifdef EXP_REC_NODEP
                continue;
#endif
                if (e >= cl) continue;
                const int sl = e + 3 - cl;                    // right-aligned dependency slot of the program
                const int kc = own.c[e];
                // matches of U row kc against the rest of this row (merge order of the reference): the U entries in
                // ascending column, each looked up among this row's columns
                const int k0 = Aptr[kc], k1 = Aptr[kc + 1];
                const int klen = k1 - k0;
                const Row8 dep = load_row8(Aidx, k0, klen > 8 ? 8 : klen, nnz);
                int kd = 0;                                   // position of U row kc's diagonal in its row of A
#pragma unroll
                for (int i = 0; i < 8; ++i) kd += dep.c[i] < kc ? 1 : 0;
                if (klen > 8 || klen - kd > 4) bad = 1;       // U row longer than a hand-off entry
                int first_off = 0, nm_e = 0;
#pragma unroll
                for (int off = 1; off < 4; ++off) {
                    if (kd + off < klen) {
                        const int m = ROW8_AT(dep, kd + off);
                        int pp = 0, found = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { pp += own.c[i] < m ? 1 : 0; found |= own.c[i] == m ? 1 : 0; }
                        if (found) {
                            if (nmt < 5) mbits |= (unsigned long long)(sl | (off << 2) | (((pp - cl + 3) & 7) << 4)) << (7 * nmt);
                            if (nm_e == 0) first_off = off;
                            ++nmt; ++nm_e;
                        }
                    }
                }
                // who produces U row kc
                const int b = block_of(kc, B, nb, start);
                const int oslot = blk2slot[b];
                const int kloc = kc - start[b];
                if (oslot == slot && kloc == k - 1) {
                    sd[e] = kOwnPrev;
                    kinds |= 1 << (2 * sl);
                    word[sl] = first_off << 27;
                } else if ((oslot >> 8) == wg) {
                    const int td = kloc + skew[oslot];
                    if (td > tau) bad = 1;                    // the sweep would wait for a later step of its own workgroup
                    if (td != tau - 1) badf = 1;              // the factor kernel's short hand-off ring must not be lapped
                    sd[e] = (oslot << 15) | kloc;
                    word[sl] = ((kloc % kFlmUF) * 3 * kThreads + (oslot & 255)) | (kloc << 12) | (first_off << 27);
                    kinds |= 2 << (2 * sl);
                } else {
                    exported[oslot] = 1;                      // read across a workgroup border: stored write-through / exchanged
                    if ((oslot >> 8) >= wg) bad = 1;          // producer not ahead of us in ticket order
                    int g = -1;
                    if (kloc < 0x7fff) {
                        const int32_t *tab = gtab + (size_t)wg * kGhosts;
                        unsigned h = ((unsigned)oslot * 0x9E3779B1u) >> 26;
                        for (int probe = 0; probe < kGhosts; ++probe, h = (h + 1) & (kGhosts - 1)) {
                            const int cur = tab[h];
                            if (cur == oslot) { g = (int)h; break; }
                            if (cur == -1) break;
                        }
                    }
                    if (g >= 0) {
                        sd[e] = ((kGhostBase + g) << 15) | kloc;
                        word[sl] = (kFlmGhostBase + (kloc & (kFlmGF - 1)) * 3 * kGhosts + g) | (kloc << 12) | (first_off << 27) | (1 << 29);
                        kinds |= 2 << (2 * sl);
                    } else {
                        sd[e] = (oslot << 15) | kloc;
                        word[sl] = (oslot << 15) | kloc;
                        kinds |= 3 << (2 * sl);
                    }
                }
            }
            if (nmt > 5) bad = 1;
        }
        bool simple = !bad && nmt == cl;
        for (int m = 0; m < 3 && simple; ++m)
            if (m < nmt) {
                const int mw = (int)((mbits >> (7 * m)) & 127);
                simple = ((mw & 3) == m + 3 - cl) && (((mw >> 4) & 7) == 3);
            }
        if (bad) atomicOr(&flags[0], 2);
        if (badf) atomicOr(&flags[4], 1);
        w0 = (len & 15) | ((cl & 3) << 4) | (((nmt > 5 ? 5 : nmt) & 7) << 6) | (int)((mbits & 0x1FFFFFull) << 9) | (simple ? (1 << 30) : 0);
        w1 = (int)((mbits >> 21) & 0x3FFFull);
        lrec.x = sd[0]; lrec.y = sd[1]; lrec.z = sd[2]; lrec.w = 1;
        dec.x = word[0]; dec.y = word[1]; dec.z = word[2];
        dec.w = 1 | (kinds << 1) | (cl << 7) | ((len - cl) << 9) | ((simple ? 1 : 0) << 12) | ((nmt > 5 ? 5 : nmt) << 13);
    }
    __builtin_nontemporal_store(lrec, pkL + ((size_t)base + c) * 192 + L);
    v4i *p = pkA + ((size_t)base + c) * 320 + L;
    __builtin_nontemporal_store(dec, p + 256);
#ifdef EXP_REC_NOVAL
    if (true) {
#else
    if (!Aval) {
#endif
        int *q = reinterpret_cast<int *>(p + 192);
        q[2] = w0; q[3] = w1;
    } else {
        // the values of this factorisation right away (k_flm_pack_a does the same for a later one on the same pattern):
        // the row of A, diagonal-aligned
        const int len = w0 & 15, cl = (w0 >> 4) & 3;
        const int a0 = Aptr[r];
        double v[8];
        if ((int64_t)a0 + 8 <= nnz) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const D2r t = *reinterpret_cast<const D2r *>(Aval + a0 + 2 * i); v[2 * i] = t.v[0]; v[2 * i + 1] = t.v[1]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (int64_t)a0 + i < nnz ? Aval[a0 + i] : 0.0;
        }
        const int sh = 3 - cl;
        double a[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int e = j - sh;
            const double x = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : e == 3 ? v[3] : e == 4 ? v[4] : e == 5 ? v[5] : v[6];
            a[j] = (e >= 0 && e < len) ? x : 0.0;
        }
        typedef double v2dr __attribute__((ext_vector_type(2)));
        v2dr x; x.x = a[0]; x.y = a[1]; __builtin_nontemporal_store(x, reinterpret_cast<v2dr *>(p));
        x.x = a[2]; x.y = a[3]; __builtin_nontemporal_store(x, reinterpret_cast<v2dr *>(p) + 64);
        x.x = a[4]; x.y = a[5]; __builtin_nontemporal_store(x, reinterpret_cast<v2dr *>(p) + 128);
        // {a6 | w0, w1} as ONE 16-byte store: two 8-byte halves written at different times make every line of this
        // piece a partial write (read-modify-write in the memory system)
        v4i last; last.x = __double2loint(a[6]); last.y = __double2hiint(a[6]); last.z = w0; last.w = w1;
        __builtin_nontemporal_store(last, p + 192);
    }
    }
}

// ---------------------------------------------------------------------------------------------
// backward pass: the U sweep reads, in stored order, the entries right of the diagonal (SWEEP_BWD_FIRST_ASC)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
k_bwd_records(const int32_t *__restrict__ Aptr, const int32_t *__restrict__ Aidx, int64_t nnz, int32_t B, int32_t nb,
              const int32_t *__restrict__ start, const int32_t *__restrict__ blk2slot,
              const int32_t *__restrict__ wtab, const int32_t *__restrict__ skew, const int32_t *__restrict__ sfirst,
              const int32_t *__restrict__ scount, const int32_t *__restrict__ gtab, int32_t *__restrict__ exported,
              v4i *__restrict__ pkU, int32_t *__restrict__ flags)
{
    const int w = blockIdx.x;
    const int L = (threadIdx.x >> 6) * 8 + (threadIdx.x & 7);
    const int wg = w >> 2;
    const int slot = wg * kThreads + (w & 3) * 64 + L;
    for (int jj = 0; jj < kRecRows; ++jj) {
    const int k = (blockIdx.y * kRecRows + jj) * 8 + ((threadIdx.x >> 3) & 7);
    if (k >= scount[slot]) continue;
    const int r = sfirst[slot] - k;
    const int tau = k + skew[slot];
    const int base = wtab[(size_t)w * 4], c = tau - wtab[(size_t)w * 4 + 1];
    v4i rec; rec.x = kNoDep; rec.y = kNoDep; rec.z = kNoDep; rec.w = 0;
    {
        const int a0 = Aptr[r], a1 = Aptr[r + 1];
        const int len = a1 - a0;
        const Row8 own = load_row8(Aidx, a0, len > 8 ? 8 : len, nnz);
        int qd = 0;                                         // the diagonal (its presence was checked by the count pass)
#pragma unroll
        for (int i = 0; i < 8; ++i) qd += own.c[i] < r ? 1 : 0;
        const int nd = len - qd - 1;
        int bad = (nd > 3 || nd < 0 || len > 8) ? 1 : 0;
        int sd[3] = {kNoDep, kNoDep, kNoDep};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= nd) continue;
            const int kc = ROW8_AT(own, qd + 1 + j);
            const int b = block_of(kc, B, nb, start);
            const int oslot = blk2slot[b];
            const int kloc = start[b + 1] - 1 - kc;
            if (oslot == slot && kloc == k - 1) {
                sd[j] = kOwnPrev;
            } else if ((oslot >> 8) == wg) {
                if (kloc + skew[oslot] > tau) bad = 1;
                sd[j] = (oslot << 15) | kloc;
            } else {
                exported[oslot] = 1;
                if ((oslot >> 8) >= wg) bad = 1;
                int g = -1;
                if (kloc < 0x7fff) {
                    const int32_t *tab = gtab + (size_t)wg * kGhosts;
                    unsigned h = ((unsigned)oslot * 0x9E3779B1u) >> 26;
                    for (int probe = 0; probe < kGhosts; ++probe, h = (h + 1) & (kGhosts - 1)) {
                        const int cur = tab[h];
                        if (cur == oslot) { g = (int)h; break; }
                        if (cur == -1) break;
                    }
                }
                sd[j] = ((g >= 0 ? kGhostBase + g : oslot) << 15) | kloc;
            }
        }
        if (bad) atomicOr(&flags[0], 2);
        rec.x = sd[0]; rec.y = sd[1]; rec.z = sd[2]; rec.w = 1;
    }
    __builtin_nontemporal_store(rec, pkU + ((size_t)base + c) * 192 + L);
    }
}

// places of the chunks that hold no row: valid = 0 in every pattern word
__global__ void __launch_bounds__(512)
k_pad_records(const int32_t *__restrict__ wtab, const int32_t *__restrict__ skew, const int32_t *__restrict__ scount,
              v4i *__restrict__ pk, v4i *__restrict__ pkA)
{
    const int w = blockIdx.x;
    const int c = blockIdx.y * 8 + (threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const int base = wtab[(size_t)w * 4], tmin = wtab[(size_t)w * 4 + 1], nch = wtab[(size_t)w * 4 + 2];
    if (c >= nch) return;
    const int slot = (w >> 2) * kThreads + (w & 3) * 64 + L;
    const int k = tmin + c - skew[slot];
    if (k >= 0 && k < scount[slot]) return;
    v4i z; z.x = kNoDep; z.y = kNoDep; z.z = kNoDep; z.w = 0;
    pk[((size_t)base + c) * 192 + L] = z;
    if (pkA) {
        v4i *p = pkA + ((size_t)base + c) * 320 + L;
        v4i zz; zz.x = 0; zz.y = 0; zz.z = 0; zz.w = 0;
        reinterpret_cast<int *>(p + 192)[2] = 0;
        reinterpret_cast<int *>(p + 192)[3] = 0;
        p[256] = zz;
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// ghost tables, skews and chunk tables of one direction from A's triangle; launches only (no sync)
static void structure_launch(hipStream_t st, const DevMat &A, const Schedule &sch, int tri, PackedSweep *ps)
{
    const int nwg = sch.nslots / kThreads;
    ps->nwg = nwg;
    ps->kind = tri > 0 ? (int)SWEEP_FWD_LAST_ASC : (int)SWEEP_BWD_FIRST_ASC;
    ILUPP_HIP(pool_malloc(&ps->skew, sizeof(int32_t) * (size_t)sch.nslots));
    ILUPP_HIP(pool_malloc(&ps->wtab, sizeof(int32_t) * 16 * (size_t)nwg));
    ILUPP_HIP(pool_malloc(&ps->flags, 64));
    ILUPP_HIP(hipMemsetAsync(ps->flags, 0, 64, st));
    hipLaunchKernelGGL(k_ghost_table_tri, dim3((unsigned)nwg), dim3(kThreads), 0, st, A.ptr, A.idx, sch.B, sch.nb, sch.start,
                       sch.blk2slot, tri, sch.sfirst, sch.scount, sch.gtab);
    hipLaunchKernelGGL(k_lm_skew_tri, dim3((unsigned)nwg), dim3(kThreads), 0, st, A.ptr, A.idx, sch.B, sch.nb, sch.start,
                       sch.blk2slot, tri, sch.sfirst, sch.scount, ps->skew, ps->wtab, ps->flags);
    hipLaunchKernelGGL(k_lm_scan_a, dim3(1), dim3(kThreads), 0, st, nwg * 4, ps->wtab, ps->flags);
}

__global__ void k_flm_xrows_a(int32_t nslots, const int32_t *__restrict__ exported, const int32_t *__restrict__ scount,
                              int32_t *__restrict__ rows)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nslots) rows[s] = exported[s] ? scount[s] : 0;
}
__global__ void k_flm_xbase_a(int32_t nslots, const int32_t *__restrict__ exported, const int32_t *__restrict__ scount,
                              int32_t *__restrict__ xbase, long long *__restrict__ xcount)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int b = xbase[s];
    if (s == nslots - 1) *xcount = ((long long)b + (exported[s] ? scount[s] : 0)) * 4;
    if (!exported[s]) xbase[s] = -1;
}

// The whole level-major analysis of an ILU(0): returns true when the factor kernel and both sweeps can run
// level-major (pl, pu, f then complete up to the values); false leaves the three objects released.
bool lm_analyse_ilu0(hipStream_t st, const DevMat &A, const Schedule &fwd, const Schedule &bwd, PackedSweep *pl,
                     PackedSweep *pu, FactorLM *f)
{
    pl->release(); pu->release(); f->release();
    static const bool off = getenv("ILUPP_NO_PACKED") != nullptr ||
                            getenv("ILUPP_CLASSIC_ANALYSIS") != nullptr;
    static const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    // rows of at most 7 entries (checked per row by the passes; nnz <= 7n is the cheap necessary condition)
    if (off || A.nnz > 7 * (int64_t)A.n || A.nnz < 16 || fwd.nslots < kThreads || !fwd.gtab || !bwd.gtab ||
        fwd.nslots != bwd.nslots) return false;
    structure_launch(st, A, fwd, +1, pl);
    structure_launch(st, A, bwd, -1, pu);
    int32_t hl[4], hu[4];
    ILUPP_HIP(d2h_async(st, hl, pl->flags, sizeof(hl)));
    ILUPP_HIP(d2h_async(st, hu, pu->flags, sizeof(hu)));
    ILUPP_HIP(stream_sync(st));
    const int64_t lim = 2 * (int64_t)A.n + 64 * 4 * (int64_t)pl->nwg;
    if (hl[0] || hu[0] || hl[1] <= 0 || hu[1] <= 0 || (int64_t)hl[1] * 64 > lim || (int64_t)hu[1] * 64 > lim) {
        if (dbg) fprintf(stderr, "[ilupp] level-major analysis: structure rejected (%d %d, %d %d chunks)\n", hl[0], hu[0], hl[1], hu[1]);
        pl->release(); pu->release();
        return false;
    }
    pl->nchunks = hl[1]; pl->max_chunks = hl[2];
    pu->nchunks = hu[1]; pu->max_chunks = hu[2];
    ILUPP_HIP(pool_malloc(&pl->pk, (size_t)pl->nchunks * 3072));
    ILUPP_HIP(pool_malloc(&pu->pk, (size_t)pu->nchunks * 3072));
    ILUPP_HIP(pool_malloc(&f->pkA, (size_t)pl->nchunks * 5120));
    pl->built = pu->built = true;
    {
        const dim3 grid((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 7) / 8));
        hipLaunchKernelGGL(k_pad_records, grid, dim3(512), 0, st, pl->wtab, pl->skew, fwd.scount, reinterpret_cast<v4i *>(pl->pk),
                           reinterpret_cast<v4i *>(f->pkA));
        const dim3 gridr((unsigned)(pl->nwg * 4), (unsigned)((pl->max_chunks + 8 * kRecRows - 1) / (8 * kRecRows)));   // every lane has at most max_chunks rows
        hipLaunchKernelGGL(k_fwd_records, gridr, dim3(512), 0, st, A.ptr, A.idx, A.val, (int64_t)A.nnz, fwd.B, fwd.nb, fwd.start, fwd.blk2slot, pl->wtab,
                           pl->skew, fwd.sfirst, fwd.scount, fwd.gtab, fwd.exported, reinterpret_cast<v4i *>(pl->pk),
                           reinterpret_cast<v4i *>(f->pkA), pl->flags);
    }
    {
        const dim3 grid((unsigned)(pu->nwg * 4), (unsigned)((pu->max_chunks + 7) / 8));
        hipLaunchKernelGGL(k_pad_records, grid, dim3(512), 0, st, pu->wtab, pu->skew, bwd.scount, reinterpret_cast<v4i *>(pu->pk),
                           static_cast<v4i *>(nullptr));
        const dim3 gridr((unsigned)(pu->nwg * 4), (unsigned)((pu->max_chunks + 8 * kRecRows - 1) / (8 * kRecRows)));
        hipLaunchKernelGGL(k_bwd_records, gridr, dim3(512), 0, st, A.ptr, A.idx, (int64_t)A.nnz, bwd.B, bwd.nb, bwd.start, bwd.blk2slot, pu->wtab,
                           pu->skew, bwd.sfirst, bwd.scount, bwd.gtab, bwd.exported, reinterpret_cast<v4i *>(pu->pk), pu->flags);
    }
    lm_link_factor(st, fwd, bwd, pu);
    lm_link_y(st, fwd, pl, pu);
    // exchange rows of the exported forward slots
    const int nslots = fwd.nslots;
    ILUPP_HIP(pool_malloc(&f->xbase, sizeof(int32_t) * (size_t)nslots));
    ILUPP_HIP(pool_malloc(&f->xcount, 64));
    int32_t *rows = nullptr;
    ILUPP_HIP(pool_malloc(&rows, sizeof(int32_t) * (size_t)nslots));
    const unsigned gb = (unsigned)((nslots + 255) / 256);
    hipLaunchKernelGGL(k_flm_xrows_a, dim3(gb), dim3(256), 0, st, nslots, fwd.exported, fwd.scount, rows);
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, rows, f->xbase, nslots, st));
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&tmp, tb));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, rows, f->xbase, nslots, st));
    hipLaunchKernelGGL(k_flm_xbase_a, dim3(gb), dim3(256), 0, st, nslots, fwd.exported, fwd.scount, f->xbase, f->xcount);
    ILUPP_HIP(pool_malloc(&f->xch, sizeof(double) * 4 * (size_t)A.n + 64));
    int32_t gl[8], gu[8];
    ILUPP_HIP(d2h_async(st, gl, pl->flags, sizeof(gl)));
    ILUPP_HIP(d2h_async(st, gu, pu->flags, sizeof(gu)));
    ILUPP_HIP(stream_sync(st));
    ILUPP_HIP(pool_free(rows));
    ILUPP_HIP(pool_free(tmp));
    if (dbg) fprintf(stderr, "[ilupp] level-major analysis: flags fwd %d/%d bwd %d link %d, %d+%d chunks\n", gl[0], gl[4], gu[0], gu[3], hl[1], hu[1]);
    if (gl[0] || gl[4] || gu[0] || gu[3]) { pl->release(); pu->release(); f->release(); return false; }
    pl->valid = pu->valid = true;
    pu->linked = true;
    f->built = true;
    f->values_packed = A.val != nullptr;
    return true;
}

}  // namespace ilupp
