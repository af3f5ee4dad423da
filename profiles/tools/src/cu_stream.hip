// What ONE compute unit can stream from HBM, as a function of what it keeps in flight (round 6; the question behind DESIGN.md §4.0:
// "a tile at work is bounded by what its CU has in flight at HBM latency").  Every workgroup owns a private region of a large buffer
// and reads it once, front to back, W waves side by side, each with K vector-memory instructions of 1 KB (64 lanes x 16 bytes) in
// flight; 140 KB of dynamic LDS keep it at one workgroup per CU.  Three ways of loading: into registers (global_load_dwordx4), by
// LDS-DMA (buffer_load_dwordx4 ... lds, what k_ilu0_wa's loaders issue), and registers with every second instruction a write-through
// store of 1 KB (sc1, what its consumers issue).
//   hipcc -O3 --offload-arch=gfx950 profiles/tools/src/cu_stream.hip -o profiles/tools/cu_stream && profiles/tools/cu_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int K>
__global__ __launch_bounds__(1024) void k_stream(const char *buf, char *out, size_t per_wg, unsigned long long *sink)
{
    extern __shared__ char lds[];
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char *base = buf + (size_t)blockIdx.x * per_wg;
    char *obase = out + (size_t)blockIdx.x * per_wg;
    const size_t per_wave = per_wg / nw;
    const int iters = (int)(per_wave / (1024 * K));
    u32x4 acc = {0, 0, 0, 0};
    if (MODE >= 5) {
        // the pattern of k_ilu0_wa's loaders: a wave looks after 64 x-lines (here: 64 pieces of its region, one after the other); a
        // block is two steps = 112 bytes of every line, fetched as the 128-byte window around them (16-byte aligned) by 8 instructions
        // of 8 lines x 8 pieces of 16 bytes; three blocks in flight
        const size_t linelen = (per_wave / 64) & ~(size_t)127;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base + (size_t)wv * per_wave), 0, (int)per_wave, 0x00020000);
        const unsigned adv = (MODE == 6 || MODE >= 9) ? 128u : 112u;   // MODE 6: whole, aligned lines (every byte new)
        const int nblk = (int)((linelen - 128) / adv);
        unsigned g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = (unsigned)((8 * q + (ln >> 3)) * linelen) + 16u * (MODE == 9 ? ((ln & 7) ^ (ln >> 3) ^ (q & 1)) : MODE == 10 ? ((ln + (ln >> 3)) & 7) : (ln & 7));
        if (MODE == 11) {
            // half lines: an instruction is 16 x-lines x 64 bytes (aligned), four instructions per round of 64 bytes per x-line
            const int nrnd = (int)((linelen - 128) / 64);
            for (int r = 0; r < nrnd; ++r) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned src = (unsigned)((16 * q + (ln >> 2)) * linelen) + 64u * (unsigned)r + 16u * (ln & 3);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + wv * (4 * 8192) + (r & 7) * 4096 + q * 1024), 16, src, 0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
        } else if (MODE == 8) {
            // a ring of four lines PER x-line, contiguous (512 bytes): an instruction is two x-lines x four slots, of which the one
            // slot of the round is fetched (16 of 64 threads active); 32 instructions per round
            const int nrnd = (int)((linelen - 128) / 128);
            const unsigned gl = (unsigned)((ln >> 5)) , sl = (unsigned)((ln >> 3) & 3);
            for (int r = 0; r < nrnd; ++r) {
                const bool on = sl == (unsigned)(r & 3);
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const unsigned src = (unsigned)((2 * q + gl) * linelen) + 128u * (unsigned)r + 16u * (ln & 7);
                    if (on) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + wv * (4 * 8192) + q * 1024), 16, src, 0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            }
        } else if (MODE == 7) {
            // two blocks at a time, the two windows of a line right behind each other
            for (int b = 0; b + 1 < nblk; b += 2) {
                const unsigned off0 = (adv * (unsigned)b) & ~15u, off1 = (adv * (unsigned)(b + 1)) & ~15u;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + wv * (4 * 8192) + (b & 3) * 8192 + q * 1024), 16, g[q] + off0, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + wv * (4 * 8192) + ((b + 1) & 3) * 8192 + q * 1024), 16, g[q] + off1, 0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
        } else
        for (int b = 0; b < nblk; ++b) {
            const unsigned off = (adv * (unsigned)b) & ~15u;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + wv * (4 * 8192) + (b & 3) * 8192 + q * 1024), 16, g[q] + off, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x = *(unsigned *)(lds + wv * 4 * 8192 + 16 * ln);
    } else if (MODE == 1) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base + (size_t)wv * per_wave), 0, (int)per_wave, 0x00020000);
        unsigned g = 16u * ln;
        // 2K instructions: K in flight while the K before them are waited for
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < K; ++q) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + wv * (2 * K * 1024) + ((i & 1) * K + q) * 1024), 16, g, 0, 0, 0);
                g += 1024;
            }
            if (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (K == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (K == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (K == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x = *(unsigned *)(lds + wv * 2 * K * 1024 + 16 * ln);
    } else {
        const u32x4 *p = (const u32x4 *)(base + (size_t)wv * per_wave) + ln;
        u32x4 *o = (u32x4 *)(obase + (size_t)wv * per_wave) + ln;
        u32x4 r[K];
#pragma unroll
        for (int q = 0; q < K; ++q) r[q] = __builtin_nontemporal_load(p + 64 * q);
        for (int i = 1; i < iters; ++i) {
            p += 64 * K;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                acc ^= r[q];
                if (MODE == 2) {
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o + 64 * q), "v"(r[q]) : "memory");
                }
                if (MODE == 3) __builtin_nontemporal_store(r[q], o + 64 * q);
                if (MODE == 4) o[64 * q] = r[q];
                r[q] = __builtin_nontemporal_load(p + 64 * q);
            }
            o += 64 * K;
        }
#pragma unroll
        for (int q = 0; q < K; ++q) acc ^= r[q];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int K>
static double run(const char *buf, char *out, size_t per_wg, int nwg, int waves, unsigned long long *sink)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void *)k_stream<MODE, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_stream<MODE, K>), dim3(nwg), dim3(64 * waves), 140 * 1024, 0, buf, out, per_wg, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const size_t per_wave = per_wg / waves;
    double bytes = (double)(per_wave / (1024 * K)) * 1024 * K * waves;
    if (MODE >= 5) { const size_t linelen = (per_wave / 64) & ~(size_t)127; bytes = (double)((linelen - 128) / 112) * 112 * 64 * waves; }
    return bytes / (ms * 1e-3) / 1e9;   // GB/s per workgroup (loads; MODE 2 stores as many again)
}

int main()
{
    const size_t per_wg = 8u << 20;
    const int maxwg = 256;
    char *buf, *out;
    unsigned long long *sink;
    CK(hipMalloc(&buf, per_wg * maxwg)); CK(hipMalloc(&out, per_wg * maxwg)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 1, per_wg * maxwg)); CK(hipMemset(out, 0, per_wg * maxwg));
    char *flush; const size_t fl = 1u << 30;
    CK(hipMalloc(&flush, fl));
    // (code objects loaded before anything is timed)
    run<0, 4>(buf, out, 1u << 20, 1, 1, sink); run<0, 16>(buf, out, 1u << 20, 1, 1, sink); run<1, 4>(buf, out, 1u << 20, 1, 1, sink);
    run<2, 4>(buf, out, 1u << 20, 1, 1, sink); run<2, 16>(buf, out, 1u << 20, 1, 1, sink);
    const int wgs[4] = {1, 16, 128, 256};
    const int wvs[5] = {1, 2, 4, 8, 16};
    const char *names[12] = {"registers", "LDS-DMA", "registers + sc1 stores", "registers + nt stores", "registers + plain stores",
                            "LDS-DMA, the factor kernel's windows (8 lines x 128 bytes per instruction, 112 of them new; 16..24 in flight; GB/s of NEW bytes)",
                            "LDS-DMA, 8 lines x 128 bytes per instruction, ALIGNED (all 128 new)",
                            "LDS-DMA, the factor kernel's windows, two blocks at a time (a line's two windows behind each other)",
                            "LDS-DMA, aligned lines, 2 x-lines x 4 slots per instruction of which one slot is fetched (16 threads active, 32 instructions per round)",
                            "LDS-DMA, aligned lines, the 8 pieces of a line XOR-permuted among its 8 threads",
                            "LDS-DMA, aligned lines, the 8 pieces of a line rotated among its 8 threads",
                            "LDS-DMA, aligned HALF lines: 16 x-lines x 64 bytes per instruction"};
    printf("GB/s of loads PER WORKGROUP (= per CU; 8 MiB each, read once, cold); rows: workgroups in the launch; columns: waves per workgroup\n");
    run<3, 4>(buf, out, 1u << 20, 1, 1, sink); run<3, 16>(buf, out, 1u << 20, 1, 1, sink); run<4, 4>(buf, out, 1u << 20, 1, 1, sink);
    run<4, 16>(buf, out, 1u << 20, 1, 1, sink); run<5, 4>(buf, out, 1u << 20, 1, 1, sink); run<6, 4>(buf, out, 1u << 20, 1, 1, sink); run<7, 4>(buf, out, 1u << 20, 1, 1, sink); run<8, 4>(buf, out, 1u << 20, 1, 1, sink); run<9, 4>(buf, out, 1u << 20, 1, 1, sink); run<10, 4>(buf, out, 1u << 20, 1, 1, sink); run<11, 4>(buf, out, 1u << 20, 1, 1, sink);
    for (int mode = (getenv("CU_FROM") ? atoi(getenv("CU_FROM")) : 0); mode < 12; ++mode)
        for (int K : {4, 16}) {
            if ((mode == 1 || mode >= 5) && K == 16) continue;   // 16 waves x 2K KB of LDS
            printf("%s, %d instructions of 1 KB in flight per wave\n        ", names[mode], K);
            for (int w : wvs) printf("%8d", w);
            printf("\n");
            for (int nwg : wgs) {
                printf("%6d  ", nwg);
                for (int w : wvs) {
                    if (mode == 1 && w * 2 * K > 136) { printf("%8s", "-"); continue; }
                    if (mode >= 5 && w * 32 > 136) { printf("%8s", "-"); continue; }
                    CK(hipMemset(flush, 0, fl));   // the Infinity Cache holds 256 MB
                    CK(hipDeviceSynchronize());
                    double g = 0;
                    if (mode == 0 && K == 4) g = run<0, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 0 && K == 16) g = run<0, 16>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 1 && K == 4) g = run<1, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 2 && K == 4) g = run<2, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 2 && K == 16) g = run<2, 16>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 3 && K == 4) g = run<3, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 3 && K == 16) g = run<3, 16>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 4 && K == 4) g = run<4, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 4 && K == 16) g = run<4, 16>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 5) g = run<5, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 6) g = run<6, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 7) g = run<7, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 8) g = run<8, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 9) g = run<9, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 10) g = run<10, 4>(buf, out, per_wg, nwg, w, sink);
                    if (mode == 11) g = run<11, 4>(buf, out, per_wg, nwg, w, sink);
                    printf("%8.1f", g);
                }
                printf("\n");
            }
        }
    return 0;
}
