// store_rate.hip -- how fast can ONE CU issue 16-byte-per-lane stores (the record stores of the static factor kernel)?
// hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate && ./store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: nt stores, 1: plain stores, 2: 16-byte loads (nt), 3: loads + stores
__global__ void k(v2d *p, const v2d *q, int iters, unsigned long long *ticks, double *sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, W = blockDim.x >> 6;
    v2d *o = p + ((size_t)blockIdx.x * iters * W) * 64;
    const v2d *in = q + ((size_t)blockIdx.x * iters * W) * 64;
    v2d v; v.x = lane; v.y = wave;
    v2d acc; acc.x = acc.y = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const size_t at = ((size_t)it * W + wave) * 64 + lane;
        if (MODE == 0) __builtin_nontemporal_store(v, o + at);
        if (MODE == 1) o[at] = v;
        if (MODE == 2 || MODE == 3) { const v2d x = __builtin_nontemporal_load(in + at); acc += x; }
        if (MODE == 3) __builtin_nontemporal_store(v, o + at);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (acc.x == 1.2345) sink[0] = acc.y;
}

int main()
{
    const int iters = 2048;
    const size_t bytes = (size_t)256 * iters * 8 * 1024 + 4096;
    v2d *p, *q; unsigned long long *t; double *sink;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&t, 8 * 256)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(q, 0, bytes));
    unsigned long long h[256];
    for (int mode = 0; mode < 4; ++mode)
        for (int waves = 1; waves <= 8; waves *= 2)
            for (int grid = 1; grid <= 256; grid *= 16) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(waves * 64), 0, 0, p, q, iters, t, sink);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(waves * 64), 0, 0, p, q, iters, t, sink);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(waves * 64), 0, 0, p, q, iters, t, sink);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(waves * 64), 0, 0, p, q, iters, t, sink);
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(h, t, 8 * grid, hipMemcpyDeviceToHost));
                double mx = 0; for (int i = 0; i < grid; ++i) mx = h[i] > mx ? h[i] : mx;
                const double ns = mx * 10.0;                    // 100 MHz
                const double kb = (double)iters * waves * (mode == 3 ? 2.0 : 1.0);
                printf("mode %d (%s) waves/WG %d WGs %3d: %.1f ns per 1 KiB wave-instruction per wave, %.1f GB/s per CU, %.2f TB/s chip\n", mode,
                       mode == 0 ? "nt store" : mode == 1 ? "store" : mode == 2 ? "nt load" : "load+store", waves, grid, ns / iters, kb * 1024 / ns,
                       kb * 1024 / ns * grid / 1000);
            }
    return 0;
}
