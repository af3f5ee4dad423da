// Microbenchmark: how fast can one 16x16 tile of grid lines run the dependency levels of a triangular sweep when the
// hand-off inside a wave goes through registers (shuffle) instead of tagged LDS ring entries, and the records are read
// directly (level-major, software-prefetched) instead of through loader waves and an LDS chunk ring?
//
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tilesolve.hip -o tilesolve && ./tilesolve
//
// Model of the L-sweep of a 7-point ILU(0) factor: lane (y, z) of the tile owns the x-line (., y, z) of `steps` rows and
// solves row k at step tau = k + y + z:   x = (rhs - v0 * x[z-1] - v1 * x[y-1] - v2 * x[own previous row]) / vd
// (sequential accumulation, separate multiply and subtract, true division -- the arithmetic of sptrsv_lm.hip).
// Wave w holds the z-rows 4w..4w+3 (64 lanes = 4 x 16): the y-1 neighbour is lane-1, the z-1 neighbour lane-16 of the same
// wave, except for the first z-row of a wave, which takes it from the previous wave through an LDS array guarded by that
// wave's step counter.  Tiles are independent here (no cross-workgroup hand-off): the number that matters is the time per
// level of ONE tile and of 256 tiles running side by side, next to sptrsv_lm's 0.86 us (alone) / 1.35 us (all tiles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
static constexpr int kDepth = 8;     // steps a wave may run ahead of the next one (hand-off ring depth)

// records: per (wave, step) 64 x {v0, v1} then 64 x {v2, vd} then 64 x rhs  (2 KB + 512 B), level-major
__global__ __launch_bounds__(256) void k_tilesolve(const double *__restrict__ rec, double *__restrict__ out, int steps, int nlev)
{
    __shared__ double xs[4][kDepth][16];          // last z-row of wave w at step tau -> first z-row of wave w+1
    __shared__ int wstep[4];                      // last level a wave has finished
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int y = lane & 15, zl = lane >> 4, z = 4 * w + zl;
    const int skew = y + z;
    if (lane == 0) wstep[w] = -1;
    __syncthreads();
    const size_t wave = (size_t)blockIdx.x * 4 + w;
    const double *p = rec + wave * (size_t)nlev * 320;             // 320 doubles per (wave, level)
    double *o = out + wave * (size_t)nlev * 64 + lane;
    double prev = 0.0;
    // prefetch ring of 2 levels
    v2d a0 = *reinterpret_cast<const v2d *>(p + 2 * lane), b0 = *reinterpret_cast<const v2d *>(p + 128 + 2 * lane);
    double r0 = p[256 + lane];
    for (int tau = 0; tau < nlev; ++tau) {
        // next level's record in flight while this one is computed
        const double *pn = p + (size_t)(tau + 1 < nlev ? tau + 1 : tau) * 320;
        const v2d a1 = *reinterpret_cast<const v2d *>(pn + 2 * lane), b1 = *reinterpret_cast<const v2d *>(pn + 128 + 2 * lane);
        const double r1 = pn[256 + lane];
        const int k = tau - skew;
        const bool valid = k >= 0 && k < steps;
        // neighbours of the previous level
        double xy = __shfl_up(prev, 1);
        double xz = __shfl_up(prev, 16);
        if (zl == 0 && w > 0) {
            // first z-row of the wave: from the previous wave, which must have finished level tau-1
            while (true) {
                asm volatile("" ::: "memory");
                if (wstep[w - 1] >= tau - 1) break;
                __builtin_amdgcn_s_sleep(0);
            }
            asm volatile("" ::: "memory");
            xz = xs[w - 1][(tau - 1) & (kDepth - 1)][y];
        }
        if (w < 3) {
            // do not lap the reader of our hand-off ring
            while (true) {
                asm volatile("" ::: "memory");
                if (wstep[w + 1] >= tau - (kDepth - 2)) break;
                __builtin_amdgcn_s_sleep(0);
            }
        }
        double x = prev;
        if (valid) {
            double acc = r0;
            if (z > 0) { const double pr = a0.x * xz; acc = acc - pr; }
            if (y > 0) { const double pr = a0.y * xy; acc = acc - pr; }
            if (k > 0) { const double pr = b0.x * prev; acc = acc - pr; }
#ifdef NO_DIV
            x = acc * b0.y;          // (experiment: the share of the division in a level)
#else
            x = acc / b0.y;
#endif
        }
        prev = x;
#ifndef NO_STORE
        __builtin_nontemporal_store(x, o + (size_t)tau * 64);
#endif
        if (zl == 3) xs[w][tau & (kDepth - 1)][y] = x;
        asm volatile("" ::: "memory");
        if (lane == 0) wstep[w] = tau;
        a0 = a1; b0 = b1; r0 = r1;
    }
}

int main()
{
    const int steps = 256, nlev = steps + 30;
    for (int nwg : {1, 256}) {
        const size_t waves = (size_t)nwg * 4;
        const size_t nrec = waves * nlev * 320, nout = waves * nlev * 64;
        std::vector<double> h(nrec);
        for (size_t i = 0; i < nrec; ++i) h[i] = ((i % 320) >= 256) ? 1.0 : (((i % 320) & 1) && (i % 320) >= 128 ? 6.0 : -1.0);
        double *rec, *out;
        CK(hipMalloc(&rec, nrec * 8)); CK(hipMalloc(&out, nout * 8));
        CK(hipMemcpy(rec, h.data(), nrec * 8, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_tilesolve, dim3(nwg), dim3(256), 0, 0, rec, out, steps, nlev);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<double> ho(64);
        CK(hipMemcpy(ho.data(), out + (size_t)(nlev - 1) * 64, 64 * 8, hipMemcpyDeviceToHost));
        printf("%3d tiles: %.3f ms for %d levels = %.3f us per level   (x[last] = %.6f)\n", nwg, best, nlev, 1e3 * best / nlev, ho[0]);
        CK(hipFree(rec)); CK(hipFree(out));
    }
    return 0;
}
