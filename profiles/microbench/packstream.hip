// Microbenchmark: level-major packed records (wave-coalesced 3 x 16 B per lane and step) + lane-private rhs read / out write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ long line_of(int wg, int t) {
    const int ty = wg & 15, tz = wg >> 4, ly = t & 15, lz = t >> 4;
    return (long)(tz * 16 + lz) * 256 + ty * 16 + ly;
}

// packed: per (wg, wave, step): [desc 64x16B][val01 64x16B][val23 64x16B]
template <int RHS, int WR, int U>
__global__ __launch_bounds__(256) void k_rows(const int4 *__restrict__ pk, const double *__restrict__ rhs, double *__restrict__ out, int rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long line = line_of(blockIdx.x, threadIdx.x);
    const int4 *p = pk + ((long)(blockIdx.x * 4 + wave) * rows) * 192 + lane;
    const double *b = rhs + line * rows;
    double *o = out + line * rows;
    double acc = 0.0;
    for (int r = 0; r < rows; r += U) {
        int4 d[U], v0[U], v1[U]; double bb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = p[(long)(r + u) * 192]; v0[u] = p[(long)(r + u) * 192 + 64]; v1[u] = p[(long)(r + u) * 192 + 128];
            bb[u] = RHS ? b[r + u] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc += bb[u] + (double)d[u].x * v0[u].x + (double)d[u].y * v0[u].z + (double)d[u].z * v1[u].x + (double)d[u].w * v1[u].z;
            if (WR) o[r + u] = acc;
        }
    }
    if (!WR && acc == 1.2345) o[0] = acc;
}

int main() {
    const int rows = 256; const long lines = 65536;
    int4 *pk; double *rhs, *out;
    CK(hipMalloc(&pk, lines * rows * 48)); CK(hipMalloc(&rhs, lines * rows * 8)); CK(hipMalloc(&out, lines * rows * 8));
    CK(hipMemset(pk, 0, lines * rows * 48)); CK(hipMemset(rhs, 0, lines * rows * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, double bytes_per_row) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int it = 0; it < 3; ++it) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("%-34s %.3f ms  %.1f GB/s  (%.2f us per row-step)\n", name, ms, lines * rows * bytes_per_row / ms * 1e-6, ms * 1e3 / rows);
    };
    run("packed only, U=1", [&] { k_rows<0, 0, 1><<<256, 256>>>(pk, rhs, out, rows); }, 48);
    run("packed only, U=2", [&] { k_rows<0, 0, 2><<<256, 256>>>(pk, rhs, out, rows); }, 48);
    run("packed only, U=4", [&] { k_rows<0, 0, 4><<<256, 256>>>(pk, rhs, out, rows); }, 48);
    run("packed + rhs, U=2", [&] { k_rows<1, 0, 2><<<256, 256>>>(pk, rhs, out, rows); }, 56);
    run("packed + rhs, U=4", [&] { k_rows<1, 0, 4><<<256, 256>>>(pk, rhs, out, rows); }, 56);
    run("packed + rhs + out, U=1", [&] { k_rows<1, 1, 1><<<256, 256>>>(pk, rhs, out, rows); }, 64);
    run("packed + rhs + out, U=2", [&] { k_rows<1, 1, 2><<<256, 256>>>(pk, rhs, out, rows); }, 64);
    run("packed + rhs + out, U=4", [&] { k_rows<1, 1, 4><<<256, 256>>>(pk, rhs, out, rows); }, 64);
    run("packed + out, U=4", [&] { k_rows<0, 1, 4><<<256, 256>>>(pk, rhs, out, rows); }, 56);
    return 0;
}
