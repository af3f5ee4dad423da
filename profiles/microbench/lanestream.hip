// Microbenchmark: bandwidth of "one private contiguous stream per lane" versus wave-coalesced streaming.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// each lane reads `per_lane` bytes from its own region, 16 B per load, U loads in flight
template <int U>
__global__ __launch_bounds__(256) void k_lane(const int4 *__restrict__ src, long per_lane16, int4 *__restrict__ sink) {
    long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int4 *p = src + lane * per_lane16;
    int4 acc = {0, 0, 0, 0};
    for (long k = 0; k < per_lane16; k += U) {
        int4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[k + u];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
    }
    if (acc.x == 0x12345678) sink[lane] = acc;
}

// wave-interleaved: lane t of wave w reads element (k*64 + t) of the wave's region
template <int U>
__global__ __launch_bounds__(256) void k_wave(const int4 *__restrict__ src, long per_lane16, int4 *__restrict__ sink) {
    long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long wave = lane >> 6; int t = lane & 63;
    const int4 *p = src + wave * per_lane16 * 64 + t;
    int4 acc = {0, 0, 0, 0};
    for (long k = 0; k < per_lane16; k += U) {
        int4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(k + u) * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
    }
    if (acc.x == 0x12345678) sink[lane] = acc;
}

// lane-private stream but every lane fetches whole 128-B lines (8 consecutive 16-B loads = one line) per step
template <int U>
__global__ __launch_bounds__(256) void k_lane_store(int4 *__restrict__ dst, long per_lane16) {
    long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int4 *p = dst + lane * per_lane16;
    int4 v = {1, 2, 3, (int)lane};
    for (long k = 0; k < per_lane16; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) p[k + u] = v;
    }
}

int main() {
    const long lanes = 256L * 256;          // 256 workgroups x 256 lanes
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int4 *sink; CK(hipMalloc(&sink, lanes * 4 * 16));
    for (long per_lane : {4096L, 14336L, 32768L}) {
        long bytes = lanes * per_lane;
        int4 *src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
        long p16 = per_lane / 16;
        auto run = [&](const char *name, auto launch) {
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            printf("per_lane %6ld B  %-18s %8.3f ms  %8.1f GB/s\n", per_lane, name, ms, bytes / ms * 1e-6);
        };
        run("lane U=1", [&] { k_lane<1><<<256, 256>>>(src, p16, sink); });
        run("lane U=4", [&] { k_lane<4><<<256, 256>>>(src, p16, sink); });
        run("lane U=8", [&] { k_lane<8><<<256, 256>>>(src, p16, sink); });
        run("lane U=16", [&] { k_lane<16><<<256, 256>>>(src, p16, sink); });
        run("lane U=8 x2occ", [&] { k_lane<8><<<512, 256>>>(src, p16 / 2, sink); });
        run("lane U=8 x4occ", [&] { k_lane<8><<<1024, 256>>>(src, p16 / 4, sink); });
        run("wave U=1", [&] { k_wave<1><<<256, 256>>>(src, p16, sink); });
        run("wave U=4", [&] { k_wave<4><<<256, 256>>>(src, p16, sink); });
        run("wave U=8", [&] { k_wave<8><<<256, 256>>>(src, p16, sink); });
        run("wave U=8 x4occ", [&] { k_wave<8><<<1024, 256>>>(src, p16 / 4, sink); });
        run("lane store U=4", [&] { k_lane_store<4><<<256, 256>>>(src, p16); });
        run("lane store U=8x4", [&] { k_lane_store<8><<<1024, 256>>>(src, p16 / 4); });
        CK(hipFree(src));
    }
    return 0;
}
