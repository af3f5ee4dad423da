// BiCGstab with SPLIT preconditioning, every vector in HBM -- the iteration behind ilupp_hip_solve (reference: _ilupp.solve,
// binding.cpp:200-230 -> solve_with_multilevel_preconditioner, solving_routines_implementation.h:81 -> bicgstab,
// iterative_solvers_implementation.h:385-530, started from the zero vector).
//
// The operator (v -> L'(A(R' v))) and the right part (y -> R' y) are handed in by the caller; this file owns the vector work.  All
// scalars of the recurrence (alpha, omega, beta, the inner products) stay in device memory and the update kernels read them there;
// the only thing the host sees per iteration is the residual norm the loop condition needs.  The inner products are two-stage
// reductions with a fixed shape (contiguous chunk per workgroup, fixed tree inside, fixed order over the workgroups): the same input
// gives the same bits on every run.  HBM-bound streaming work: 8-byte coalesced accesses, one pass per update.
#include "common.h"

#include <cmath>

namespace ilupp {

namespace {

constexpr int kDotThreads = 256;
constexpr int kDotMaxBlocks = 1024;

struct BicgScalars { double rho, alpha, omega, beta, res, pad[3]; };

__device__ __forceinline__ double block_sum(double v, double *sh)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = kDotThreads / 2; s > 0; s >>= 1) {
        if (t < s) sh[t] = sh[t] + sh[t + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// partial[b] = sum over the block's chunk of a0*b0, partial[nb + b] = ... of a1*b1 (a pair with a null first vector is skipped)
__global__ __launch_bounds__(kDotThreads) void k_bicg_dots(int32_t n, int32_t chunk, const double *__restrict__ a0, const double *__restrict__ b0,
                                                           const double *__restrict__ a1, const double *__restrict__ b1, double *__restrict__ partial)
{
    __shared__ double sh[kDotThreads];
    const int32_t lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    double s0 = 0.0, s1 = 0.0;
    for (int32_t i = lo + threadIdx.x; i < hi; i += kDotThreads) {
        if (a0) s0 = s0 + a0[i] * b0[i];
        if (a1) s1 = s1 + a1[i] * b1[i];
    }
    const double t0 = block_sum(s0, sh), t1 = block_sum(s1, sh);
    if (threadIdx.x == 0) { partial[blockIdx.x] = t0; partial[gridDim.x + blockIdx.x] = t1; }
}

// one workgroup: the partial sums in their order, then the scalar of the recurrence this stage produces
//   mode 0: alpha = rho / (Ap, r0*)          mode 1: omega = (As, s) / (As, As)
//   mode 2: beta = ((r, r0*) / rho) * (alpha / omega); rho = (r, r0*); res = ||r||      mode 3: rho = (r, r0*), res = ||r|| (the start)
__global__ __launch_bounds__(kDotThreads) void k_bicg_scalars(int32_t nb, const double *__restrict__ partial, BicgScalars *sc, int mode)
{
    __shared__ double sh[kDotThreads];
    double s0 = 0.0, s1 = 0.0;
    for (int32_t i = threadIdx.x; i < nb; i += kDotThreads) { s0 = s0 + partial[i]; s1 = s1 + partial[nb + i]; }
    const double d0 = block_sum(s0, sh), d1 = block_sum(s1, sh);
    if (threadIdx.x != 0) return;
    if (mode == 0) sc->alpha = sc->rho / d1;
    else if (mode == 1) sc->omega = d0 / d1;
    else if (mode == 2) { sc->beta = (d0 / sc->rho) * (sc->alpha / sc->omega); sc->rho = d0; sc->res = sqrt(d1); }
    else { sc->rho = d0; sc->res = sqrt(d1); }
}

// s = r - alpha * Ap (scaled_vector_addition(r, -alpha, Ap))
__global__ void k_bicg_s(int32_t n, const double *__restrict__ r, const double *__restrict__ Ap, const BicgScalars *__restrict__ sc, double *__restrict__ s)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ma = -sc->alpha;
    s[i] = r[i] + ma * Ap[i];
}

// y += alpha p; y += omega s; r = s - omega As -- and the chunk's parts of (r, r0*) and (r, r) on the way
__global__ __launch_bounds__(kDotThreads) void k_bicg_update(int32_t n, int32_t chunk, double *__restrict__ y, const double *__restrict__ p,
                                                             const double *__restrict__ s, const double *__restrict__ As, double *__restrict__ r,
                                                             const double *__restrict__ r0s, const BicgScalars *__restrict__ sc, double *__restrict__ partial)
{
    __shared__ double sh[kDotThreads];
    const double alpha = sc->alpha, omega = sc->omega, mo = -omega;
    const int32_t lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    double s0 = 0.0, s1 = 0.0;
    for (int32_t i = lo + threadIdx.x; i < hi; i += kDotThreads) {
        double yi = y[i] + alpha * p[i];
        yi = yi + omega * s[i];
        y[i] = yi;
        const double ri = s[i] + mo * As[i];
        r[i] = ri;
        s0 = s0 + ri * r0s[i];
        s1 = s1 + ri * ri;
    }
    const double t0 = block_sum(s0, sh), t1 = block_sum(s1, sh);
    if (threadIdx.x == 0) { partial[blockIdx.x] = t0; partial[gridDim.x + blockIdx.x] = t1; }
}

// p = p - omega Ap; p = beta p + r
__global__ void k_bicg_p(int32_t n, double *__restrict__ p, const double *__restrict__ Ap, const double *__restrict__ r, const BicgScalars *__restrict__ sc)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mo = -sc->omega, beta = sc->beta;
    const double pi = p[i] + mo * Ap[i];
    p[i] = beta * pi + r[i];
}

struct HostPin {
    double *p = nullptr;
    ~HostPin() { if (p) (void)hipHostFree(p); }
};

}  // namespace

// r: on entry L' b, on exit the last residual; y: the result in the preconditioned variable (the caller applies R').
// op(in, out): out = L'(A(R' in)), `in` kept.  Returns ILUPP_OK with *it / *rel / *res as bicgstab leaves them (:496-499).
int bicgstab_split(hipStream_t st, int32_t n, const std::function<int(const double *, double *)> &op, double *r, double *y, int32_t min_iter,
                   int32_t max_iter, double rtol, double atol, int32_t *it_out, double *rel_out, double *res_out)
{
    const int nb = std::max(1, std::min(kDotMaxBlocks, (n + kDotThreads - 1) / kDotThreads));
    const int32_t chunk = (int32_t)(((int64_t)n + nb - 1) / nb);
    const int gb = (n + 255) / 256;
    PoolBlock b_vec, b_part, b_sc;
    ILUPP_HIP(b_vec.alloc(sizeof(double) * (size_t)n * 5));
    ILUPP_HIP(b_part.alloc(sizeof(double) * (size_t)nb * 2));
    ILUPP_HIP(b_sc.alloc(sizeof(BicgScalars)));
    double *r0s = b_vec.as<double>(), *p = r0s + n, *s = p + n, *Ap = s + n, *As = Ap + n, *part = b_part.as<double>();
    BicgScalars *sc = b_sc.as<BicgScalars>();
    HostPin pin;
    ILUPP_HIP(hipHostMalloc(reinterpret_cast<void **>(&pin.p), sizeof(double), hipHostMallocDefault));
    ILUPP_HIP(hipMemcpyAsync(r0s, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    ILUPP_HIP(hipMemcpyAsync(p, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    ILUPP_HIP(hipMemsetAsync(y, 0, sizeof(double) * (size_t)n, st));
    ILUPP_HIP(hipMemsetAsync(sc, 0, sizeof(BicgScalars), st));
    hipLaunchKernelGGL(k_bicg_dots, dim3(nb), dim3(kDotThreads), 0, st, n, chunk, (const double *)r, (const double *)r0s, (const double *)r, (const double *)r, part);
    hipLaunchKernelGGL(k_bicg_scalars, dim3(1), dim3(kDotThreads), 0, st, nb, (const double *)part, sc, 3);
    auto residual = [&](double *out) -> int {
        ILUPP_HIP(hipMemcpyAsync(pin.p, &sc->res, sizeof(double), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        *out = *pin.p;
        return ILUPP_OK;
    };
    double initial_res = 0.0, res = 0.0;
    { const int rc = residual(&initial_res); if (rc) return rc; }
    res = initial_res;
    int32_t it = 0;
    // (IEEE: a zero right-hand side gives 0 / 0 = NaN, every comparison with it is false -- min_iter iterations, then "did not converge")
    while ((((res / initial_res > rtol) || res > atol) && it < max_iter) || it < min_iter) {
        ++it;
        { const int rc = op(p, Ap); if (rc) return rc; }
        hipLaunchKernelGGL(k_bicg_dots, dim3(nb), dim3(kDotThreads), 0, st, n, chunk, (const double *)nullptr, (const double *)nullptr, (const double *)Ap,
                           (const double *)r0s, part);
        hipLaunchKernelGGL(k_bicg_scalars, dim3(1), dim3(kDotThreads), 0, st, nb, (const double *)part, sc, 0);
        hipLaunchKernelGGL(k_bicg_s, dim3(gb), dim3(256), 0, st, n, (const double *)r, (const double *)Ap, (const BicgScalars *)sc, s);
        { const int rc = op(s, As); if (rc) return rc; }
        hipLaunchKernelGGL(k_bicg_dots, dim3(nb), dim3(kDotThreads), 0, st, n, chunk, (const double *)As, (const double *)s, (const double *)As, (const double *)As, part);
        hipLaunchKernelGGL(k_bicg_scalars, dim3(1), dim3(kDotThreads), 0, st, nb, (const double *)part, sc, 1);
        hipLaunchKernelGGL(k_bicg_update, dim3(nb), dim3(kDotThreads), 0, st, n, chunk, y, (const double *)p, (const double *)s, (const double *)As, r,
                           (const double *)r0s, (const BicgScalars *)sc, part);
        hipLaunchKernelGGL(k_bicg_scalars, dim3(1), dim3(kDotThreads), 0, st, nb, (const double *)part, sc, 2);
        hipLaunchKernelGGL(k_bicg_p, dim3(gb), dim3(256), 0, st, n, p, (const double *)Ap, (const double *)r, (const BicgScalars *)sc);
        { const int rc = residual(&res); if (rc) return rc; }
    }
    *it_out = it; *rel_out = res / initial_res; *res_out = res;
    return ILUPP_OK;
}

}  // namespace ilupp
