// ilupp_amd/csrc/spmv.hip -- y = A x for a CSR matrix resident in HBM: the other half of a GPU-resident Krylov step
// (the preconditioner's apply being the first), so that an iteration never crosses PCIe.
//
// Reference: matrix_sparse::generic_matrix_vector_multiplication_addition, sparse_implementation.h:2733-2760, ROW/ID
// branch: v[i] += data[j] * x[indices[j]] for j in stored order, starting from 0 -- one lane per row keeps exactly that
// order (separate multiply and add: the library is built with -ffp-contract=off), so the product is bit-identical to the
// reference's and to scipy's csr_matvec.
#include "common.h"

namespace ilupp {

// rows of at most 8 entries: the column indices with two 16-byte loads, the values with four
struct __attribute__((aligned(8))) D2m { double v[2]; };

__global__ void __launch_bounds__(256)
k_spmv_rows(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val, int64_t nnz,
            const double *__restrict__ x, double *__restrict__ y)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int q0 = ptr[r], q1 = ptr[r + 1];
    double acc = 0.0;
    if (q1 - q0 <= 8 && (int64_t)q0 + 8 <= nnz) {
        const Row8 c = load_row8(idx, q0, q1 - q0, nnz);
        double v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const D2m t = *reinterpret_cast<const D2m *>(val + q0 + 2 * i); v[2 * i] = t.v[0]; v[2 * i + 1] = t.v[1]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < q1 - q0) { const double p = v[i] * x[c.c[i]]; acc = acc + p; }
        }
    } else {
        for (int q = q0; q < q1; ++q) { const double p = val[q] * x[idx[q]]; acc = acc + p; }
    }
    y[r] = acc;
}

// longer rows (more than 8 entries on average): 8 lanes per row fetch 8 entries at a time -- coalesced index and value loads, the
// gathers of x in parallel -- and ONE of them adds the 8 products in stored order: the same sum, bit for bit, as the loop above,
// without a row's loads queueing up behind each other on one lane
__global__ void __launch_bounds__(256)
k_spmv_rows8(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
             const double *__restrict__ x, double *__restrict__ y)
{
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    const bool live = r < n;
    const int q0 = live ? ptr[r] : 0, q1 = live ? ptr[r + 1] : 0;
    double acc = 0.0;
    // (every lane of the wave runs the longest row's trip count: the shuffles below need all lanes)
    int trips = (q1 - q0 + 7) >> 3;
    for (int off = 32; off > 0; off >>= 1) trips = max(trips, __shfl_xor(trips, off));
    for (int t = 0; t < trips; ++t) {
        const int q = q0 + t * 8 + l;
        const double p = q < q1 ? val[q] * x[idx[q]] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double pj = __shfl(p, j, 8);
            if (q0 + t * 8 + j < q1) acc = acc + pj;
        }
    }
    if (live && l == 0) y[r] = acc;
}

}  // namespace ilupp

extern "C" int ilupp_hip_spmv_device(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr, int32_t n, int64_t nnz,
                                     const double *d_x, double *d_y, void *hip_stream)
{
    if (n <= 0 || !d_data || !d_indices || !d_indptr || !d_x || !d_y) { ilupp::set_error("spmv: null argument"); return ILUPP_ERR_INVALID; }
    if (nnz > 8 * (int64_t)n)
        hipLaunchKernelGGL(ilupp::k_spmv_rows8, dim3((unsigned)(((int64_t)n * 8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), n,
                           d_indptr, d_indices, d_data, d_x, d_y);
    else
        hipLaunchKernelGGL(ilupp::k_spmv_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), n,
                           d_indptr, d_indices, d_data, nnz, d_x, d_y);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ilupp::set_error(hipGetErrorString(e)); return ILUPP_ERR_HIP; }
    return ILUPP_OK;
}
