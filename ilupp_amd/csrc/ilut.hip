// ilupp_amd/csrc/ilut.hip -- Saad's ILUT(p, tau), heap variant, for gfx950: the host driver.
//
// Replaces ILUT_heap (reference ILUT.hpp:199-278), threshold_and_drop (dropping.hpp:8-34), the working-row
// containers vector_sparse_dynamic / vector_sparse_ordered (sparse.h:169-323, sparse_implementation.h:950-1093),
// append_row_with_prefix/suffix (:3190-3230) and compress (:3696-3722).
//
// The sparsity pattern of L and U depends on the VALUES (thresholds, top-k by magnitude), so nothing can be scheduled ahead:
// the rows are computed by a row-wise dataflow kernel, one wave per row (ilut_wp.hip: the working row as three
// insertion-ordered pieces in LDS, in 64 K global pieces, or -- the largest capacity class -- in pieces as long as the matrix is
// wide).  Rows go to fixed-pitch slabs (L: p entries per row in plain arrays; U: one record of whole 128-byte lines per row -- length,
// columns, values -- because other rows fetch it while the kernel runs); the compaction pass here drops exact zeros like compress(0.0)
// and makes the CSR arrays of L (unit diagonal last) and U (pivot first).
#include <stdio.h>
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ilupp {

// ---- compaction of the fixed-pitch slabs into CSR, dropping exact zeros (compress(0.0), :3696-3722) -------
// (a row of the slab is lay.si ints / lay.sv doubles / lay.sl ints behind the previous one in the three arrays: p, p, 1 for L's plain
// slabs; the strides of the record slab for U -- common.h: UrowLayout)
__global__ void k_slab_count(int32_t n, UrowLayout lay, const int32_t *__restrict__ rlen, const double *__restrict__ rval, int32_t *__restrict__ cnt)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int c = 0;
    const size_t b = (size_t)r * lay.sv;
    const int len = rlen[(size_t)r * lay.sl];
    for (int q = 0; q < len; ++q) c += (fabs(rval[b + q]) > 0.0) ? 1 : 0;
    cnt[r] = c;
}
__global__ void k_slab_fill(int32_t n, UrowLayout lay, const int32_t *__restrict__ rlen, const int32_t *__restrict__ ridx,
                            const double *__restrict__ rval, const int32_t *__restrict__ ptr,
                            int32_t *__restrict__ idx, double *__restrict__ val)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int o = ptr[r];
    const size_t bi = (size_t)r * lay.si, bv = (size_t)r * lay.sv;
    const int len = rlen[(size_t)r * lay.sl];
    for (int q = 0; q < len; ++q)
        if (fabs(rval[bv + q]) > 0.0) { idx[o] = ridx[bi + q]; val[o] = rval[bv + q]; ++o; }
}

static void compact_slab(hipStream_t st, int32_t n, UrowLayout lay, const int32_t *rlen, const int32_t *ridx, const double *rval, DevMat *M)
{
    int32_t *cnt = nullptr;
    ILUPP_HIP(pool_malloc(&cnt, sizeof(int32_t) * (size_t)n));
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_slab_count, dim3(gb), dim3(256), 0, st, n, lay, rlen, rval, cnt);
    M->n = n; M->is_csr = true; M->owns = true;
    ILUPP_HIP(pool_malloc(&M->ptr, sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(hipMemsetAsync(M->ptr, 0, sizeof(int32_t), st));
    size_t tmp_bytes = 0;
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, cnt, M->ptr + 1, n, st));
    void *tmp = nullptr;
    ILUPP_HIP(pool_malloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    ILUPP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, cnt, M->ptr + 1, n, st));
    int32_t tot = 0;
    ILUPP_HIP(hipMemcpyAsync(&tot, M->ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    ILUPP_HIP(pool_free(tmp)); ILUPP_HIP(pool_free(cnt));
    M->nnz = tot;
    ILUPP_HIP(pool_malloc(&M->idx, sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1)));
    ILUPP_HIP(pool_malloc(&M->val, sizeof(double) * (size_t)(tot > 0 ? tot : 1)));
    hipLaunchKernelGGL(k_slab_fill, dim3(gb), dim3(256), 0, st, n, lay, rlen, ridx, rval, M->ptr, M->idx, M->val);
}

// ILUT of the row-major view held in A; on ILUPP_ERR_ZERO_PIVOT *err_row is the first row with a zero pivot
int ilut_factor(hipStream_t st, const DevMat &A, int32_t max_fill_in, double threshold, DevMat *L, DevMat *U,
                int32_t *err_row, float *kernel_ms)
{
    const int32_t n = A.n;
    int32_t p = max_fill_in;
    if (p < 1) p = 1;
    if (p > n) p = n;                                           // ILUT.hpp:211-212
    const size_t slab = (size_t)n * p;
    // U's rows as records of whole 128-byte lines: [length][p columns][pad to 8][p values] (other rows fetch them while the kernel runs)
    const size_t voff = ((size_t)4 + 4 * (size_t)p + 7) & ~(size_t)7;
    const size_t rec = (voff + 8 * (size_t)p + 127) & ~(size_t)127;
    if (rec / 4 > 0x7fffffffULL) { set_error("ILUT: the fill budget is too large for the row records"); return ILUPP_ERR_UNSUPPORTED; }
    const UrowLayout ulay = {(int32_t)(rec / 4), (int32_t)(rec / 8), (int32_t)(rec / 4)}, llay = {p, p, 1};
    PoolBlock b_Lri, b_Urec, b_Lrv, b_Llen, b_ctrl;
    ILUPP_HIP(b_Lri.alloc(sizeof(int32_t) * slab));
    ILUPP_HIP(b_Urec.alloc(rec * (size_t)n));
    ILUPP_HIP(b_Lrv.alloc(sizeof(double) * slab));
    ILUPP_HIP(b_Llen.alloc(sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(b_ctrl.alloc(256));
    unsigned char *urec = b_Urec.as<unsigned char>();
    int32_t *Lri = b_Lri.as<int32_t>(), *Uri = reinterpret_cast<int32_t *>(urec + 4), *Llen = b_Llen.as<int32_t>(),
            *Ulen = reinterpret_cast<int32_t *>(urec), *ctrl = b_ctrl.as<int32_t>();
    double *Lrv = b_Lrv.as<double>(), *Urv = reinterpret_cast<double *>(urec + voff);
    const int rw = ilut_rows_wp(st, A, p, threshold, Lri, Lrv, Llen, Uri, Urv, Ulen, ulay, ctrl, kernel_ms);
    if (rw == 1) {
        set_error("ILUT: a working row does not fit the largest capacity class (the matrix is too wide for the memory budget)");
        return ILUPP_ERR_UNSUPPORTED;
    }
    int32_t hw[4] = {0, 0, 0, 0};
    ILUPP_HIP(hipMemcpyAsync(hw, ctrl, 16, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    int rc = rw;
    if (rc == ILUPP_OK && hw[2] != 0x7fffffff) { rc = ILUPP_ERR_ZERO_PIVOT; if (err_row) *err_row = hw[2]; }
    if (rc == ILUPP_OK) {
        compact_slab(st, n, llay, Llen, Lri, Lrv, L);
        compact_slab(st, n, ulay, Ulen, Uri, Urv, U);
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    return rc;
}

}  // namespace ilupp
