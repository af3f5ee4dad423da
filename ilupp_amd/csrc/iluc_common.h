// ilupp_amd/csrc/iluc_common.h -- what the two Crout dataflow kernels share (iluc_df.hip: ILUC2, reference ILUC.hpp:112-207;
// piluc_df.hip: partialILUC of the multilevel preconditioner, reference ILUCDP.hpp:1405-2231): the layout of the control words and
// ready queues, and the passes over A that run before the steps (defined in iluc_df.hip).
#pragma once

#include "common.h"

namespace ilupp {

static constexpr int kCuQ = 64;           // ready queues (step i goes to queue i % nq)
static constexpr int kCuQBase = 64;       // ctrl word of queue 0's head; queue q: head at kCuQBase + 64 q, tail 32 words later
static constexpr unsigned kCuSpinLimit = 1u << 22;

__device__ __forceinline__ unsigned long long cu_pack2(int lo, int hi)
{
    return (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32);
}

// pending[x]: entries of A left of the diagonal in row x and above it in column x; colcnt[c]: rows below the diagonal in column c
__global__ void k_iluc_prep(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, int32_t *pending, int32_t *colcnt);
// the sub-diagonal part of A by columns: for column c the CSR positions of its entries (r, c), r > c ...
__global__ void k_iluc_colfill(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const int32_t *__restrict__ colptr,
                               int32_t *fill, int32_t *colpos);
// ... in the order in which the reference's listA / headA chains are walked (ILUC.hpp:74-101)
__global__ void k_iluc_colorder(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const int32_t *__restrict__ rowof,
                                const int32_t *__restrict__ colptr, const int32_t *__restrict__ colpos, int32_t *__restrict__ colord);
__global__ void k_iluc_rowof(int32_t n, const int32_t *__restrict__ ptr, int32_t *__restrict__ rowof);
// ... for every column: by k_iluc_colorder where columns are short, by the sequential threading itself (host, pattern only) where they are long
int iluc_column_order(hipStream_t st, int32_t m, const DevMat &Av, const int32_t *colcnt, const int32_t *colptr, int32_t *fillc, int32_t *colpos,
                      const int32_t *rowof, int32_t *colord);
// the steps nothing reaches go to the ready queues
__global__ void k_iluc_seed(int32_t m, int32_t nq, const int32_t *__restrict__ pending, int32_t *rq, int32_t *ctrl);

}  // namespace ilupp
