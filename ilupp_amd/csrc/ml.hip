// ilupp_amd/csrc/ml.hip -- the multilevel ILU++ preconditioner without pivoting (the precon_parameter 10
// family of the reference): the loop over the levels of multilevelILUCDPPreconditioner::make_preprocessed_multilevelILUCDP
// (preconditioner_implementation.h:1350-1665, use_ILUC branch) with the preprocessing of matrix_sparse::preprocess
// (sparse_implementation.h:5214-5460) around piluc_level (piluc_df.hip), and the vector kernels of the multilevel apply (:433-488).
//
// Where things run.  The matrix of every level stays in HBM from the input to the last Schur complement: normalisation (column and
// row 2-norms, summed in the reference's storage order), scaling, the application of the permutations, the factorisation, the Schur
// complement.  PQ ordering: the candidate weights and columns are computed on the device (one sequential sum per row) and sorted there
// when all weights are distinct (the order is then unique); with equal weights the order among them is what the reference's own unstable
// quicksort (sparse_implementation.h:471-505) leaves, and that algorithm runs on the host on n weights.  The greedy selection
// (:4611-4634) is sequential by definition: it walks the sorted candidates on the host and the two permutations go back as 2 n
// integers.  The matching (MAX_WEIGHTED_MATCHING_ORDERING) and the move-to-corner ordering are sequential algorithms too (ml_order.hip).
#include <stdlib.h>

#include <chrono>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "iluc_common.h"

namespace ilupp {

// ml_order.hip: the maximum-weight matching (device + host), and the orderings that are sequential algorithms on the host
int mwm_order(hipStream_t st, const DevMat &A, int32_t *p1, double *D1, double *D2);
bool dd_move_corner_host(int32_t n, const int32_t *ptr, const int32_t *idx, const double *val, const int32_t *tptr, const int32_t *tidx, const double *tval,
                         std::vector<int32_t> &P);
void sparse_first_host(int32_t n, std::vector<int32_t> &counts, std::vector<int32_t> &p2);

// ---------------------------------------------- normalisation ----------------------------------------------
// vector_dense::norm2_of_dim1 along the storage order (sparse_implementation.h:820-833) + inverse_scale (:3271-3282), rows of a ROW matrix
__global__ void k_ml_row_norms_scale(int32_t n, const int32_t *__restrict__ ptr, double *__restrict__ val, double *__restrict__ D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) { const double sq = val[j] * val[j]; s = s + sq; }
    const double d = sqrt(s);
    D[i] = d;
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) val[j] = val[j] / d;
}
// ... and columns: the entries of a column in the order of the storage = by increasing row = the order of the transposed storage
__global__ void k_ml_col_norms(int32_t n, const int32_t *__restrict__ tptr, const double *__restrict__ tval, double *__restrict__ D)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double s = 0.0;
    for (int j = tptr[c]; j < tptr[c + 1]; ++j) { const double sq = tval[j] * tval[j]; s = s + sq; }
    D[c] = sqrt(s);
}
__global__ void k_ml_col_scale(int64_t nnz, const int32_t *__restrict__ idx, double *__restrict__ val, const double *__restrict__ D)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nnz) val[j] = val[j] / D[idx[j]];
}
// Dtot[i] *= D[inv[i]]   (D.permute(inv); Dtot.multiply(D), :5243-5244 / :5249-5250)
__global__ void k_ml_fold_scaling(int32_t n, double *__restrict__ Dtot, const double *__restrict__ D, const int32_t *__restrict__ inv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Dtot[i] = Dtot[i] * D[inv ? inv[i] : i];
}
__global__ void k_ml_fill_f64(int32_t n, double *p, double v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// rows divided by D (inverse_scale(D, ROW), sparse_implementation.h:3271-3278)
__global__ void k_ml_row_scale(int32_t n, const int32_t *__restrict__ ptr, double *__restrict__ val, const double *__restrict__ D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = D[i];
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) val[j] = val[j] / d;
}
// unit_or_zero_diagonal (:5172-5178) + inverse_scale(D, ROW): D[i] = the stored diagonal entry if it is not zero, else 1
__global__ void k_ml_unit_diag_scale(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, double *__restrict__ val, double *__restrict__ D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = 1.0;
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) if (idx[j] == i && val[j] != 0.0) d = val[j];
    D[i] = d;
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) val[j] = val[j] / d;
}
__global__ void k_ml_col_counts(int64_t nnz, const int32_t *__restrict__ idx, int32_t *__restrict__ cnt)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nnz) atomicAdd(&cnt[idx[j]], 1);
}

// ---------------------------------------------- PQ ordering ----------------------------------------------
// Algorithm 3.1 of matrix_sparse::ddPQ (:4584-4606): per row the column of the largest magnitude (the first of equals) and the
// weight  - max / (|row|_1 * entries)
__global__ void k_ml_pq_candidates(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ val,
                                   double *__restrict__ W, int32_t *__restrict__ J)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double current_max = 0.0, w = 0.0;
    int jk = 0;
    for (int j = ptr[k]; j < ptr[k + 1]; ++j) {
        const double a = fabs(val[j]);
        w = w + a;
        if (a > current_max) { current_max = a; jk = idx[j]; }
    }
    const double divisor = w * (double)(ptr[k + 1] - ptr[k]);
    W[k] = divisor == 0.0 ? 0.0 : -current_max / divisor;
    J[k] = jk;
}

// sym_ddPQ's weight of a row (:4930-4936): the absolute row sum in storage order, times the number of stored entries
__global__ void k_ml_sym_pq_weights(int32_t n, const int32_t *__restrict__ ptr, const double *__restrict__ val, double *__restrict__ W)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double w = 0.0;
    for (int j = ptr[k]; j < ptr[k + 1]; ++j) w = w + fabs(val[j]);
    W[k] = w * (double)(ptr[k + 1] - ptr[k]);
}

// are two neighbours of the sorted weights equal (or is one a NaN)?  Then the order the reference's quicksort leaves them in is its own
__global__ void k_ml_sorted_ties(int32_t n, const double *__restrict__ w, int32_t *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = w[i];
    if (a != a || (i + 1 < n && !(a < w[i + 1]))) *flag = 1;
}

// ---------------------------------------------- applying permutations to the matrix ----------------------------------------------
// matrix_sparse::permute(p1, p2, ip1, ip2) on a ROW matrix (:5570-5573 -> :5533-5548): new row i = old row p1[i], a column index c
// becomes ip2[c], the rows by increasing index again.  Keys (new row, new column) sorted as 64-bit integers carry the values along.
__global__ void k_ml_perm_lengths(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ p1, int32_t *__restrict__ len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    len[i] = i < n ? ptr[p1[i] + 1] - ptr[p1[i]] : 0;
}
__global__ void k_ml_perm_keys(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const int32_t *__restrict__ ip1,
                               const int32_t *__restrict__ ip2, unsigned long long *__restrict__ keys)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;      // old row
    if (r >= n) return;
    const unsigned long long hi = (unsigned long long)(unsigned)ip1[r] << 32;
    for (int j = ptr[r]; j < ptr[r + 1]; ++j) keys[j] = hi | (unsigned)ip2[idx[j]];
}
__global__ void k_ml_keys_low(int64_t nnz, const unsigned long long *__restrict__ keys, int32_t *__restrict__ idx)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nnz) idx[j] = (int32_t)(unsigned)keys[j];
}

static int scan_i32(hipStream_t st, const int32_t *in, int32_t *out, int count)
{
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, count, st));
    PoolBlock tmp;
    ILUPP_HIP(tmp.alloc(tb > 0 ? tb : 1));
    ILUPP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, in, out, count, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    return ILUPP_OK;
}

// A (ROW storage, owned) := P A Q with the rows re-sorted
static int permute_matrix(hipStream_t st, DevMat *A, const int32_t *d_p1, const int32_t *d_ip1, const int32_t *d_ip2)
{
    const int32_t n = A->n;
    const int64_t nnz = A->nnz;
    PoolBlock b_len, b_k0, b_k1, b_v1, b_tmp, b_nptr;                          // (b_nptr: the new row pointers, handed to A at the end)
    ILUPP_HIP(b_len.alloc(sizeof(int32_t) * (size_t)(n + 1)));
    ILUPP_HIP(b_nptr.alloc(sizeof(int32_t) * (size_t)(n + 1)));
    int32_t *nptr = b_nptr.as<int32_t>();
    hipLaunchKernelGGL(k_ml_perm_lengths, dim3((n + 256) / 256), dim3(256), 0, st, n, A->ptr, d_p1, b_len.as<int32_t>());
    { const int rc = scan_i32(st, b_len.as<int32_t>(), nptr, n + 1); if (rc) return rc; }
    if (nnz > 0) {
        ILUPP_HIP(b_k0.alloc(sizeof(unsigned long long) * (size_t)nnz));
        ILUPP_HIP(b_k1.alloc(sizeof(unsigned long long) * (size_t)nnz));
        ILUPP_HIP(b_v1.alloc(sizeof(double) * (size_t)nnz));
        hipLaunchKernelGGL(k_ml_perm_keys, dim3((n + 255) / 256), dim3(256), 0, st, n, A->ptr, A->idx, d_ip1, d_ip2, b_k0.as<unsigned long long>());
        size_t tb = 0;
        ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, b_k0.as<unsigned long long>(), b_k1.as<unsigned long long>(), A->val, b_v1.as<double>(),
                                                     (int)nnz, 0, 64, st));
        ILUPP_HIP(b_tmp.alloc(tb > 0 ? tb : 1));
        ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, tb, b_k0.as<unsigned long long>(), b_k1.as<unsigned long long>(), A->val, b_v1.as<double>(),
                                                     (int)nnz, 0, 64, st));
        hipLaunchKernelGGL(k_ml_keys_low, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, b_k1.as<unsigned long long>(), A->idx);
        ILUPP_HIP(hipMemcpyAsync(A->val, b_v1.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, st));
    }
    ILUPP_HIP(hipStreamSynchronize(st));
    (void)pool_free(A->ptr);
    A->ptr = static_cast<int32_t *>(b_nptr.release());
    return ILUPP_OK;
}

static void upload_i32(hipStream_t st, int32_t *dst, const std::vector<int32_t> &src)
{
    if (!src.empty()) ILUPP_HIP(hipMemcpyAsync(dst, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice, st));
}

// ---- permutations on the device ----
__global__ void k_ml_gather_i32(int32_t n, const int32_t *__restrict__ a, const int32_t *__restrict__ p, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[p[i]];
}
__global__ void k_ml_invert_i32(int32_t n, const int32_t *__restrict__ p, int32_t *__restrict__ inv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[p[i]] = i;
}
// P := P o p1 (index_list::compose_right, sparse_implementation.h:6165-6183), then its inverse
static void compose_and_invert(hipStream_t st, int32_t n, int32_t *P, int32_t *invP, const int32_t *p1, int32_t *tmp)
{
    const int gb = (n + 255) / 256;
    hipLaunchKernelGGL(k_ml_gather_i32, dim3(gb), dim3(256), 0, st, n, P, p1, tmp);
    ILUPP_HIP(hipMemcpyAsync(P, tmp, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_ml_invert_i32, dim3(gb), dim3(256), 0, st, n, P, invP);
}

// ---- the greedy selection of ddPQ (:4608-4634) on the device.  A candidate k (the row I[k], in sorted order) is taken iff its weight passes
// tau and no EARLIER candidate took its column J[I[k]] -- every row is a candidate exactly once, so "row still free" always holds, and the
// candidates that fail tau are a suffix of the sorted order: the winner of a column is the candidate of smallest k that wants it.
__global__ void k_ml_pq_first(int32_t n, const double *__restrict__ W, const int32_t *__restrict__ I, const int32_t *__restrict__ J, double tau,
                              int32_t *__restrict__ first)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && -W[k] >= tau) atomicMin(&first[J[I[k]]], k);
}
__global__ void k_ml_pq_sel(int32_t n, const double *__restrict__ W, const int32_t *__restrict__ I, const int32_t *__restrict__ J, double tau,
                            const int32_t *__restrict__ first, int32_t *__restrict__ sel)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n) return;
    sel[k] = (k < n && -W[k] >= tau && first[J[I[k]]] == k) ? 1 : 0;
}
__global__ void k_ml_pq_assign(int32_t n, const int32_t *__restrict__ I, const int32_t *__restrict__ J, const int32_t *__restrict__ sel,
                               const int32_t *__restrict__ rank, int32_t *__restrict__ ip, int32_t *__restrict__ iq)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && sel[k]) { const int r = I[k]; ip[r] = rank[k]; iq[J[r]] = rank[k]; }
}
__global__ void k_ml_flag_free(int32_t n, const int32_t *__restrict__ ip, int32_t *__restrict__ flag)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    flag[r] = (r < n && ip[r] < 0) ? 1 : 0;
}
// the rows (columns) nobody took follow in index order (:4624-4634); total = rank[n] = the number taken
__global__ void k_ml_pq_rest(int32_t n, int32_t *__restrict__ ip, const int32_t *__restrict__ flag, const int32_t *__restrict__ off,
                             const int32_t *__restrict__ rank)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && flag[r]) ip[r] = rank[n] + off[r];
}

// Indices by increasing weight, as vector_dense::quicksort(list, 0, n-1) leaves them.  With all weights distinct the order is unique: a
// radix sort on the device.  With equal weights (every interior row of a stencil matrix has the same) the order among them is whatever
// the reference's unstable quicksort (sparse_implementation.h:471-505) leaves -- then that algorithm runs, on the host, on the n weights.
// w: the weights (kept); ws / order: the sorted weights and the list that followed them, device arrays of n.
static int sort_weights(hipStream_t st, int32_t n, const double *w, double *ws, int32_t *order, const char *what)
{
    if (n <= 0) return ILUPP_OK;
    const int gb = (n + 255) / 256;
    PoolBlock b_I, b_flag, b_tmp;
    ILUPP_HIP(b_I.alloc(sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(b_flag.alloc(64));
    ILUPP_HIP(hipMemsetAsync(b_flag.p, 0, 64, st));
    iota_i32(st, b_I.as<int32_t>(), n);
    size_t tb = 0;
    ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, w, ws, b_I.as<int32_t>(), order, n, 0, 64, st));
    ILUPP_HIP(b_tmp.alloc(tb > 0 ? tb : 1));
    ILUPP_HIP(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, tb, w, ws, b_I.as<int32_t>(), order, n, 0, 64, st));
    hipLaunchKernelGGL(k_ml_sorted_ties, dim3(gb), dim3(256), 0, st, n, ws, b_flag.as<int32_t>());
    int32_t ties = 0;
    ILUPP_HIP(hipMemcpyAsync(&ties, b_flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    if (ties) {
        std::vector<double> W((size_t)n);
        std::vector<int32_t> I((size_t)n);
        ILUPP_HIP(hipMemcpyAsync(W.data(), w, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        for (int32_t k = 0; k < n; ++k) I[(size_t)k] = k;
        ref_quicksort(W.data(), I.data(), 0, (long)n - 1);
        ILUPP_HIP(hipMemcpyAsync(ws, W.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
        ILUPP_HIP(hipMemcpyAsync(order, I.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] ml: %s sorted %s\n", what, ties ? "on the host (equal weights)" : "on the device");
    return ILUPP_OK;
}

// matrix_sparse::preprocess (:5214-5460) for the steps this build has; A: ROW storage, replaced by the preprocessed matrix.
// P, Q, invP, invQ (permutation_rows / _columns of the level and their inverses), Drow, Dcol: device arrays of n entries.
static int preprocess_level(hipStream_t st, DevMat *A, const MlParams &IP, int32_t *P, int32_t *Q, int32_t *invP, int32_t *invQ, double *Drow, double *Dcol,
                            int32_t *bad_at)
{
    const int32_t n = A->n;
    const int gb = (n + 255) / 256, gb1 = (n + 1 + 255) / 256;
    iota_i32(st, P, n); iota_i32(st, Q, n); iota_i32(st, invP, n); iota_i32(st, invQ, n);
    bool permuted_rows = false, permuted_cols = false;
    // The reference keeps ONE set of work permutations for all steps of a preprocess() call (sparse_implementation.h:5217-5218) and two
    // steps only resize them (std::vector::resize keeps existing elements): after a step that filled p1 the matching starts with every
    // column "matched", finds no augmenting path and returns the identity with unit scalings (pmwm_implementation.h:411, :460-471);
    // a second PQ step finds everything "taken" and returns the first one's permutations again (:4580-4581, :4613).  Kept as it behaves.
    bool p1_filled = false, pq_done = false;
    *bad_at = n;                                                               // preprocessing_bad_at: what the call returns (:5221, :5459)
    hipLaunchKernelGGL(k_ml_fill_f64, dim3(gb), dim3(256), 0, st, n, Drow, 1.0);
    hipLaunchKernelGGL(k_ml_fill_f64, dim3(gb), dim3(256), 0, st, n, Dcol, 1.0);
    PoolBlock b_D, b_tmpi, b_p1, b_ip1, b_p2, b_ip2, b_id, b_h, b_ih;
    ILUPP_HIP(b_D.alloc(sizeof(double) * (size_t)n));
    for (PoolBlock *b : {&b_tmpi, &b_p1, &b_ip1, &b_p2, &b_ip2, &b_id, &b_h, &b_ih}) ILUPP_HIP(b->alloc(sizeof(int32_t) * (size_t)(n + 1)));
    int32_t *tmpi = b_tmpi.as<int32_t>(), *p1 = b_p1.as<int32_t>(), *ip1 = b_ip1.as<int32_t>(), *p2 = b_p2.as<int32_t>(), *ip2 = b_ip2.as<int32_t>(),
            *ident = b_id.as<int32_t>(), *hq = b_h.as<int32_t>(), *ihq = b_ih.as<int32_t>();   // p1..ip2: the PQ step's; hq, ihq: a host-made ordering
    iota_i32(st, ident, n);
    const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    for (int s = 0; s < IP.n_pre; ++s) {
        struct StepTimer {
            bool on; int step; std::chrono::steady_clock::time_point t0; hipStream_t st;
            ~StepTimer() { if (on) { (void)hipStreamSynchronize(st); fprintf(stderr, "[ilupp] ml: preprocessing step %d: %.2f ms\n", step,
                                                                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } }
        } timer{dbg, IP.pre[s], std::chrono::steady_clock::now(), st};
        if (IP.pre[s] != ML_PRE_PQ_ORDERING) *bad_at = n;
        switch (IP.pre[s]) {
        case ML_PRE_NORMALIZE_COLUMNS: {                                       // :5241-5246
            struct Temp { DevMat m; ~Temp() { m.release(); } } tmp_T;
            DevMat &T = tmp_T.m;
            transpose_storage(st, *A, &T);                                     // (column-major copy: the entries of a column by increasing row)
            hipLaunchKernelGGL(k_ml_col_norms, dim3(gb), dim3(256), 0, st, n, T.ptr, T.val, b_D.as<double>());
            if (A->nnz > 0)
                hipLaunchKernelGGL(k_ml_col_scale, dim3((unsigned)((A->nnz + 255) / 256)), dim3(256), 0, st, A->nnz, A->idx, A->val, b_D.as<double>());
            hipLaunchKernelGGL(k_ml_fold_scaling, dim3(gb), dim3(256), 0, st, n, Dcol, b_D.as<double>(), permuted_cols ? (const int32_t *)invQ : (const int32_t *)nullptr);
            ILUPP_HIP(hipStreamSynchronize(st));
            T.release();
            break;
        }
        case ML_PRE_NORMALIZE_ROWS:                                            // :5247-5252
            hipLaunchKernelGGL(k_ml_row_norms_scale, dim3(gb), dim3(256), 0, st, n, A->ptr, A->val, b_D.as<double>());
            hipLaunchKernelGGL(k_ml_fold_scaling, dim3(gb), dim3(256), 0, st, n, Drow, b_D.as<double>(), permuted_rows ? (const int32_t *)invP : (const int32_t *)nullptr);
            break;
        case ML_PRE_PQ_ORDERING: {                                             // :5264-5275
            if (!pq_done) {
                PoolBlock b_J, b_W2, b_I2, b_sel, b_rank, b_off;
                ILUPP_HIP(b_J.alloc(sizeof(int32_t) * (size_t)n));
                hipLaunchKernelGGL(k_ml_pq_candidates, dim3(gb), dim3(256), 0, st, n, A->ptr, A->idx, A->val, b_D.as<double>(), b_J.as<int32_t>());
                ILUPP_HIP(b_W2.alloc(sizeof(double) * (size_t)n));
                ILUPP_HIP(b_I2.alloc(sizeof(int32_t) * (size_t)n));
                { const int rc = sort_weights(st, n, b_D.as<double>(), b_W2.as<double>(), b_I2.as<int32_t>(), "PQ candidates"); if (rc) return rc; }
                // the greedy selection, on the device (see k_ml_pq_first)
                ILUPP_HIP(b_sel.alloc(sizeof(int32_t) * (size_t)(n + 1)));
                ILUPP_HIP(b_rank.alloc(sizeof(int32_t) * (size_t)(n + 1)));
                ILUPP_HIP(b_off.alloc(sizeof(int32_t) * (size_t)(n + 1)));
                const double *W2 = b_W2.as<double>();
                const int32_t *I2 = b_I2.as<int32_t>(), *J = b_J.as<int32_t>();
                ILUPP_HIP(hipMemsetAsync(tmpi, 0x7f, sizeof(int32_t) * (size_t)n, st));                  // first[c] = 0x7f7f7f7f: nobody yet
                ILUPP_HIP(hipMemsetAsync(ip1, 0xff, sizeof(int32_t) * (size_t)n, st));
                ILUPP_HIP(hipMemsetAsync(ip2, 0xff, sizeof(int32_t) * (size_t)n, st));
                hipLaunchKernelGGL(k_ml_pq_first, dim3(gb), dim3(256), 0, st, n, W2, I2, J, IP.pq_threshold, tmpi);
                hipLaunchKernelGGL(k_ml_pq_sel, dim3(gb1), dim3(256), 0, st, n, W2, I2, J, IP.pq_threshold, tmpi, b_sel.as<int32_t>());
                { const int rc = scan_i32(st, b_sel.as<int32_t>(), b_rank.as<int32_t>(), n + 1); if (rc) return rc; }
                hipLaunchKernelGGL(k_ml_pq_assign, dim3(gb), dim3(256), 0, st, n, I2, J, b_sel.as<int32_t>(), b_rank.as<int32_t>(), ip1, ip2);
                for (int32_t *ip : {ip1, ip2}) {
                    hipLaunchKernelGGL(k_ml_flag_free, dim3(gb1), dim3(256), 0, st, n, ip, b_sel.as<int32_t>());
                    { const int rc = scan_i32(st, b_sel.as<int32_t>(), b_off.as<int32_t>(), n + 1); if (rc) return rc; }
                    hipLaunchKernelGGL(k_ml_pq_rest, dim3(gb), dim3(256), 0, st, n, ip, b_sel.as<int32_t>(), b_off.as<int32_t>(), b_rank.as<int32_t>());
                }
                hipLaunchKernelGGL(k_ml_invert_i32, dim3(gb), dim3(256), 0, st, n, ip1, p1);
                hipLaunchKernelGGL(k_ml_invert_i32, dim3(gb), dim3(256), 0, st, n, ip2, p2);
                ILUPP_HIP(hipMemcpyAsync(bad_at, b_rank.as<int32_t>() + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));     // the number taken (:4635)
                ILUPP_HIP(hipStreamSynchronize(st));
                pq_done = true;
            } else *bad_at = 0;                                                // (the repeated call finds nothing to take: count = -1, returns pos + 1 = 0)
            // (a later PQ step of the same call: ip1 / ip2 / p1 / p2 still hold the first one's permutations)
            p1_filled = true;
            { const int rc = permute_matrix(st, A, p1, ip1, ip2); if (rc) return rc; }
            compose_and_invert(st, n, P, invP, p1, tmpi);                      // P.compose_right(p1); Q.compose_right(p2); the inverses (:5269-5272)
            compose_and_invert(st, n, Q, invQ, p2, tmpi);
            permuted_rows = permuted_cols = true;
            break;
        }
        case ML_PRE_MAX_WEIGHTED_MATCHING_ORDERING: {                         // :5276-5292
            // the matching: initialisation and first matching on the device, logarithms / exponentials and the augmenting-path searches
            // for what is left on the host (ml_order.hip); the permutation and both scalings stay on the device
            const int64_t nnz = A->nnz;
            PoolBlock b_D2;
            ILUPP_HIP(b_D2.alloc(sizeof(double) * (size_t)n));
            if (!p1_filled) { const int rc = mwm_order(st, *A, hq, b_D.as<double>(), b_D2.as<double>()); if (rc) return rc; }
            else {                                                             // (see above: the reference's search starts "all matched")
                iota_i32(st, hq, n);
                hipLaunchKernelGGL(k_ml_fill_f64, dim3(gb), dim3(256), 0, st, n, b_D.as<double>(), 1.0);
                hipLaunchKernelGGL(k_ml_fill_f64, dim3(gb), dim3(256), 0, st, n, b_D2.as<double>(), 1.0);
            }
            p1_filled = true;
            hipLaunchKernelGGL(k_ml_invert_i32, dim3(gb), dim3(256), 0, st, n, hq, ihq);
            hipLaunchKernelGGL(k_ml_row_scale, dim3(gb), dim3(256), 0, st, n, A->ptr, A->val, b_D.as<double>());           // inverse_scale(D1, ROW)
            if (nnz > 0)
                hipLaunchKernelGGL(k_ml_col_scale, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, A->idx, A->val, b_D2.as<double>());   // (D2, COLUMN)
            { const int rc = permute_matrix(st, A, hq, ihq, ident); if (rc) return rc; }                                    // permute(p1, ROW)
            hipLaunchKernelGGL(k_ml_fold_scaling, dim3(gb), dim3(256), 0, st, n, Drow, b_D.as<double>(), permuted_rows ? (const int32_t *)invP : (const int32_t *)nullptr);
            hipLaunchKernelGGL(k_ml_fold_scaling, dim3(gb), dim3(256), 0, st, n, Dcol, b_D2.as<double>(), permuted_cols ? (const int32_t *)invQ : (const int32_t *)nullptr);
            compose_and_invert(st, n, P, invP, hq, tmpi);                      // P.compose_right(p1); invP.invert(P)
            ILUPP_HIP(hipStreamSynchronize(st));
            permuted_rows = true;
            break;
        }
        case ML_PRE_UNIT_OR_ZERO_DIAGONAL_SCALING:                             // :5309-5314 (Drow.multiply(D1) without a permutation, as there)
            hipLaunchKernelGGL(k_ml_unit_diag_scale, dim3(gb), dim3(256), 0, st, n, A->ptr, A->idx, A->val, b_D.as<double>());
            hipLaunchKernelGGL(k_ml_fold_scaling, dim3(gb), dim3(256), 0, st, n, Drow, b_D.as<double>(), (const int32_t *)nullptr);
            break;
        case ML_PRE_SPARSE_FIRST_ORDERING: {                                   // :5293-5299
            PoolBlock b_cnt;
            ILUPP_HIP(b_cnt.alloc(sizeof(int32_t) * (size_t)n));
            ILUPP_HIP(hipMemsetAsync(b_cnt.p, 0, sizeof(int32_t) * (size_t)n, st));
            if (A->nnz > 0)
                hipLaunchKernelGGL(k_ml_col_counts, dim3((unsigned)((A->nnz + 255) / 256)), dim3(256), 0, st, A->nnz, A->idx, b_cnt.as<int32_t>());
            std::vector<int32_t> cnt((size_t)n), hp2, hip2((size_t)n);
            ILUPP_HIP(hipMemcpyAsync(cnt.data(), b_cnt.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
            ILUPP_HIP(hipStreamSynchronize(st));
            sparse_first_host(n, cnt, hp2);
            for (int32_t i = 0; i < n; ++i) hip2[(size_t)hp2[(size_t)i]] = i;
            upload_i32(st, hq, hp2); upload_i32(st, ihq, hip2);
            { const int rc = permute_matrix(st, A, ident, ident, ihq); if (rc) return rc; }                                  // permute(p2, COLUMN)
            compose_and_invert(st, n, Q, invQ, hq, tmpi);
            ILUPP_HIP(hipStreamSynchronize(st));
            permuted_cols = true;
            break;
        }
        case ML_PRE_SYMM_PQ: {                                                 // :5352-5360
            // sym_ddPQ (:4926-4940): rows AND columns by increasing (sum of |a_ij|) * (entries of the row); its work list starts as the identity
            PoolBlock b_W2;
            ILUPP_HIP(b_W2.alloc(sizeof(double) * (size_t)(n > 0 ? n : 1)));
            hipLaunchKernelGGL(k_ml_sym_pq_weights, dim3(gb), dim3(256), 0, st, n, A->ptr, A->val, b_D.as<double>());
            { const int rc = sort_weights(st, n, b_D.as<double>(), b_W2.as<double>(), hq, "symmetric PQ weights"); if (rc) return rc; }
            p1_filled = true;
            hipLaunchKernelGGL(k_ml_invert_i32, dim3(gb), dim3(256), 0, st, n, hq, ihq);
            { const int rc = permute_matrix(st, A, hq, ihq, ihq); if (rc) return rc; }     // permute(p1, p1)
            compose_and_invert(st, n, P, invP, hq, tmpi);
            compose_and_invert(st, n, Q, invQ, hq, tmpi);
            ILUPP_HIP(hipStreamSynchronize(st));
            permuted_rows = permuted_cols = true;
            break;
        }
        case ML_PRE_DD_SYMM_MOVE_CORNER_ORDERING_IM: {                         // :5441-5450
            const int64_t nnz = A->nnz;
            struct Temp { DevMat m; ~Temp() { m.release(); } } tmp_T;
            DevMat &T = tmp_T.m;
            transpose_storage(st, *A, &T);
            std::vector<int32_t> hp((size_t)n + 1), hi((size_t)(nnz > 0 ? nnz : 1)), tp((size_t)n + 1), ti((size_t)(nnz > 0 ? nnz : 1)), hp1, hip1((size_t)n);
            std::vector<double> hv((size_t)(nnz > 0 ? nnz : 1)), tv((size_t)(nnz > 0 ? nnz : 1));
            ILUPP_HIP(hipMemcpyAsync(hp.data(), A->ptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
            ILUPP_HIP(hipMemcpyAsync(tp.data(), T.ptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
            if (nnz > 0) {
                ILUPP_HIP(hipMemcpyAsync(hi.data(), A->idx, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, st));
                ILUPP_HIP(hipMemcpyAsync(hv.data(), A->val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, st));
                ILUPP_HIP(hipMemcpyAsync(ti.data(), T.idx, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, st));
                ILUPP_HIP(hipMemcpyAsync(tv.data(), T.val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, st));
            }
            ILUPP_HIP(hipStreamSynchronize(st));
            T.release();
            if (!dd_move_corner_host(n, hp.data(), hi.data(), hv.data(), tp.data(), ti.data(), tv.data(), hp1)) {
                set_error("DD_SYMM_MOVE_CORNER_ORDERING_IM: the ordering rejects an index of this matrix, and the reference's result is then undefined (it refills its "
                          "container with stale flags, sparse_implementation.h:5014 / arrays_implementation.h:55-63, and returns indices that repeat): refused");
                return ILUPP_ERR_UNSUPPORTED;
            }
            p1_filled = true;
            for (int32_t i = 0; i < n; ++i) hip1[(size_t)hp1[(size_t)i]] = i;
            upload_i32(st, hq, hp1); upload_i32(st, ihq, hip1);
            { const int rc = permute_matrix(st, A, hq, ihq, ihq); if (rc) return rc; }     // permute(p1, p1)
            compose_and_invert(st, n, P, invP, hq, tmpi);
            compose_and_invert(st, n, Q, invQ, hq, tmpi);
            ILUPP_HIP(hipStreamSynchronize(st));
            permuted_rows = permuted_cols = true;
            break;
        }
        default:
            set_error("ILU++ preprocessing step " + std::to_string(IP.pre[s]) + " is not built");
            return ILUPP_ERR_UNSUPPORTED;
        }
    }
    ILUPP_HIP(hipStreamSynchronize(st));
    return ILUPP_OK;
}

void MlLevelDev::release()
{
    L.release(); U.release();
    for (double *p : {D, Dl, Dr}) if (p) (void)pool_free(p);
    for (int32_t *p : {pr, pc, ipr, ipc}) if (p) (void)pool_free(p);
    D = Dl = Dr = nullptr; pr = pc = ipr = ipc = nullptr;
}

// make_preprocessed_multilevelILUCDP (preconditioner_implementation.h:1350-1665), use_ILUC branch.  A: the input in either storage.
int ml_build(hipStream_t st, const DevMat &A, const MlParams &IP, std::vector<MlLevelDev> *levels, float *kernel_ms)
{
    DevMat Ak;                                                                 // the matrix of the current level, ROW storage, owned
    if (A.is_csr) {
        Ak.n = A.n; Ak.nnz = A.nnz; Ak.is_csr = true; Ak.owns = true;
        ILUPP_HIP(pool_malloc(&Ak.ptr, sizeof(int32_t) * (size_t)(A.n + 1)));
        ILUPP_HIP(pool_malloc(&Ak.idx, sizeof(int32_t) * (size_t)(A.nnz > 0 ? A.nnz : 1)));
        ILUPP_HIP(pool_malloc(&Ak.val, sizeof(double) * (size_t)(A.nnz > 0 ? A.nnz : 1)));
        ILUPP_HIP(hipMemcpyAsync(Ak.ptr, A.ptr, sizeof(int32_t) * (size_t)(A.n + 1), hipMemcpyDeviceToDevice, st));
        if (A.nnz > 0) {
            ILUPP_HIP(hipMemcpyAsync(Ak.idx, A.idx, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToDevice, st));
            ILUPP_HIP(hipMemcpyAsync(Ak.val, A.val, sizeof(double) * (size_t)A.nnz, hipMemcpyDeviceToDevice, st));
        }
    } else {
        transpose_storage(st, A, &Ak);                                         // change_orientation(), :1388-1391
        Ak.is_csr = true;
    }
    struct Guard { DevMat *m; std::vector<MlLevelDev> *lv; bool ok = false; ~Guard() { m->release(); if (!ok) { for (auto &l : *lv) l.release(); lv->clear(); } } } guard{&Ak, levels};
    double tau = IP.threshold;
    int32_t matrix_size = Ak.n;
    int64_t nonzeroes = Ak.nnz;
    int nlev = 0;
    for (;;) {
        const bool in_loop = matrix_size > IP.min_ml_size && nlev < IP.max_levels - 1 && nonzeroes > 0;     // :1405
        if (!in_loop && !(matrix_size > 0)) break;                                                           // :1550
        const int32_t m = Ak.n;
        levels->emplace_back();
        MlLevelDev &l = levels->back();
        l.n = m;
        ILUPP_HIP(pool_malloc(&l.Dl, sizeof(double) * (size_t)m));
        ILUPP_HIP(pool_malloc(&l.Dr, sizeof(double) * (size_t)m));
        for (int32_t **d : {&l.pr, &l.pc, &l.ipr, &l.ipc}) ILUPP_HIP(pool_malloc(d, sizeof(int32_t) * (size_t)m));
        int32_t end_PQ = m;
        { const int rc = preprocess_level(st, &Ak, IP, l.pr, l.pc, l.ipr, l.ipc, l.Dl, l.Dr, &end_PQ); if (rc) return rc; }
        if (!in_loop && IP.use_final_threshold) tau *= IP.final_threshold;     // :1580-1581
        DevMat Anext;
        int32_t kterm = m;
        const auto tl0 = std::chrono::steady_clock::now();
        int rc;
        if (!IP.pil.pivoting()) {
            // dataflow over all CUs -- unless the dropping rules are recurrences over all steps (inverse-based, weighted), which only a
            // sequential walk can run: the chain kernel (pilucdp.hip: k_piluc_chain); ILUPP_PILUC_CHAIN=1 sends everything there (tests)
            const bool sequential_rules = (IP.pil.rules & (PILUC_DROP_INVERSE | PILUC_DROP_WEIGHTED | PILUC_DROP_WEIGHTED2)) != 0;
            if (sequential_rules || getenv("ILUPP_PILUC_CHAIN")) {
                rc = piluc_chain_level(st, Ak, IP.pil, !in_loop, tau, &l.L, &l.U, &l.D, &Anext, &kterm, kernel_ms, sequential_rules || getenv("ILUPP_PILUC_CHAIN_MEM") != nullptr);
                if (rc == 1) {
                    l.L.release(); l.U.release(); Anext.release();
                    if (sequential_rules) {
                        set_error("partialILUC with inverse-based / weighted dropping: a working row of more than 32768 entries (the chain kernel's capacity)");
                        rc = ILUPP_ERR_UNSUPPORTED;
                    } else rc = piluc_level(st, Ak, IP.pil, !in_loop, tau, &l.L, &l.U, &l.D, &Anext, &kterm, kernel_ms);
                }
            } else rc = piluc_level(st, Ak, IP.pil, !in_loop, tau, &l.L, &l.U, &l.D, &Anext, &kterm, kernel_ms);
        } else {
            // the windows of the pivoting and of the row reordering, :1442-1459 (a level of the loop) / :1564-1577 (the last one)
            const int32_t last_row_to_eliminate = in_loop ? (m - 1) / 2 : m - 1;
            int32_t bp, bpr, epr;
            switch (IP.pil.permute_rows) {
            case 0: bpr = 0; epr = 0; break;
            case 1: bpr = end_PQ; epr = m - 1; break;
            case 2: bpr = 0; epr = in_loop ? last_row_to_eliminate : m - 1; break;
            default: bpr = 0; epr = m - 1; break;
            }
            switch (IP.pil.total_piv) {
            case 0: bp = m; break;
            case 1: bp = in_loop ? last_row_to_eliminate + 1 : end_PQ; break;
            default: bp = 0; break;
            }
            PoolBlock b_p;
            ILUPP_HIP(b_p.alloc(sizeof(int32_t) * (size_t)m * 3));
            int32_t *pc2 = b_p.as<int32_t>(), *pr2 = pc2 + m, *tmp = pr2 + m;
            rc = pilucdp_level(st, Ak, IP.pil, !in_loop, tau, bp, bpr, epr, &l.L, &l.U, &l.D, &Anext, pc2, pr2, kernel_ms);
            if (rc == ILUPP_OK) {
                // permutation_columns.compose(pc1, pc2), permutation_rows.compose(pr1, pr2) and their inverses, :1516-1521
                compose_and_invert(st, m, l.pc, l.ipc, pc2, tmp);
                compose_and_invert(st, m, l.pr, l.ipr, pr2, tmp);
                ILUPP_HIP(hipStreamSynchronize(st));
            }
        }
        if (getenv("ILUPP_DEBUG")) fprintf(stderr, "[ilupp] ml: level %d (n %d): factorisation %.2f ms\n", nlev, m,
                                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count());
        if (rc) { Anext.release(); return rc; }
        ++nlev;
        Ak.release();
        Ak = Anext;                                                            // :1532
        Ak.owns = true;
        if (!in_loop) break;
        matrix_size = Ak.n; nonzeroes = Ak.nnz;
        tau *= IP.vary_threshold_factor;
    }
    guard.ok = true;
    return ILUPP_OK;
}

// ---------------------------------------------- the vector kernels of the apply (:441-486) ----------------------------------------------
// out[i] = x[perm[i]] / D[perm[i]]         inverse_scale_at_end + permute_first (sparse_implementation.h:170-176, :4142-4153)
__global__ void k_ml_scale_perm(int32_t n, const double *__restrict__ x, const double *__restrict__ D, const int32_t *__restrict__ perm, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int s = perm[i]; out[i] = x[s] / D[s]; }
}
// x[i] = w[i] (* D[i]); w[i] = the sweeps' "not yet" mark again
__global__ void k_ml_take(int32_t n, double *__restrict__ w, const double *__restrict__ D, double *__restrict__ x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = w[i];
    x[i] = D ? v * D[i] : v;
    reinterpret_cast<unsigned long long *>(w)[i] = kSentinel;
}
// out[i] = x[i] * D[i]                      scale_at_end (:153-159)
__global__ void k_ml_scale(int32_t n, const double *__restrict__ x, const double *__restrict__ D, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * D[i];
}
// x[i] = w[perm[i]] / D[i]                  permute_last + inverse_scale_at_end (:4156-4165, :170-176)
__global__ void k_ml_perm_scale(int32_t n, const double *__restrict__ w, const int32_t *__restrict__ perm, const double *__restrict__ D, double *__restrict__ x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = w[perm[i]] / D[i];
}

void ml_scale_perm(hipStream_t st, int32_t n, const double *x, const double *D, const int32_t *perm, double *out)
{ hipLaunchKernelGGL(k_ml_scale_perm, dim3((n + 255) / 256), dim3(256), 0, st, n, x, D, perm, out); }
void ml_take(hipStream_t st, int32_t n, double *w, const double *D, double *x)
{ hipLaunchKernelGGL(k_ml_take, dim3((n + 255) / 256), dim3(256), 0, st, n, w, D, x); }
void ml_scale(hipStream_t st, int32_t n, const double *x, const double *D, double *out)
{ hipLaunchKernelGGL(k_ml_scale, dim3((n + 255) / 256), dim3(256), 0, st, n, x, D, out); }
void ml_perm_scale(hipStream_t st, int32_t n, const double *w, const int32_t *perm, const double *D, double *x)
{ hipLaunchKernelGGL(k_ml_perm_scale, dim3((n + 255) / 256), dim3(256), 0, st, n, w, perm, D, x); }

}  // namespace ilupp
