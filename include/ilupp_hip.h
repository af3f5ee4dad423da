/*
 * include/ilupp_hip.h -- C ABI of the MI355X-native incomplete-factorisation engine.
 *
 * Drop-in boundary for the hot path of c-f-h/ilupp: every entry point below
 * replaces one definition of the reference's pybind11 translation unit src/binding.cpp (cited per
 * function, paths relative to /root/reference).  Plain pointers and sizes only; no torch, numpy or
 * pybind types.  Indices are int32 (the reference's default `Integer`, declarations.h:49-53), values
 * fp64.
 *
 * All functions return ILUPP_OK (0) or a negative ilupp_status; ilupp_hip_last_error() gives the
 * message the reference would have thrown (same wording where the reference has one).
 *
 * Host-pointer entry points borrow the caller's buffers for the duration of the call only
 * (binding.cpp:85-89: non-owning views).  `_device` variants take pointers that already live in this
 * GPU's HBM; they are what bench.py times.
 *
 * Threads: the reference has no internal threads and no global state on this path ("Threading /
 * reentrancy"); its Python callers are single-threaded.  Here every object owns one HIP stream and its entry points are
 * serialised by the caller per object; applies of DISTINCT finished objects may run from distinct host threads.  CONSTRUCTIONS
 * (the *_create*, *_refactor* entry points) may be CALLED from several threads, but the library runs them one at a time: one
 * process-wide mutex, also across devices -- a process that drives two GPUs gets no overlap of its constructions.  The reason: their
 * work arrays come from one process-wide pool of device blocks that knows nothing of streams -- a block released while the releasing
 * object's stream still uses it is only safe to hand to work queued on that same stream or after that stream has been waited for,
 * which every construction does before it returns.  The ways to build several factorisations at once: ilupp_hip_ml_create_batch (one
 * call, many matrices, the pool's blocks partitioned per worker), or one process per GPU (bench.py --gpus N, batched.py).
 */
#ifndef ILUPP_HIP_H
#define ILUPP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ilupp_precond ilupp_precond;   /* opaque; owns the factors in HBM */

typedef enum {
    ILUPP_OK = 0,
    ILUPP_ERR_INVALID = -1,        /* bad argument (binding.cpp:77-83 "matrix has size 0!", size mismatch) */
    ILUPP_ERR_WRONG_SIZE = -2,     /* binding.cpp:241-242 "vector has wrong size for preconditioner!" */
    ILUPP_ERR_ZERO_PIVOT = -3,     /* ILUT.hpp:269-270 "ILUT_heap: encountered zero pivot in row N" */
    ILUPP_ERR_NOT_TRIANGULAR = -4, /* IChol.hpp:54-55,105-107 */
    ILUPP_ERR_NO_DIAGONAL = -5,    /* ILU(0): structurally missing diagonal (reference: undefined behaviour, ILU0.hpp:39-40,64) */
    ILUPP_ERR_HIP = -6,            /* HIP runtime failure */
    ILUPP_ERR_TIMEOUT = -7,        /* dependency wait exceeded its bound (cyclic/invalid structure) */
    ILUPP_ERR_UNSUPPORTED = -8,    /* path not built yet in this round */
    ILUPP_ERR_MEMORY = -9,         /* sparse_implementation.h:3178-3179 "insufficient memory reserved" */
    ILUPP_ERR_NOT_SPD = -10,       /* ICholT: the pivot of a column is NaN (the matrix is not positive definite).  The reference has no
                                      positivity check (IChol.hpp:115-117): such a column's entries are all NaN, none passes the threshold
                                      test, the column is stored empty and its NaNs spread through the diagonals it touched.  With
                                      ILUPP_REFERENCE_NANS=1 in the environment this library returns exactly that factor instead */
    ILUPP_ERR_INTERNAL = -12,      /* an invariant of this build does not hold (a bug here, never a property of the input) */
    ILUPP_ERR_NOT_CONVERGED = -13, /* ilupp_hip_solve: binding.cpp:227 "did not converge" */
    ILUPP_ERR_DIAG_DROPPED = -11   /* ICholT: a finite pivot was dropped by the threshold or the top-k budget (dropping.hpp:8-34 does not
                                      protect it).  The reference keeps such a factor and solves with whatever entry comes first in
                                      the column; this build reports it instead (documented deviation) */
} ilupp_status;

/* binding.cpp:279  m.def("index_size") -> sizeof(Integer) */
int ilupp_hip_index_size(void);

/* message of the last failure on this thread ("" if none) */
const char *ilupp_hip_last_error(void);

/* device selection (default 0); one HIP stream per preconditioner object */
int ilupp_hip_set_device(int device);
int ilupp_hip_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Factories.  (data, indices, indptr, is_csr) exactly as the reference's make_matrix receives them
 * (binding.cpp:68-90): n = len(indptr)-1, nnz = indptr[n]; column indices sorted ascending per row
 * (the Python wrapper guarantees it, ilupp/__init__.py:70).
 * ------------------------------------------------------------------------------------------- */

/* binding.cpp:366-375  ILU0Preconditioner(data, indices, indptr, is_csr) -> GenericLUPreconditioner
 * (ILU0.hpp:69-106) */
int ilupp_hip_ilu0_create(const double *data, const int32_t *indices, const int32_t *indptr,
                          int32_t n, int is_csr, ilupp_precond **out);
int ilupp_hip_ilu0_create_device(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr,
                                 int32_t n, int is_csr, ilupp_precond **out);
/* ... for a caller that KNOWS the number of stored entries (the length of the arrays it holds -- what the reference reads as pointer[last],
 * sparse_implementation.h:3076-3089): nothing is read back before the construction starts when a matrix of this (n, nnz) was a box grid
 * before in this process (the dimensions are guessed again, grid.hip; the proof on the device covers indptr[n] == nnz and every row, and a
 * failed proof, or an (n, nnz) not seen before, takes the reading way of ilupp_hip_ilu0_create_device).  A wrong nnz is an error;
 * d_indices and d_data MUST hold at least nnz entries (the proof and the factor kernel address them with nnz as their bound before
 * indptr[n] has been compared with it). */
int ilupp_hip_ilu0_create_device_nnz(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr,
                                     int32_t n, int64_t nnz, int is_csr, ilupp_precond **out);

/* binding.cpp:299-310  ILUTPreconditioner.__init__(..., max_fill_in, threshold)  (ILUT.hpp:199-278) */
int ilupp_hip_ilut_create(const double *data, const int32_t *indices, const int32_t *indptr,
                          int32_t n, int is_csr, int32_t max_fill_in, double threshold, ilupp_precond **out);

/* binding.cpp:377-386  IChol0Preconditioner(...) -> GenericLLTPreconditioner (IChol.hpp:63-73) */
int ilupp_hip_ichol0_create(const double *data, const int32_t *indices, const int32_t *indptr,
                            int32_t n, int is_csr, ilupp_precond **out);

/* binding.cpp:388-397  ICholTPreconditioner(..., add_fill_in, threshold) (IChol.hpp:158-164) */
int ilupp_hip_icholt_create(const double *data, const int32_t *indices, const int32_t *indptr,
                            int32_t n, int is_csr, int32_t add_fill_in, double threshold, ilupp_precond **out);

/* SURVEY 8(f1).  binding.cpp:329-340  ILUCPreconditioner(A_data, A_indices, A_indptr, is_csr, max_fill_in, threshold)
 * -> preconditioner_implementation.h:940-958 -> ILUC2, ILUC.hpp:112-207 (Crout ILU of Li, Saad, Chow; dropping.hpp:8-34).
 * Two factors: #0 stored column-wise, #1 row-wise (binding.cpp:449-460: for COLUMN input the two are interchanged).
 * Errors: ILUPP_ERR_ZERO_PIVOT ("ILUC2: zero pivot on diagonal, k=..."), ILUPP_ERR_MEMORY (the reference's reservation of
 * min(max_fill_in n, 10 nnz) entries per factor exceeded).  The _device form borrows a matrix in HBM. */
int ilupp_hip_iluc_create(const double *data, const int32_t *indices, const int32_t *indptr,
                          int32_t n, int is_csr, int32_t max_fill_in, double threshold, ilupp_precond **out);
int ilupp_hip_iluc_create_device(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr,
                                 int32_t n, int is_csr, int32_t max_fill_in, double threshold, ilupp_precond **out);

/* The three above on a matrix that already lives in this GPU's HBM (borrowed for the duration of the call): what a
 * GPU-resident caller -- and bench.py's C3/C4 legs -- use, as ilupp_hip_ilu0_create_device does for ILU(0). */
int ilupp_hip_ilut_create_device(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr,
                                 int32_t n, int is_csr, int32_t max_fill_in, double threshold, ilupp_precond **out);
int ilupp_hip_ichol0_create_device(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr,
                                   int32_t n, int is_csr, ilupp_precond **out);
int ilupp_hip_icholt_create_device(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr,
                                   int32_t n, int is_csr, int32_t add_fill_in, double threshold, ilupp_precond **out);

void ilupp_hip_destroy(ilupp_precond *p);

/* ---------------------------------------------------------------------------------------------
 * Preconditioner object members (binding.cpp:233-264, wrapPreconditioner<P>)
 * ------------------------------------------------------------------------------------------- */

/* binding.cpp:237-245 apply(x): in place on a contiguous fp64 buffer of length `len` */
int ilupp_hip_apply(ilupp_precond *p, double *x, int64_t len);
/* binding.cpp:246-254 apply_trans(x) */
int ilupp_hip_apply_trans(ilupp_precond *p, double *x, int64_t len);
/* the same on a vector that already lives in HBM (asynchronous on the object's stream unless sync!=0) */
int ilupp_hip_apply_device(ilupp_precond *p, double *d_x, int64_t len, int transpose, int sync);

/* Stream ordering of the device-pointer entry points (*_create_device, ilupp_hip_ilu0_refactor_device,
 * ilupp_hip_apply_device).  Every object works on a private non-blocking HIP stream.  By default the caller must
 * have synchronised the producer of the device buffers before the call, and a call with sync=0 must be followed by
 * ilupp_hip_sync() before the result is read.  With a caller stream set for the calling thread (enable != 0;
 * hip_stream = the caller's hipStream_t, NULL = the legacy default stream), every such call is ordered after the work
 * already submitted to that stream, and an asynchronous apply makes that stream wait for the result. */
int ilupp_hip_set_caller_stream(void *hip_stream, int enable);

/* y = A x for a CSR matrix in HBM (row-major storage read as given; for a CSC matrix this is A^T x), on the caller's
 * stream: with ilupp_hip_apply_device the two halves of a GPU-resident Krylov iteration.  One lane per row, accumulation in
 * stored order from 0 (reference: sparse_implementation.h:2733-2760, ROW/ID branch) -- bit-identical to the reference. */
int ilupp_hip_spmv_device(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr, int32_t n, int64_t nnz,
                          const double *d_x, double *d_y, void *hip_stream);

/* binding.cpp:255  total_nnz  (conventions per class) */
int64_t ilupp_hip_total_nnz(const ilupp_precond *p);
/* binding.cpp:257-261 */
double ilupp_hip_memory_used_calculations(const ilupp_precond *p);
double ilupp_hip_memory_allocated_calculations(const ilupp_precond *p);
double ilupp_hip_memory(const ilupp_precond *p);
int ilupp_hip_exists(const ilupp_precond *p);
const char *ilupp_hip_special_info(const ilupp_precond *p);
/* binding.cpp:262 print_info() */
void ilupp_hip_print_info(const ilupp_precond *p);
/* pre_image_dimension(), used by the size check binding.cpp:241 */
int32_t ilupp_hip_dimension(const ilupp_precond *p);

/* ---------------------------------------------------------------------------------------------
 * Factor egress (binding.cpp:118-175 wrap_matrix / wrap_all_factor_matrices -> factors_info()).
 * LU objects expose [L, U], LL^T objects [L].
 * ------------------------------------------------------------------------------------------- */
int ilupp_hip_num_factors(const ilupp_precond *p);
int ilupp_hip_factor_info(const ilupp_precond *p, int which, int32_t *rows, int32_t *cols,
                          int64_t *nnz, int *is_csr);
/* copies the factor into caller-allocated host arrays of nnz / nnz / rows+1 elements */
int ilupp_hip_factor_copy(const ilupp_precond *p, int which, double *data, int32_t *indices, int32_t *indptr);
/* device pointers of a factor (valid while p lives); for GPU-resident callers */
int ilupp_hip_factor_device_ptrs(const ilupp_precond *p, int which, const double **d_data,
                                 const int32_t **d_indices, const int32_t **d_indptr);

/* ---------------------------------------------------------------------------------------------
 * the multilevel ILU++ preconditioner, binding.cpp:284-298
 *   MultilevelILUCDPPreconditioner(A_data, A_indices, A_indptr, is_csr, iluplusplus_precond_parameter)
 *   -> multilevelILUCDPPreconditioner::make_preprocessed_multilevelILUCDP (preconditioner_implementation.h:1350-1665),
 *      apply :433-488, total_nnz preconditioner.h:312.
 * Built: BOTH factorisations of make_preprocessed_multilevelILUCDP, chosen as the reference chooses (:1376-1382):
 *   - WITHOUT pivoting (use_ILUC: PERMUTE_ROWS 0 / 1, TOTAL_PIV off, piv_tol 0 -- precon_parameter 10 of parameters_implementation.h:927-934,
 *     e.g. default_configuration(1)): matrix_sparse::partialILUC (ILUCDP.hpp:1405-2231) as a dataflow computation over all CUs;
 *   - WITH pivoting (the default-constructed parameters, default_configuration(0), (10)): matrix_sparse::partialILUCDP (:268-1404), a chain
 *     of data-dependent steps walked by one wave; many matrices at once: ilupp_hip_ml_create_batch;
 * dropping by the combined weight of the standard / error-propagation / pivot rules, the inverse-based rule (ILUPP_DROP_INVERSE;
 * precon_parameter 1, 11) and the weighted rule (ILUPP_DROP_WEIGHTED / _WEIGHTED2; precon_parameter 2, 12) -- with the last two, whose
 * estimates are recurrences over all steps, the factorisation without pivoting runs as a chain as well (working rows up to 2048 entries in LDS, up to 32768 in global memory);
 * unbounded or bounded fill; levels ended by small pivots or by the fill of L (FINAL_ROW_CRIT -1 .. 9); preprocessing steps
 * NORMALIZE_COLUMNS, NORMALIZE_ROWS, PQ_ORDERING, MAX_WEIGHTED_MATCHING_ORDERING, UNIT_OR_ZERO_DIAGONAL_SCALING, SPARSE_FIRST_ORDERING,
 * DD_SYMM_MOVE_CORNER_ORDERING_IM, SYMM_PQ (sparse_implementation.h:5214-5460) in any sequence of at most 8.  Every other parameter
 * combination (the improved Schur complement, positional dropping, FINAL_ROW_CRIT
 * < -1, an external final row) is refused with ILUPP_ERR_UNSUPPORTED: nothing is silently replaced.
 * ------------------------------------------------------------------------------------------- */
typedef struct ilupp_ml ilupp_ml;

enum {                               /* preprocessing_type values (orderings.h) this build has */
    ILUPP_PRE_NORMALIZE_COLUMNS = 1,
    ILUPP_PRE_NORMALIZE_ROWS = 2,
    ILUPP_PRE_PQ_ORDERING = 3,
    ILUPP_PRE_MAX_WEIGHTED_MATCHING_ORDERING = 4,     /* the matching itself runs on the host (sequential augmenting paths) */
    ILUPP_PRE_DD_SYMM_MOVE_CORNER_ORDERING_IM = 5,    /* refused for matrices on which the reference's own result is undefined */
    ILUPP_PRE_UNIT_OR_ZERO_DIAGONAL_SCALING = 6,
    ILUPP_PRE_SPARSE_FIRST_ORDERING = 7,
    ILUPP_PRE_SYMM_PQ = 8                             /* rows and columns by sym_ddPQ's weights (sparse_implementation.h:4926-4940, :5352-5360) */
};

enum { ILUPP_DROP_STANDARD = 1, ILUPP_DROP_STANDARD2 = 2, ILUPP_DROP_ERR_PROP = 4, ILUPP_DROP_ERR_PROP2 = 8, ILUPP_DROP_PIVOT = 16,
       ILUPP_DROP_INVERSE = 32, ILUPP_DROP_WEIGHTED = 64, ILUPP_DROP_WEIGHTED2 = 128 };   /* USE_INVERSE_DROPPING (ILUCDP.hpp:680-713, :882-916), USE_WEIGHTED_DROPPING[2]
                                       (:629-631, :670-674): estimates that accumulate over the steps in their order -- the factorisation runs as a chain */

typedef struct {                     /* the fields of iluplusplus_precond_parameter (parameters.h:120-235) the built family reads */
    double threshold;                /* threshold */
    int32_t n_preprocessing;         /* PREPROCESSING: number of steps, */
    int32_t preprocessing[8];        /*                the steps (ILUPP_PRE_*) */
    double pq_threshold;             /* PQ_THRESHOLD */
    int32_t max_levels;              /* MAX_LEVELS */
    int32_t min_ml_size;             /* MIN_ML_SIZE */
    int32_t small_pivot_terminates;  /* SMALL_PIVOT_TERMINATES */
    double min_pivot;                /* MIN_PIVOT */
    double min_elim_factor;          /* MIN_ELIM_FACTOR */
    double threshold_shift_schur;    /* THRESHOLD_SHIFT_SCHUR */
    double vary_threshold_factor;    /* VARY_THRESHOLD_FACTOR */
    int32_t use_final_threshold;     /* USE_FINAL_THRESHOLD */
    double final_threshold;          /* FINAL_THRESHOLD */
    int32_t max_fill_in;             /* 0: MAX_FILLIN_IS_INF; else fill_in (entries a row of U / a column of L may have, the 1 included) */
    int32_t drop_rules;              /* ILUPP_DROP_*: USE_STANDARD_DROPPING, _DROPPING2, USE_ERR_PROP_DROPPING, _DROPPING2, USE_PIVOT_DROPPING */
    double weight_standard_drop, weight_standard_drop2, weight_err_prop_drop, weight_err_prop_drop2, weight_pivot_drop;   /* WEIGHT_* */
    int32_t combine_factor;          /* COMBINE_FACTOR */
    double neutral_element, min_weight;   /* NEUTRAL_ELEMENT, MIN_WEIGHT */
    int32_t scale_weight_invdiag;    /* SCALE_WEIGHT_INVDIAG */
    /* the factorisation WITH pivoting (reference partialILUCDP, ILUCDP.hpp:268-1404) is taken unless PERMUTE_ROWS is 0 or 1, total pivoting
     * is off (BEGIN_TOTAL_PIV 0 or TOTAL_PIV 0) and piv_tol is 0 (preconditioner_implementation.h:1376-1382): the reference's
     * default-constructed parameters select it */
    double piv_tol;                      /* piv_tol */
    int32_t permute_rows;                /* PERMUTE_ROWS 0..3 */
    int32_t total_piv;                   /* TOTAL_PIV 0..2 */
    int32_t begin_total_piv;             /* BEGIN_TOTAL_PIV */
    int32_t final_row_crit;              /* FINAL_ROW_CRIT -1..9 */
    double move_level_factor;            /* MOVE_LEVEL_FACTOR */
    double row_u_max;                    /* ROW_U_MAX */
    double weight_inverse_drop;          /* WEIGHT_INVERSE_DROP (with ILUPP_DROP_INVERSE) */
    double weight_weighted_drop;         /* WEIGHT_WEIGHTED_DROP (with ILUPP_DROP_WEIGHTED) */
    double init_weights_lu;              /* INIT_WEIGHTS_LU */
} ilupp_ml_params;

/* default_configuration(1) (parameters_implementation.h:546-549: set_PQ + precon_parameter 10) with threshold 0 */
void ilupp_hip_ml_default_params(ilupp_ml_params *p);
int ilupp_hip_ml_create(const double *data, const int32_t *indices, const int32_t *indptr, int32_t n, int is_csr,
                        const ilupp_ml_params *params, ilupp_ml **out);
int ilupp_hip_ml_create_device(const double *d_data, const int32_t *d_indices, const int32_t *d_indptr, int32_t n, int is_csr,
                               const ilupp_ml_params *params, ilupp_ml **out);
/* BASELINE config 5's batched shape: `count` independent matrices (host arrays, all CSR or all CSC), one preconditioner each -- what a
 * caller of the reference does by calling the constructor once per matrix (preconditioner_implementation.h:1483-1494 runs once per
 * matrix).  The constructions run side by side (one host thread and HIP stream per matrix, at most ILUPP_BATCH_WORKERS = 64 at a
 * time), and the sequential chains of the factorisation with pivoting -- one wave each -- are launched TOGETHER, one workgroup per
 * matrix, so that a batch costs about what its slowest member costs.  out[i] / status[i] per matrix (status may be NULL); returns the
 * first error.  Every result is identical to what ilupp_hip_ml_create gives for that matrix. */
int ilupp_hip_ml_create_batch(int32_t count, const double *const *data, const int32_t *const *indices, const int32_t *const *indptr, const int32_t *n,
                              int is_csr, const ilupp_ml_params *params, ilupp_ml **out, int32_t *status);
void ilupp_hip_ml_destroy(ilupp_ml *p);
/* binding.cpp:237-254 apply / apply_trans, in place on a host vector; the _device form on a vector in HBM (sync as above) */
int ilupp_hip_ml_apply(ilupp_ml *p, double *x, int64_t len, int transpose);
int ilupp_hip_ml_apply_device(ilupp_ml *p, double *d_x, int64_t len, int transpose, int sync);
/* one half of the split preconditioner on a vector in HBM: apply_preconditioner_left / _right (preconditioner_implementation.h:441-453,
 * :468-486), what the reference's solver loop uses with SPLIT preconditioning (solving_routines_implementation.h:81); left != 0: the left part */
int ilupp_hip_ml_apply_part_device(ilupp_ml *p, double *d_x, int64_t len, int transpose, int left, int sync);
int ilupp_hip_ml_sync(ilupp_ml *p);
/* levels() (preconditioner.h:298), total_nnz (:312), dim(k) */
int32_t ilupp_hip_ml_levels(const ilupp_ml *p);
int64_t ilupp_hip_ml_total_nnz(const ilupp_ml *p);
/* one level (extract_left_matrix(k) ... extract_right_scaling(k), preconditioner.h:288-296): sizes, then copies to host buffers
 * (any pointer may be NULL): L by columns, U by rows (data / indices of nnz entries, indptr of n + 1), the middle diagonal,
 * the four permutations and the two scalings, n entries each */
int ilupp_hip_ml_level_info(const ilupp_ml *p, int32_t level, int32_t *n, int64_t *nnz_left, int64_t *nnz_right);
int ilupp_hip_ml_level_copy(const ilupp_ml *p, int32_t level, double *l_data, int32_t *l_indices, int32_t *l_indptr, double *u_data,
                            int32_t *u_indices, int32_t *u_indptr, double *middle, int32_t *perm_rows, int32_t *perm_cols,
                            int32_t *inv_perm_rows, int32_t *inv_perm_cols, double *d_left, double *d_right);
/* GPU milliseconds of the construction (whole, and the factorisation kernels alone) and of the last apply */
int ilupp_hip_ml_timings(const ilupp_ml *p, float *construct_ms, float *kernel_ms, float *last_apply_ms);

/* _ilupp.solve (binding.cpp:200-230, bound at :281): the multilevel preconditioner of `params` is built for A, then BiCGstab with SPLIT
 * preconditioning runs from the zero vector (solve_with_multilevel_preconditioner, solving_routines_implementation.h:81 -> bicgstab,
 * iterative_solvers_implementation.h:385-530) until res / initial_res <= rtol and res <= atol, or max_iter iterations (at least one).
 * A, rhs and x are host arrays (x: n doubles, written in every case that reaches the iteration); the matrix, the preconditioner and
 * every vector of the iteration live in HBM, the host sees one residual norm per iteration.  *iterations, *rel_reached (res /
 * initial_res) and *abs_reached (res) are what the reference returns as (max_iter, 10^-rel_tol, 10^-abs_tol).
 * Returns ILUPP_ERR_NOT_CONVERGED where the reference throws "did not converge", ILUPP_ERR_WRONG_SIZE for "right-hand side has wrong
 * size!" (:209-210), and the errors of ilupp_hip_ml_create. */
int ilupp_hip_solve(const double *data, const int32_t *indices, const int32_t *indptr, int32_t n, int is_csr, const double *rhs, int64_t rhs_len,
                    double rtol, double atol, int32_t max_iter, const ilupp_ml_params *params, double *x, int32_t *iterations, double *rel_reached,
                    double *abs_reached);

/* ---------------------------------------------------------------------------------------------
 * ILUCP: Crout ILU with column pivoting.  Replaces binding.cpp:343-356 (ILUCPPreconditioner.__init__ ->
 * preconditioner_implementation.h:1117-1147 -> ILUCP4, ILUC.hpp:212-370) and its apply (triangular_solve_perm,
 * sparse_implementation.h:4166-4253).  A chain of n data-dependent steps: one wave of the GPU walks it (ilucp.hip).
 * Errors: ILUPP_ERR_MEMORY ("ILUCP4: Insufficient memory reserved. Increase mem_factor", ILUC.hpp:287-289, :344-346).
 * ------------------------------------------------------------------------------------------- */
typedef struct ilupp_ilucp ilupp_ilucp;
int ilupp_hip_ilucp_create(const double *data, const int32_t *indices, const int32_t *indptr, int32_t n, int is_csr, int32_t max_fill_in,
                           double threshold, double piv_tol, int32_t row_pos, double mem_factor, ilupp_ilucp **out);
void ilupp_hip_ilucp_destroy(ilupp_ilucp *p);
/* binding.cpp:237-254 apply / apply_trans, in place on a host vector */
int ilupp_hip_ilucp_apply(ilupp_ilucp *p, double *x, int64_t len, int transpose);
int64_t ilupp_hip_ilucp_total_nnz(const ilupp_ilucp *p);
int32_t ilupp_hip_ilucp_zero_pivots(const ilupp_ilucp *p);
/* the factors as ILUCP4 returns them for the major-order view of the input (L by columns with its 1 first; U by rows with the pivot first
 * and the original column indices) and the permutation (binding.cpp:178-196 permutations(): the right one for COLUMN input, the left one
 * for ROW input); any pointer may be NULL */
int ilupp_hip_ilucp_info(const ilupp_ilucp *p, int32_t *n, int64_t *nnz_l, int64_t *nnz_u, float *kernel_ms);
int ilupp_hip_ilucp_copy(const ilupp_ilucp *p, double *l_data, int32_t *l_indices, int32_t *l_indptr, double *u_data, int32_t *u_indices,
                         int32_t *u_indptr, int32_t *perm);

/* ILUTP: ILUT with column pivoting.  Replaces binding.cpp:313-326 (ILUTPPreconditioner.__init__ -> preconditioner_implementation.h:1050-1078 ->
 * ILUTP2, ILUTP.hpp:13-140).  The object is of the same type as ILUCP's and shares its apply / total_nnz / info / copy / destroy entry points;
 * its factors belong to the ROWS of the view: L by rows (its 1 last, columns in the permuted numbering), U by rows (the pivot first, original
 * column indices, ordered by permuted position).  Errors: ILUPP_ERR_MEMORY ("ILUTP2: memory reserved was insufficient."), ILUPP_ERR_ZERO_PIVOT
 * ("matrix_sparse::ILUTP2: encountered zero pivot in row N") */
int ilupp_hip_ilutp_create(const double *data, const int32_t *indices, const int32_t *indptr, int32_t n, int is_csr, int32_t max_fill_in,
                           double threshold, double piv_tol, int32_t row_pos, double mem_factor, ilupp_ilucp **out);

/* ---------------------------------------------------------------------------------------------
 * Measurement hooks used by bench.py (not part of the reference's surface).
 * Times are GPU milliseconds from hipEvents recorded on the object's stream.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float analysis_ms;   /* symbolic pass: diagonal positions, L/U pattern split, row scheduling */
    float numeric_ms;    /* numeric factorisation kernel(s) */
    float last_apply_ms; /* most recent apply / apply_trans on device data */
    float numeric_kernel_ms;  /* the dominant numeric kernel alone */
    float lsolve_kernel_ms;   /* forward-solve kernel of the last apply */
    float usolve_kernel_ms;   /* backward-solve kernel of the last apply */
} ilupp_timings;
int ilupp_hip_get_timings(const ilupp_precond *p, ilupp_timings *t);
/* which kernel family built this object ("ilu0:static-direct", "ilu0:static-level-major", "ilu0:level-order",
 * "ilu0:csr-program", "ilu0:csr", "ilut", "ichol0", "icholt"): bench.py names the kernel its roofline line is about */
const char *ilupp_hip_path(const ilupp_precond *p);
/* how the row blocks of an ILU(0) object were found: "grid" (the pattern is a lexicographic box-grid stencil: guessed from row 0, proven
 * for every row by one streaming pass next to the lane-table kernels, grid.hip) or "general" (the pass over the pattern that finds the
 * chains of any matrix, symbolic.hip); "" for other objects.  Measurement / test hook, no counterpart in binding.cpp. */
const char *ilupp_hip_analysis_path(const ilupp_precond *p);
/* diagnostics (tests compare the closed-form tables of grid.hip with the ones the general kernels make): table `which` of a static ILU(0)
 * object as 32-bit words -- 0 / 1 lane tables (forward / backward schedule), 2 / 3 chunk tables, 4 backward right-hand-side map, 5 rows-pass
 * records, 6 / 7 export ordinals, 8 / 9 exchange layout, 10 forward -> backward slots, 11 / 12 skews; returns the words copied (<= cap), -1
 * where there is no such table.  No counterpart in binding.cpp. */
long long ilupp_hip_debug_static_table(ilupp_precond *p, int which, int32_t *out, long long cap);
/* measurement hook: the kernels a static ILU(0) object runs, "factor;forward sweep;backward sweep"; for an LL^T object (IChol0, ICholT) its
 * factor kernel and, once an apply has built them, the sweeps of its factor pair; "" otherwise.  No counterpart in binding.cpp */
const char *ilupp_hip_kernel_names(const ilupp_precond *p);
/* redo the numeric phase on (possibly new) values with the SAME pattern (buffers reused); times it */
int ilupp_hip_ilu0_refactor_device(ilupp_precond *p, const double *d_data, const int32_t *d_indices,
                                   const int32_t *d_indptr);
/* wait for asynchronous applies (ilupp_hip_apply_device with sync=0) and report their status */
int ilupp_hip_sync(ilupp_precond *p);
/* The library keeps freed device buffers (up to a limit, the oldest go first) for the next construction: a factorisation that is
 * repeated asks for the same sizes again and hipMalloc / hipFree of multi-GB buffers cost more than the kernels.  This hands them
 * back to the driver (no counterpart in binding.cpp: the reference's buffers are host memory). */
int ilupp_hip_release_cached_memory(void);
/* the limit of that cache in bytes (default 48 GiB, or ILUPP_CACHE_LIMIT_MB; 0 = keep nothing); blocks over the new limit are
 * handed back at once.  A process that shares the GPU with other allocators sets this before its first factorisation. */
int ilupp_hip_set_cache_limit(unsigned long long bytes);
/* bytes currently kept in the cache / blocks currently handed out (diagnostics; a leak shows as blocks that never come back) */
unsigned long long ilupp_hip_cached_bytes(void);
unsigned long long ilupp_hip_live_blocks(void);

#ifdef __cplusplus
}
#endif
#endif
