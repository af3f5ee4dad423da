/*
 * oracle/ref_shim.cpp -- C-ABI shim over the REAL reference (c-f-h/ilupp), TEST INFRASTRUCTURE ONLY.
 *
 * Compiled by oracle/Makefile against the reference headers where they lie (-I/root/reference/src);
 * no reference source is copied into this repository.  The resulting oracle/_ref/libilupp_ref.so is
 * used (1) to pin oracle/ilupp_oracle.c, (2) to emit tests/golden/*.npz, (3) as the "reference" CPU
 * baseline of bench.py.  It calls exactly the functions the reference's own binding calls
 * (src/binding.cpp:366-447) and the same preconditioner classes for apply (binding.cpp:237-254).
 *
 * The exported signatures are identical to oracle/ilupp_oracle.h (prefix ref_ instead of orc_).
 */
#include "ilupp/iluplusplus_interface.cpp"

#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "ilupp_oracle.h"

using namespace iluplusplus;

namespace {

void export_mat(const matrix &M, orc_mat *out)
{
    const Integer n = M.rows();
    const Integer nnz = M.actual_non_zeroes();
    out->n = n;
    out->nnz = nnz;
    out->is_csr = (M.orient() == ROW) ? 1 : 0;
    out->ptr = (orc_int *)std::malloc(sizeof(orc_int) * (size_t)(n + 1));
    out->idx = (orc_int *)std::malloc(sizeof(orc_int) * (size_t)(nnz > 0 ? nnz : 1));
    out->val = (double *)std::malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    for (Integer i = 0; i <= n; ++i) out->ptr[i] = M.read_pointer(i);
    for (Integer i = 0; i < nnz; ++i) { out->idx[i] = M.read_index(i); out->val[i] = M.read_data(i); }
}

matrix view(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr)
{
    return matrix(const_cast<double *>(val), const_cast<Integer *>(idx), const_cast<Integer *>(ptr),
                  n, n, is_csr ? ROW : COLUMN, true);
}

}  // namespace

extern "C" {

void ref_free_mat(orc_mat *M)
{
    if (!M) return;
    std::free(M->ptr); std::free(M->idx); std::free(M->val);
    M->ptr = M->idx = nullptr; M->val = nullptr;
}

int ref_ilu0(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_mat *Lo, orc_mat *Uo)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    matrix L, U;
    ILU0(A, L, U);                                   // binding.cpp:421-430
    export_mat(L, Lo); export_mat(U, Uo);
    return ORC_OK;
}

int ref_ilut(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_int max_fill_in, double threshold, orc_mat *Lo, orc_mat *Uo, orc_int *err_row)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    matrix L, U;
    Real time;
    try {
        ILUT_heap(A, L, U, max_fill_in, threshold, time);   // binding.cpp:432-447
    } catch (const std::runtime_error &e) {
        const char *p = std::strstr(e.what(), "row ");
        if (err_row) *err_row = p ? (orc_int)std::atoi(p + 4) : -1;
        return ORC_ERR_ZERO_PIVOT;
    }
    if (!is_csr) {
        L.interchange(U);
        L.transpose_in_place();
        U.transpose_in_place();
    }
    export_mat(L, Lo); export_mat(U, Uo);
    return ORC_OK;
}

int ref_ichol0(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr, orc_mat *Lo)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    try {
        matrix L = IChol0(A);                        // binding.cpp:399-408
        export_mat(L, Lo);
    } catch (const std::logic_error &) {
        return ORC_ERR_NOT_TRIANGULAR;
    }
    return ORC_OK;
}

int ref_icholt(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
               orc_int add_fill_in, double threshold, orc_mat *Lo)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    try {
        matrix L = ICholT(A, add_fill_in, threshold);   // binding.cpp:410-419
        export_mat(L, Lo);
    } catch (const std::logic_error &) {
        return ORC_ERR_NOT_TRIANGULAR;
    } catch (const std::runtime_error &) {
        return ORC_ERR_MEMORY;
    }
    return ORC_OK;
}

int ref_iluc(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_int max_fill_in, double threshold, orc_mat *Lo, orc_mat *Uo, orc_int *err_row)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    matrix L, U;
    if (err_row) *err_row = -1;
    try {
        ILUC2(A, L, U, max_fill_in, threshold);     // binding.cpp:449-460
    } catch (const std::runtime_error &e) {
        const char *p = std::strstr(e.what(), "k=");
        if (p) { if (err_row) *err_row = (orc_int)std::atoi(p + 2); return ORC_ERR_ZERO_PIVOT; }
        return ORC_ERR_MEMORY;                      // "append_row_with_prefix: insufficient memory reserved"
    }
    if (!is_csr) L.interchange(U);
    export_mat(L, Lo); export_mat(U, Uo);
    return ORC_OK;
}

void ref_trisolve(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
                  int form, int use, double *x)
{
    matrix M = view(n, ptr, idx, val, is_csr);
    vector v(n, x, true);
    M.triangular_solve(form == ORC_LOWER ? LOWER_TRIANGULAR : UPPER_TRIANGULAR,
                       use == ORC_ID ? ID : TRANSPOSE, v);
}

/* apply through the reference's own preconditioner classes, as binding.cpp:237-254 does */
void ref_apply_lu(const orc_mat *L, const orc_mat *U, int use, double *x)
{
    matrix Lm = view(L->n, L->ptr, L->idx, L->val, L->is_csr);
    matrix Um = view(U->n, U->ptr, U->idx, U->val, U->is_csr);
    matrix Lc(Lm), Uc(Um);
    indirect_split_triangular_preconditioner<Real, matrix, vector> P(std::move(Lc), LOWER_TRIANGULAR, std::move(Uc), UPPER_TRIANGULAR);
    vector v(L->n, x, true);
    P.apply_preconditioner_only(use == ORC_ID ? ID : TRANSPOSE, v);
}

void ref_apply_llt(const orc_mat *L, int use, double *x)
{
    matrix Lm = view(L->n, L->ptr, L->idx, L->val, L->is_csr);
    matrix Lc(Lm);
    indirect_split_triangular_symmetric_preconditioner<Real, matrix, vector> P(std::move(Lc), LOWER_TRIANGULAR);
    vector v(L->n, x, true);
    P.apply_preconditioner_only(use == ORC_ID ? ID : TRANSPOSE, v);
}

/* total_nnz conventions as the Python-visible objects report them */
orc_int ref_total_nnz_lu_generic(const orc_mat *L, const orc_mat *U)
{
    matrix Lm = view(L->n, L->ptr, L->idx, L->val, L->is_csr);
    matrix Um = view(U->n, U->ptr, U->idx, U->val, U->is_csr);
    matrix Lc(Lm), Uc(Um);
    indirect_split_triangular_preconditioner<Real, matrix, vector> P(std::move(Lc), LOWER_TRIANGULAR, std::move(Uc), UPPER_TRIANGULAR);
    return P.total_nnz();
}

orc_int ref_total_nnz_ilut(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
                           orc_int max_fill_in, double threshold)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    try {
        ILUTPreconditioner<Real, matrix, vector> P(A, max_fill_in, threshold);
        return P.total_nnz();
    } catch (...) {
        return -1;
    }
}

/* full ILUT preconditioner path (ctor + apply), preconditioner_implementation.h:992-1011 */
int ref_ilut_precond_apply(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
                           orc_int max_fill_in, double threshold, int use, double *x)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    try {
        ILUTPreconditioner<Real, matrix, vector> P(A, max_fill_in, threshold);
        vector v(n, x, true);
        P.apply_preconditioner_only(use == ORC_ID ? ID : TRANSPOSE, v);
    } catch (...) {
        return ORC_ERR_ZERO_PIVOT;
    }
    return ORC_OK;
}

/* the reference's BiCGstab (iterative_solvers_implementation.h:385-530) with an ILU(0) preconditioner applied from the LEFT,
 * exactly `iters` iterations from the zero start vector: x receives the iterate (pins ilupp_amd.device.bicgstab) */
int ref_bicgstab_ilu0_left(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
                           const double *b, orc_int iters, double *x)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    matrix L, U;
    ILU0(A, L, U);
    indirect_split_triangular_preconditioner<Real, matrix, vector> P(std::move(L), LOWER_TRIANGULAR, std::move(U), UPPER_TRIANGULAR);
    vector bv(n, const_cast<double *>(b), true);
    vector xv(n, 0.0);
    Integer max_iter = iters;
    Real rel_tol = 300.0, abs_tol = 300.0;            // (-log10 of the tolerances: never reached)
    bicgstab<Real, matrix, vector>(P, LEFT, A, bv, xv, iters, max_iter, rel_tol, abs_tol, true);
    if (xv.dimension() != n) return ORC_ERR_MEMORY;
    for (orc_int i = 0; i < n; ++i) x[i] = xv[i];
    return ORC_OK;
}

/* the multilevel ILU++ preconditioner (binding.cpp:284-298 -> preconditioner_implementation.h:1350-1665) with the parameters of
 * iluplusplus_precond_parameter::default_configuration(config) (parameters_implementation.h:538-609), threshold and (fill_in >= 0)
 * fill_in set as ilupp/__init__.py:171-203 does.  Emits golden vectors for SURVEY section 8 row f3: apply(x) in place, the number
 * of levels, total_nnz. */
int ref_ilupp_apply(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
                    orc_int config, double threshold, orc_int fill_in, int use, double *x, orc_int *levels, orc_int *total_nnz)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    iluplusplus_precond_parameter param;
    if (config >= 0) param.default_configuration(config);
    param.set_threshold(threshold);
    if (fill_in >= 0) param.set_fill_in(fill_in);
    multilevelILUCDPPreconditioner<Real, matrix, vector> P;
    try {
        P.make_preprocessed_multilevelILUCDP(A, param);
    } catch (...) {
        return ORC_ERR_ZERO_PIVOT;
    }
    if (!P.exists()) return ORC_ERR_ZERO_PIVOT;
    if (levels) *levels = P.levels();
    if (total_nnz) *total_nnz = P.total_nnz();
    vector v(n, x, true);
    P.apply_preconditioner_only(use == ORC_ID ? ID : TRANSPOSE, v);
    return ORC_OK;
}


/* ---- the multilevel preconditioner of binding.cpp:284-298 behind the orc_ml_* ABI (prefix ref_): parameters = default_parameters
 * + init(sequence, 10) (parameters_implementation.h:832-934) with the knobs of orc_ml_params set through the reference's own setters ---- */
struct ref_ml {
    multilevelILUCDPPreconditioner<Real, matrix, vector> P;
    orc_int n;
    std::vector<orc_mat> Ls, Us;
    std::vector<std::vector<double>> D, Dl, Dr;
    std::vector<std::vector<orc_int>> pr, pc, ipr, ipc;
};

/* the parameter object of a case: default_parameters + init(sequence, 10) with the knobs of orc_ml_params set through the reference's setters */
static int param_of(const orc_ml_params *IP, iluplusplus_precond_parameter &param)
{
    preprocessing_sequence seq;
    seq.resize(IP->n_preprocessing);
    for (int i = 0; i < IP->n_preprocessing; ++i) {
        switch (IP->preprocessing[i]) {
        case ORC_PRE_NORMALIZE_COLUMNS: seq.set(i) = NORMALIZE_COLUMNS; break;
        case ORC_PRE_NORMALIZE_ROWS: seq.set(i) = NORMALIZE_ROWS; break;
        case ORC_PRE_PQ_ORDERING: seq.set(i) = PQ_ORDERING; break;
        case ORC_PRE_MAX_WEIGHTED_MATCHING_ORDERING: seq.set(i) = MAX_WEIGHTED_MATCHING_ORDERING; break;
        case ORC_PRE_DD_SYMM_MOVE_CORNER_ORDERING_IM: seq.set(i) = DD_SYMM_MOVE_CORNER_ORDERING_IM; break;
        case ORC_PRE_UNIT_OR_ZERO_DIAGONAL_SCALING: seq.set(i) = UNIT_OR_ZERO_DIAGONAL_SCALING; break;
        case ORC_PRE_SPARSE_FIRST_ORDERING: seq.set(i) = SPARSE_FIRST_ORDERING; break;
        case ORC_PRE_SYMM_PQ: seq.set(i) = SYMM_PQ; break;
        default: return ORC_ERR_UNSUPPORTED;
        }
    }
    param.init(seq, 10, "");
    param.set_threshold(IP->threshold);
    param.set_PQ_THRESHOLD(IP->pq_threshold);
    param.set_MAX_LEVELS(IP->max_levels);
    param.set_MIN_ML_SIZE(IP->min_ml_size);
    param.set_SMALL_PIVOT_TERMINATES(IP->small_pivot_terminates != 0);
    param.set_MIN_PIVOT(IP->min_pivot);
    param.set_MIN_ELIM_FACTOR(IP->min_elim_factor);
    param.set_THRESHOLD_SHIFT_SCHUR(IP->threshold_shift_schur);
    param.set_VARY_THRESHOLD_FACTOR(IP->vary_threshold_factor);
    param.set_USE_FINAL_THRESHOLD(IP->use_final_threshold != 0);
    param.set_FINAL_THRESHOLD(IP->final_threshold);
    if (IP->max_fill_in > 0) { param.set_MAX_FILLIN_IS_INF(false); param.set_fill_in(IP->max_fill_in); }
    param.set_USE_STANDARD_DROPPING((IP->drop_rules & ORC_DROP_STANDARD) != 0);
    param.set_USE_STANDARD_DROPPING2((IP->drop_rules & ORC_DROP_STANDARD2) != 0);
    param.set_USE_ERR_PROP_DROPPING((IP->drop_rules & ORC_DROP_ERR_PROP) != 0);
    param.set_USE_ERR_PROP_DROPPING2((IP->drop_rules & ORC_DROP_ERR_PROP2) != 0);
    param.set_USE_PIVOT_DROPPING((IP->drop_rules & ORC_DROP_PIVOT) != 0);
    param.set_USE_INVERSE_DROPPING((IP->drop_rules & ORC_DROP_INVERSE) != 0);
    param.set_WEIGHT_INVERSE_DROP(IP->weight_inverse_drop);
    param.set_USE_WEIGHTED_DROPPING((IP->drop_rules & ORC_DROP_WEIGHTED) != 0);
    param.set_USE_WEIGHTED_DROPPING2((IP->drop_rules & ORC_DROP_WEIGHTED2) != 0);
    param.set_WEIGHT_WEIGHTED_DROP(IP->weight_weighted_drop);
    param.set_INIT_WEIGHTS_LU(IP->init_weights_lu);
    param.set_WEIGHT_STANDARD_DROP(IP->weight_standard_drop); param.set_WEIGHT_STANDARD_DROP2(IP->weight_standard_drop2);
    param.set_WEIGHT_ERR_PROP_DROP(IP->weight_err_prop_drop); param.set_WEIGHT_ERR_PROP_DROP2(IP->weight_err_prop_drop2);
    param.set_WEIGHT_PIVOT_DROP(IP->weight_pivot_drop);
    param.set_COMBINE_FACTOR(IP->combine_factor); param.set_NEUTRAL_ELEMENT(IP->neutral_element); param.set_MIN_WEIGHT(IP->min_weight);
    param.set_SCALE_WEIGHT_INVDIAG(IP->scale_weight_invdiag != 0);
    param.set_piv_tol(IP->piv_tol);
    param.set_PERMUTE_ROWS(IP->permute_rows);
    param.set_TOTAL_PIV(IP->total_piv);
    param.set_BEGIN_TOTAL_PIV(IP->begin_total_piv != 0);
    param.set_FINAL_ROW_CRIT(IP->final_row_crit);
    param.set_MOVE_LEVEL_FACTOR(IP->move_level_factor);
    param.set_ROW_U_MAX(IP->row_u_max);
    return ORC_OK;
}

int ref_ml_create(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr, const orc_ml_params *IP, ref_ml **out)
{
    *out = nullptr;
    matrix A = view(n, ptr, idx, val, is_csr);
    iluplusplus_precond_parameter param;
    { const int rc = param_of(IP, param); if (rc) return rc; }
    ref_ml *R = new ref_ml;
    R->n = n;
    try {
        R->P.make_preprocessed_multilevelILUCDP(A, param);
    } catch (...) {
        delete R;
        return ORC_ERR_ZERO_PIVOT;
    }
    if (!R->P.exists()) { delete R; return ORC_ERR_ZERO_PIVOT; }
    const int nl = R->P.levels();
    R->Ls.resize(nl); R->Us.resize(nl); R->D.resize(nl); R->Dl.resize(nl); R->Dr.resize(nl);
    R->pr.resize(nl); R->pc.resize(nl); R->ipr.resize(nl); R->ipc.resize(nl);
    for (int k = 0; k < nl; ++k) {
        export_mat(R->P.extract_left_matrix(k), &R->Ls[k]);
        export_mat(R->P.extract_right_matrix(k), &R->Us[k]);
        const vector &d = R->P.extract_middle_matrix(k);
        const Integer m = d.dimension();
        R->D[k].resize(m); R->Dl[k].resize(m); R->Dr[k].resize(m); R->pr[k].resize(m); R->pc[k].resize(m); R->ipr[k].resize(m); R->ipc[k].resize(m);
        for (Integer i = 0; i < m; ++i) {
            R->D[k][i] = d[i];
            R->Dl[k][i] = R->P.extract_left_scaling(k)[i];
            R->Dr[k][i] = R->P.extract_right_scaling(k)[i];
            R->pr[k][i] = R->P.extract_permutation_rows(k)[i];
            R->pc[k][i] = R->P.extract_permutation_columns(k)[i];
            R->ipr[k][i] = R->P.extract_inverse_permutation_rows(k)[i];
            R->ipc[k][i] = R->P.extract_inverse_permutation_columns(k)[i];
        }
    }
    *out = R;
    return ORC_OK;
}

int ref_ml_levels(const ref_ml *R) { return R->P.levels(); }
orc_int ref_ml_total_nnz(const ref_ml *R) { return R->P.total_nnz(); }

int ref_ml_level(const ref_ml *R, int k, orc_ml_level_view *v)
{
    if (k < 0 || k >= R->P.levels()) return ORC_ERR_UNSUPPORTED;
    v->n = (orc_int)R->D[k].size();
    v->L = R->Ls[k]; v->U = R->Us[k]; v->D = R->D[k].data();
    v->perm_rows = R->pr[k].data(); v->perm_cols = R->pc[k].data(); v->inv_perm_rows = R->ipr[k].data(); v->inv_perm_cols = R->ipc[k].data();
    v->D_l = R->Dl[k].data(); v->D_r = R->Dr[k].data();
    v->zero_pivots = R->P.zero_pivots_encountered(k);
    return ORC_OK;
}

void ref_ml_apply(const ref_ml *R, int use, double *x)
{
    vector v(R->n, x, true);
    R->P.apply_preconditioner_only(use == ORC_ID ? ID : TRANSPOSE, v);
}

void ref_ml_free(ref_ml *R)
{
    if (!R) return;
    for (auto &m : R->Ls) ref_free_mat(&m);
    for (auto &m : R->Us) ref_free_mat(&m);
    delete R;
}

/* libstdc++'s own std::sort with the comparator of dropping.hpp:25-26, to pin orc_sort_slots_by_abs_desc */
/* ---- ILUCPPreconditioner (binding.cpp:343-356) behind the orc_ilucp / orc_apply_ilucp ABI: the factors as ILUCP4 returned them for the
 * major-order view (for ROW input the class holds them transposed in place: the arrays are the same), the permutation, the apply ---- */
struct ref_ilucp_obj { ILUCPPreconditioner<Real, matrix, vector> *P; };

int ref_ilucp(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr, orc_int max_fill_in, double threshold, double piv_tol,
              orc_int rp, double mem_factor, orc_mat *Lo, orc_mat *Uo, orc_int *perm, orc_int *zero_pivots, void **handle)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    ILUCPPreconditioner<Real, matrix, vector> *P = nullptr;
    try {
        P = new ILUCPPreconditioner<Real, matrix, vector>(A, max_fill_in, threshold, piv_tol, rp, mem_factor);
    } catch (const std::runtime_error &) {
        return ORC_ERR_MEMORY;                      // "ILUCP4: Insufficient memory reserved. Increase mem_factor"
    }
    if (!P->exists()) { delete P; return ORC_ERR_ZERO_PIVOT; }
    // COLUMN input: left = L (by columns), right = U (by rows); ROW input: left = U^T (labelled COLUMN), right = L^T (labelled ROW)
    if (!is_csr) { export_mat(P->left_matrix(), Lo); export_mat(P->right_matrix(), Uo); }
    else { export_mat(P->right_matrix(), Lo); export_mat(P->left_matrix(), Uo); }
    for (orc_int k = 0; k < n; ++k) perm[k] = P->extract_permutation()[k];
    if (zero_pivots) *zero_pivots = -1;            // (private in the class)
    if (handle) *handle = P; else delete P;
    return ORC_OK;
}

void ref_ilucp_apply(void *handle, orc_int n, int use, double *x)
{
    auto *P = static_cast<ILUCPPreconditioner<Real, matrix, vector> *>(handle);
    vector v(n, x, true);
    P->apply_preconditioner_only(use == ORC_ID ? ID : TRANSPOSE, v);
}

void ref_ilucp_free(void *handle) { delete static_cast<ILUCPPreconditioner<Real, matrix, vector> *>(handle); }

/* ---- ILUTPPreconditioner (binding.cpp:313-326) behind the orc_ilutp / orc_apply_ilutp ABI ---- */
int ref_ilutp(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr, orc_int max_fill_in, double threshold, double piv_tol,
              orc_int bp, double mem_factor, orc_mat *Lo, orc_mat *Uo, orc_int *perm, orc_int *zero_pivots, void **handle)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    ILUTPPreconditioner<Real, matrix, vector> *P = nullptr;
    try {
        P = new ILUTPPreconditioner<Real, matrix, vector>(A, max_fill_in, threshold, piv_tol, bp, mem_factor);
    } catch (const std::runtime_error &e) {
        return std::strstr(e.what(), "zero pivot") ? ORC_ERR_ZERO_PIVOT : ORC_ERR_MEMORY;
    }
    if (!P->exists()) { delete P; return ORC_ERR_ZERO_PIVOT; }
    // ROW input: left = L, right = U; COLUMN input: left = U^T (U's arrays), right = L^T (L's arrays)
    if (is_csr) { export_mat(P->left_matrix(), Lo); export_mat(P->right_matrix(), Uo); }
    else { export_mat(P->right_matrix(), Lo); export_mat(P->left_matrix(), Uo); }
    for (orc_int k = 0; k < n; ++k) perm[k] = P->extract_permutation()[k];
    if (zero_pivots) *zero_pivots = -1;
    if (handle) *handle = P; else delete P;
    return ORC_OK;
}

void ref_ilutp_apply(void *handle, orc_int n, int use, double *x)
{
    auto *P = static_cast<ILUTPPreconditioner<Real, matrix, vector> *>(handle);
    vector v(n, x, true);
    P->apply_preconditioner_only(use == ORC_ID ? ID : TRANSPOSE, v);
}

void ref_ilutp_free(void *handle) { delete static_cast<ILUTPPreconditioner<Real, matrix, vector> *>(handle); }

void ref_sort_slots_by_abs_desc(orc_int *list, orc_int len, const double *key)
{
    std::sort(list, list + len, [&](orc_int x, orc_int y) { return std::abs(key[x]) > std::abs(key[y]); });
}

/* _ilupp.solve as binding.cpp:200-230 runs it: solve_with_multilevel_preconditioner (solving_routines_implementation.h:193-217 -> BiCGstab,
 * SPLIT, MIN_ITER 1, from the zero vector).  Returns 1 on success, 0 for "did not converge"; it / rel / res: what the binding returns. */
int ref_solve(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr, const orc_ml_params *IP, const double *rhs,
              double rtol, double atol, orc_int max_iter, double *x_out, orc_int *it, double *rel, double *res)
{
    matrix A = view(n, ptr, idx, val, is_csr);
    iluplusplus_precond_parameter param;
    if (param_of(IP, param)) return -1;
    vector b(n, const_cast<double *>(rhs), true), x(n, x_out, true), x_exact;
    Real rel_tol = -std::log10(rtol), abs_tol = -std::log10(atol), abs_error = 0;
    Integer iters = max_iter;
    const bool ok = solve_with_multilevel_preconditioner(A, b, x_exact, x, false, rel_tol, abs_tol, iters, abs_error, "", "", param);
    *it = iters; *rel = std::pow(10.0, -rel_tol); *res = std::pow(10.0, -abs_tol);
    return ok ? 1 : 0;
}

}  // extern "C"
