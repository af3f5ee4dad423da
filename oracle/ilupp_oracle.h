/*
 * oracle/ilupp_oracle.h -- CPU restatement of the ilupp hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C, single-threaded restatement of the reference's
 * algorithms, operation for operation.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (ilupp_amd) never does.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here bit-for-bit against
 * golden vectors produced by the reference itself (oracle/_ref, built from /root/reference by
 * oracle/Makefile; generator tests/golden/make_golden.py).
 *
 * All matrices are compressed sparse (CSR or CSC), int32 indices, fp64 values
 * (reference: declarations.h:49-58).
 */
#ifndef ILUPP_ORACLE_H
#define ILUPP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t orc_int;

/* A compressed sparse matrix; arrays are malloc()ed by the oracle and released with orc_free_mat. */
typedef struct {
    orc_int n;        /* square dimension */
    orc_int nnz;      /* = ptr[n] */
    orc_int *ptr;     /* n+1 */
    orc_int *idx;     /* nnz */
    double  *val;     /* nnz */
    int      is_csr;  /* 1 = ROW orientation, 0 = COLUMN */
} orc_mat;

enum { ORC_LOWER = 0, ORC_UPPER = 1 };
enum { ORC_ID = 0, ORC_TRANSPOSE = 1 };

/* error codes */
enum {
    ORC_OK = 0,
    ORC_ERR_ZERO_PIVOT = 1,      /* ILUT.hpp:269-270; *err_row receives the row */
    ORC_ERR_NOT_TRIANGULAR = 2,  /* IChol.hpp:54-55, :105-107 */
    ORC_ERR_MEMORY = 3,          /* sparse_implementation.h:3178-3179 ("insufficient memory reserved") */
    ORC_ERR_UNSUPPORTED = 4      /* a knob of the multilevel preconditioner that is not restated */
};

void orc_free_mat(orc_mat *M);

/* ILU0.hpp:69-106 (compute_ilu0 :26-66, sparse_vec_update :8-23). */
int orc_ilu0(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_mat *L, orc_mat *U);

/* ILUT.hpp:199-278 (ILUT_heap) + orientation handling of binding.cpp:432-447 /
 * preconditioner_implementation.h:992-1011; dropping.hpp:8-34. */
int orc_ilut(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_int max_fill_in, double threshold, orc_mat *L, orc_mat *U, orc_int *err_row);

/* IChol.hpp:63-73 (compute_ichol0 :33-59). */
int orc_ichol0(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
               orc_mat *L);

/* IChol.hpp:158-164 -> ICholT_tri :78-155. */
int orc_icholt(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
               orc_int add_fill_in, double threshold, orc_mat *L);

/* ILUC.hpp:112-207 (ILUC2) with the helpers :31-101; results as binding.cpp:449-460 returns them: the first factor is
 * stored column-wise (is_csr = 0: for ROW input L with its unit diagonal first in every column, for COLUMN input U), the second
 * row-wise. */
int orc_iluc(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_int max_fill_in, double threshold, orc_mat *L, orc_mat *U, orc_int *err_row);

/* sparse_implementation.h:4040-4087: in-place triangular solve, loop chosen by (form, orientation, use). */
void orc_trisolve(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
                  int form, int use, double *x);

/* preconditioner_implementation.h:103-111 + :321-334: LU apply (ID: L then U; TRANSPOSE: U^T then L^T). */
void orc_apply_lu(const orc_mat *L, const orc_mat *U, int use, double *x);

/* preconditioner_implementation.h:103-111 + :381-394: LL^T apply. */
void orc_apply_llt(const orc_mat *L, int use, double *x);

/* libstdc++ std::sort restated (bits/stl_algo.h:1855-1957), exposed for its own unit test:
 * sorts slot ids `list[0..len)` by DEcreasing |key[slot]|. */
void orc_sort_slots_by_abs_desc(orc_int *list, orc_int len, const double *key);

/* dropping.hpp:8-34 on raw working-row arrays; returns the number of kept slots written to `list`. */
orc_int orc_threshold_and_drop(const double *wdata, const orc_int *wpointer, orc_int wnnz,
                               orc_int *list, orc_int n, double tau, orc_int from, orc_int to);

/* ---- multilevel ILU++ without pivoting (the precon_parameter 10 family): preconditioner_implementation.h:1350-1665 over
 * ILUCDP.hpp:1405-2231, preprocessing sparse_implementation.h:5214-5460, apply preconditioner_implementation.h:433-488 ---- */
enum {                                   /* steps of a preprocessing_sequence (orderings.h:27-63) that are restated */
    ORC_PRE_NORMALIZE_COLUMNS = 1,
    ORC_PRE_NORMALIZE_ROWS = 2,
    ORC_PRE_PQ_ORDERING = 3,
    ORC_PRE_MAX_WEIGHTED_MATCHING_ORDERING = 4,
    ORC_PRE_DD_SYMM_MOVE_CORNER_ORDERING_IM = 5,
    ORC_PRE_UNIT_OR_ZERO_DIAGONAL_SCALING = 6,
    ORC_PRE_SPARSE_FIRST_ORDERING = 7,
    ORC_PRE_SYMM_PQ = 8
};

enum { ORC_DROP_STANDARD = 1, ORC_DROP_STANDARD2 = 2, ORC_DROP_ERR_PROP = 4, ORC_DROP_ERR_PROP2 = 8, ORC_DROP_PIVOT = 16, ORC_DROP_INVERSE = 32,
       ORC_DROP_WEIGHTED = 64, ORC_DROP_WEIGHTED2 = 128 };   /* USE_WEIGHTED_DROPPING (the accumulated weights enter the dropping weight), _DROPPING2 (they are only accumulated) */

typedef struct {
    double threshold;                    /* tau of the first level */
    int n_preprocessing;
    int preprocessing[8];
    double pq_threshold;                 /* PQ_THRESHOLD */
    int max_levels;                      /* MAX_LEVELS */
    orc_int min_ml_size;                 /* MIN_ML_SIZE */
    int small_pivot_terminates;          /* SMALL_PIVOT_TERMINATES */
    double min_pivot;                    /* MIN_PIVOT */
    double min_elim_factor;              /* MIN_ELIM_FACTOR */
    double threshold_shift_schur;        /* THRESHOLD_SHIFT_SCHUR */
    double vary_threshold_factor;        /* VARY_THRESHOLD_FACTOR */
    int use_final_threshold;             /* USE_FINAL_THRESHOLD */
    double final_threshold;              /* FINAL_THRESHOLD */
    orc_int max_fill_in;                 /* 0: MAX_FILLIN_IS_INF; else fill_in (entries a row of U / a column of L may have, the 1 included) */
    int drop_rules;                      /* ORC_DROP_*: USE_STANDARD_DROPPING, _DROPPING2, USE_ERR_PROP_DROPPING, _DROPPING2, USE_PIVOT_DROPPING, USE_INVERSE_DROPPING */
    double weight_standard_drop, weight_standard_drop2, weight_err_prop_drop, weight_err_prop_drop2, weight_pivot_drop;   /* WEIGHT_* */
    int combine_factor;                  /* COMBINE_FACTOR */
    double neutral_element, min_weight;  /* NEUTRAL_ELEMENT, MIN_WEIGHT */
    int scale_weight_invdiag;            /* SCALE_WEIGHT_INVDIAG */
    /* the factorisation WITH pivoting (partialILUCDP, ILUCDP.hpp:268-1404) is taken unless PERMUTE_ROWS is 0 (or 1), TOTAL_PIV is off
     * and piv_tol is 0 (preconditioner_implementation.h:1376-1382) */
    double piv_tol;                      /* piv_tol */
    int permute_rows;                    /* PERMUTE_ROWS 0..3 */
    int total_piv;                       /* TOTAL_PIV 0..2 */
    int begin_total_piv;                 /* BEGIN_TOTAL_PIV */
    int final_row_crit;                  /* FINAL_ROW_CRIT -1..9 (the ones that count the entries of a row of L) */
    double move_level_factor;            /* MOVE_LEVEL_FACTOR */
    double row_u_max;                    /* ROW_U_MAX */
    double weight_inverse_drop;          /* WEIGHT_INVERSE_DROP (with ORC_DROP_INVERSE: USE_INVERSE_DROPPING, ILUCDP.hpp:680-713, :882-916) */
    double weight_weighted_drop;         /* WEIGHT_WEIGHTED_DROP (with ORC_DROP_WEIGHTED, :726, :921) */
    double init_weights_lu;              /* INIT_WEIGHTS_LU: what weightsL / weightsU start from (:426-428) */
} orc_ml_params;

typedef struct orc_ml orc_ml;

typedef struct {                         /* one level, borrowed from the object */
    orc_int n;
    orc_mat L, U;
    const double *D;
    const orc_int *perm_rows, *perm_cols, *inv_perm_rows, *inv_perm_cols;
    const double *D_l, *D_r;
    orc_int zero_pivots;
} orc_ml_level_view;

void orc_ml_default_params(orc_ml_params *p);      /* default_configuration(1): NORMALIZE_COLUMNS, NORMALIZE_ROWS, PQ + precon_parameter 10 */
int orc_ml_create(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr, const orc_ml_params *IP, orc_ml **out);
int orc_ml_levels(const orc_ml *P);
orc_int orc_ml_total_nnz(const orc_ml *P);
int orc_ml_level(const orc_ml *P, int k, orc_ml_level_view *v);
void orc_ml_apply(const orc_ml *P, int use, double *x);
void orc_ml_free(orc_ml *P);


/* ---- ILUCP (ILUC.hpp:212-370, preconditioner_implementation.h:1117-1147, sparse_implementation.h:4166-4253): Crout ILU with
 * column pivoting on the major-order view of the arrays; the apply of ILUCPPreconditioner for the input's orientation ---- */
int orc_ilucp(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, orc_int max_fill_in, double threshold, double piv_tol,
              orc_int rp, double mem_factor, orc_mat *L, orc_mat *U, orc_int *perm, orc_int *zero_pivots);
void orc_apply_ilucp(const orc_mat *L, const orc_mat *U, const orc_int *perm, int input_is_csr, int use, double *x);

/* ---- ILUTP (ILUTP.hpp:13-140, preconditioner_implementation.h:1050-1078): ILUT with column pivoting on the major-order view of
 * the arrays (rows); the apply of ILUTPPreconditioner for the input's orientation ---- */
int orc_ilutp(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, orc_int max_fill_in, double threshold, double piv_tol,
              orc_int bp, double mem_factor, orc_mat *L, orc_mat *U, orc_int *perm, orc_int *zero_pivots);
void orc_apply_ilutp(const orc_mat *L, const orc_mat *U, const orc_int *perm, int input_is_csr, int use, double *x);

#ifdef __cplusplus
}
#endif
#endif
