"""The cases of the multilevel ILU++ tests (preset 10 family): shared by the generator of the golden
vectors (tests/golden/make_golden_ml10.py), the CPU test of the oracle and the GPU parity test.

A case = (name, matrix builder, parameter changes on top of default_configuration(1)); the matrices are the reference's own test
matrices (test/tests.py:9-36) and config-shaped small ones, each as CSR and CSC."""
import numpy as np
import scipy.sparse as sp

import matgen

PRE = {"NORMALIZE_COLUMNS": 1, "NORMALIZE_ROWS": 2, "PQ_ORDERING": 3, "MAX_WEIGHTED_MATCHING_ORDERING": 4, "DD_SYMM_MOVE_CORNER_ORDERING_IM": 5,
       "UNIT_OR_ZERO_DIAGONAL_SCALING": 6, "SPARSE_FIRST_ORDERING": 7, "SYMM_PQ": 8}


def laplace_matrix(n):
    return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr") * (n + 1) ** 2


def laplace2d_matrix(n_total):
    n = int(np.sqrt(n_total))
    A, I = laplace_matrix(n), sp.eye(n)
    return (sp.kron(A, I) + sp.kron(I, A)).tocsr()


def weak_random(n, density, diag, seed):
    """a random matrix with a weak diagonal: small pivots end levels, the Schur complements fill (stored in the fixture as arrays:
    scipy's sampler is version dependent)"""
    return (sp.random(n, n, density=density, random_state=np.random.default_rng(seed), format="csr") + sp.eye(n) * diag).tocsr()


def offdiag_random(n, seed):
    """large entries off the diagonal, none on it: the matching has to permute"""
    R = weak_random(n, 0.02, 0.0, seed)
    return (R + sp.diags([np.full(n - 1, 3.0)], [1], shape=(n, n)) + sp.diags([np.full(1, 2.5)], [-(n - 1)], shape=(n, n))).tocsr()


def matrices():
    yield "offdiag_150", offdiag_random(150, 3)
    d, i, p = matgen.poisson3d(6, 7, 5)
    yield "p3d_6_7_5", sp.csr_matrix((d, i, p), shape=(210, 210))
    yield "laplace2d_400", laplace2d_matrix(400)
    yield "rdd_300", sp.csr_matrix(matgen.random_dd(300, k=7, diag=3.0), shape=(300, 300))
    yield "weak_300", weak_random(300, 0.03, 0.5, 5)
    yield "weak_120", weak_random(120, 0.05, 0.2, 11)
    yield "weak_200", weak_random(200, 0.04, 0.05, 23)


# parameter sets: (tag, threshold, preprocessing, other knobs by the reference's names)
PARAMS = [
    ("t0", 0.0, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {}),
    ("t0.01", 0.01, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {}),
    ("t0.1", 0.1, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {}),
    ("t1", 1.0, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {}),
    ("t0.1_nopre", 0.1, (), {}),
    ("t0.1_norm", 0.1, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS"), {}),
    ("t0.1_rows_first", 0.1, ("NORMALIZE_ROWS", "PQ_ORDERING", "NORMALIZE_COLUMNS"), {}),
    ("t0.05_maxlev3", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"MAX_LEVELS": 3}),
    ("t0.05_shift", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"THRESHOLD_SHIFT_SCHUR": 1e-3, "MIN_PIVOT": 0.1}),
    ("t0.05_pq0.3", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"PQ_THRESHOLD": 0.3, "MIN_ELIM_FACTOR": 0.25}),
    ("t0.02_vary", 0.02, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"VARY_THRESHOLD_FACTOR": 2.0, "USE_FINAL_THRESHOLD": True, "FINAL_THRESHOLD": 0.5}),
    ("t0.05_noterm", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"SMALL_PIVOT_TERMINATES": False}),
    # maximum-weight matching (the I-matrix preprocessing of default_configuration(10 .. 13)) and what is combined with it
    ("t0.05_mwm", 0.05, ("MAX_WEIGHTED_MATCHING_ORDERING",), {}),
    ("t0_mwm", 0.0, ("MAX_WEIGHTED_MATCHING_ORDERING",), {}),
    ("t0.05_norm_mwm", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "MAX_WEIGHTED_MATCHING_ORDERING"), {}),
    ("t0.05_sf_mwm", 0.05, ("SPARSE_FIRST_ORDERING", "MAX_WEIGHTED_MATCHING_ORDERING"), {}),
    ("t0.05_mwm_unit", 0.05, ("MAX_WEIGHTED_MATCHING_ORDERING", "UNIT_OR_ZERO_DIAGONAL_SCALING"), {}),
    ("t0.05_mwm_pq", 0.05, ("MAX_WEIGHTED_MATCHING_ORDERING", "PQ_ORDERING"), {}),
    # the symmetric PQ step (rows and columns by increasing row weight, orderings_implementation.h:573-584), alone and after the matching
    ("t0.05_mwm_spq", 0.05, ("MAX_WEIGHTED_MATCHING_ORDERING", "SYMM_PQ"), {}),
    ("t0.02_norm_mwm_spq", 0.02, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "MAX_WEIGHTED_MATCHING_ORDERING", "SYMM_PQ"), {}),
    ("t0.1_spq", 0.1, ("SYMM_PQ",), {}),
    # bounded fill (presets 1000+: MAX_FILLIN_IS_INF false): at most fill_in entries per row of U / column of L, the largest by the
    # reference's own selection algorithm; THRESHOLD_SHIFT_SCHUR as in preset 1010
    ("t0_fill4", 0.0, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"fill_in": 4}),
    ("t0.02_fill7_shift", 0.02, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"fill_in": 7, "THRESHOLD_SHIFT_SCHUR": 1e-3}),
    ("t0.01_fill1", 0.01, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"fill_in": 1}),
    ("t0_mwm_fill12", 0.0, ("MAX_WEIGHTED_MATCHING_ORDERING",), {"fill_in": 12, "THRESHOLD_SHIFT_SCHUR": 1e-3}),
    # the other dropping rules that are local to a step: dual threshold (preset 13), pivot, the second error-propagation rule; combined by
    # maximum, sum, product (COMBINE_FACTOR), weights and neutral element changed
    ("t0.05_std", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"USE_STANDARD_DROPPING": True, "USE_ERR_PROP_DROPPING": False}),
    ("t0.05_std_err_sum", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"USE_STANDARD_DROPPING": True, "COMBINE_FACTOR": 1, "WEIGHT_STANDARD_DROP": 0.3}),
    ("t0.05_piv_err2_prod", 0.05, ("MAX_WEIGHTED_MATCHING_ORDERING",), {"USE_PIVOT_DROPPING": True, "USE_ERR_PROP_DROPPING2": True, "USE_ERR_PROP_DROPPING": False,
                                                                          "COMBINE_FACTOR": 2, "NEUTRAL_ELEMENT": 1.0, "WEIGHT_PIVOT_DROP": 0.5}),
    ("t0.1_std2_minw_invdiag", 0.1, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"USE_STANDARD_DROPPING2": True, "COMBINE_FACTOR": 3, "MIN_WEIGHT": 0.5,
                                                                                              "SCALE_WEIGHT_INVDIAG": True}),
    # the rules whose estimates are recurrences over all steps (presets 11, 12: inverse-based, weighted dropping; ILUCDP.hpp:1676-1710,
    # :1856-1892, :1634-1636, :1670-1675): the factorisation runs as a chain
    ("t0.05_inv", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"USE_INVERSE_DROPPING": True, "USE_ERR_PROP_DROPPING": False}),
    ("t0.02_inv_err_mwm", 0.02, ("MAX_WEIGHTED_MATCHING_ORDERING",), {"USE_INVERSE_DROPPING": True, "COMBINE_FACTOR": 1, "WEIGHT_INVERSE_DROP": 0.5}),
    ("t0.05_wgt", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {"USE_WEIGHTED_DROPPING": True, "USE_ERR_PROP_DROPPING": False, "INIT_WEIGHTS_LU": 0.5}),
    # default_configuration(11): where the move-to-corner ordering rejects an index the reference's result is undefined:
    # the oracle and the engine refuse, and the fixture records that
    ("t0.05_mwm_dd", 0.05, ("MAX_WEIGHTED_MATCHING_ORDERING", "DD_SYMM_MOVE_CORNER_ORDERING_IM"), {}),
]

_ORC_FIELDS = {"MAX_LEVELS": "max_levels", "THRESHOLD_SHIFT_SCHUR": "threshold_shift_schur", "MIN_PIVOT": "min_pivot", "PQ_THRESHOLD": "pq_threshold",
               "MIN_ELIM_FACTOR": "min_elim_factor", "VARY_THRESHOLD_FACTOR": "vary_threshold_factor", "USE_FINAL_THRESHOLD": "use_final_threshold",
               "FINAL_THRESHOLD": "final_threshold", "SMALL_PIVOT_TERMINATES": "small_pivot_terminates", "MIN_ML_SIZE": "min_ml_size",
               "fill_in": "max_fill_in", "WEIGHT_STANDARD_DROP": "weight_standard_drop", "WEIGHT_STANDARD_DROP2": "weight_standard_drop2",
               "WEIGHT_ERR_PROP_DROP": "weight_err_prop_drop", "WEIGHT_ERR_PROP_DROP2": "weight_err_prop_drop2", "WEIGHT_PIVOT_DROP": "weight_pivot_drop",
               "COMBINE_FACTOR": "combine_factor", "NEUTRAL_ELEMENT": "neutral_element", "MIN_WEIGHT": "min_weight", "SCALE_WEIGHT_INVDIAG": "scale_weight_invdiag",
               "piv_tol": "piv_tol", "PERMUTE_ROWS": "permute_rows", "TOTAL_PIV": "total_piv", "BEGIN_TOTAL_PIV": "begin_total_piv",
               "FINAL_ROW_CRIT": "final_row_crit", "MOVE_LEVEL_FACTOR": "move_level_factor", "ROW_U_MAX": "row_u_max",
               "WEIGHT_INVERSE_DROP": "weight_inverse_drop", "WEIGHT_WEIGHTED_DROP": "weight_weighted_drop", "INIT_WEIGHTS_LU": "init_weights_lu"}

# the reference's DEFAULT-constructed parameters (precon_parameter 0, parameters_implementation.h:430-501): the factorisation with pivoting
PIVOTING = {"piv_tol": 1.0, "PERMUTE_ROWS": 3, "TOTAL_PIV": 1, "BEGIN_TOTAL_PIV": True, "SMALL_PIVOT_TERMINATES": False, "MIN_ELIM_FACTOR": 0.5}


def pivoting(**kw):
    d = dict(PIVOTING)
    d.update(kw)
    return d


# (name, threshold, preprocessing, knobs) for the factorisation WITH pivoting (partialILUCDP)
PIVOT_PARAMS = [
    ("p_t1_pq", 1.0, ("PQ_ORDERING",), pivoting()),                                  # ILUppPreconditioner(A): threshold 1.0, set_PQ
    ("p_t0_pq", 0.0, ("PQ_ORDERING",), pivoting()),                                  # ILUppPreconditioner(A, threshold=0) (test/tests.py:390, :398)
    ("p_t0.01_pq", 1e-2, ("PQ_ORDERING",), pivoting()),                              # tests.py:346-349
    ("p_t0.01_mwm", 1e-2, ("MAX_WEIGHTED_MATCHING_ORDERING",), pivoting()),          # default_configuration(10); tests.py:355-358
    ("p_t0.01_sf", 1e-2, ("SPARSE_FIRST_ORDERING",), pivoting()),                    # tests.py:373-376
    ("p_t0.01_mwm_spq", 1e-2, ("MAX_WEIGHTED_MATCHING_ORDERING", "SYMM_PQ"), pivoting()),   # tests.py:362-369
    ("p_t0.1_norm_pq", 0.1, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), pivoting()),
    ("p_t0.01_tol0.1", 1e-2, ("PQ_ORDERING",), pivoting(piv_tol=0.1)),
    ("p_t0.01_rows0_piv2", 1e-2, ("PQ_ORDERING",), pivoting(PERMUTE_ROWS=0, TOTAL_PIV=2)),
    ("p_t0.01_rows1_crit0", 1e-2, ("PQ_ORDERING",), pivoting(PERMUTE_ROWS=1, FINAL_ROW_CRIT=0, MIN_ELIM_FACTOR=0.2)),
    ("p_t0.01_rows2_crit7", 1e-2, ("MAX_WEIGHTED_MATCHING_ORDERING",), pivoting(PERMUTE_ROWS=2, FINAL_ROW_CRIT=7, MIN_ELIM_FACTOR=0.0)),
    ("p_t0.001_fill5_std", 1e-3, ("PQ_ORDERING",), pivoting(fill_in=5, USE_STANDARD_DROPPING=True, THRESHOLD_SHIFT_SCHUR=0.5)),
    ("p_t0.05_smallpiv", 0.05, ("PQ_ORDERING",), pivoting(SMALL_PIVOT_TERMINATES=True, MIN_ELIM_FACTOR=0.1, MOVE_LEVEL_FACTOR=0.5)),
    # inverse-based dropping (precon_parameter 1 = the default-constructed parameters with USE_INVERSE_DROPPING instead of the error-propagation
    # rule, parameters_implementation.h:872-876), alone, combined with another rule, with its weight changed
    ("p_t0.1_inv", 0.1, ("PQ_ORDERING",), pivoting(USE_INVERSE_DROPPING=True, USE_ERR_PROP_DROPPING=False)),
    ("p_t0.02_inv_mwm", 0.02, ("MAX_WEIGHTED_MATCHING_ORDERING",), pivoting(USE_INVERSE_DROPPING=True, USE_ERR_PROP_DROPPING=False)),
    ("p_t0.05_inv_err_sum", 0.05, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), pivoting(USE_INVERSE_DROPPING=True, COMBINE_FACTOR=1, WEIGHT_INVERSE_DROP=0.25)),
    # weighted dropping (precon_parameter 2, :877-881): the sums of |entries| a column of U / a row of L has collected so far weigh its step
    ("p_t0.1_wgt", 0.1, ("PQ_ORDERING",), pivoting(USE_WEIGHTED_DROPPING=True, USE_ERR_PROP_DROPPING=False)),
    ("p_t0.05_wgt_mwm_init", 0.05, ("MAX_WEIGHTED_MATCHING_ORDERING",), pivoting(USE_WEIGHTED_DROPPING=True, WEIGHT_WEIGHTED_DROP=0.1, INIT_WEIGHTS_LU=0.5)),
]


_RULES = (("USE_STANDARD_DROPPING", 1, False), ("USE_STANDARD_DROPPING2", 2, False), ("USE_ERR_PROP_DROPPING", 4, True), ("USE_ERR_PROP_DROPPING2", 8, False),
          ("USE_PIVOT_DROPPING", 16, False), ("USE_INVERSE_DROPPING", 32, False), ("USE_WEIGHTED_DROPPING", 64, False), ("USE_WEIGHTED_DROPPING2", 128, False))


def oracle_params(O, thr, pre, knobs):
    """the parameter block of oracle/ilupp_oracle.h for a case"""
    kw = {_ORC_FIELDS[k]: (int(v) if isinstance(v, bool) else v) for k, v in knobs.items() if k in _ORC_FIELDS}
    kw["drop_rules"] = sum(bit for name, bit, default in _RULES if knobs.get(name, default))
    return O.ml_params(thr, preprocessing=tuple(PRE[s] for s in pre), **kw)


def engine_params(ilupp, thr, pre, knobs):
    """the reference-style parameter object of ilupp_amd for a case"""
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(1)
    p.threshold = thr
    p.PREPROCESSING = ilupp.preprocessing_sequence(pre)
    for k, v in knobs.items():
        setattr(p, k, v)
    if "fill_in" in knobs:
        p.MAX_FILLIN_IS_INF = False
    return p


def rhs(n):
    return 1.0 + (np.arange(n, dtype=np.float64) % 17) / 16.0 - (np.arange(n, dtype=np.float64) % 5) / 8.0


def level_arrays(lv):
    """a level (dict from oracle.ML.level or ilupp_amd._native.MultilevelPreconditioner.level) as a flat list of arrays"""
    return [np.asarray(lv["L"][0]), np.asarray(lv["L"][1]), np.asarray(lv["L"][2]), np.asarray(lv["U"][0]), np.asarray(lv["U"][1]), np.asarray(lv["U"][2]),
            lv["D"], lv["perm_rows"], lv["perm_cols"], lv["inv_perm_rows"], lv["inv_perm_cols"], lv["D_l"], lv["D_r"]]


def block_to_oracle(O, blk):
    """the C-ABI parameter block of the package (ilupp_amd._native.MLParams) as the oracle's block: same fields, by name"""
    p = O.MLParams()
    for name, _ in O.MLParams._fields_:
        v = getattr(blk, name)
        if name == "preprocessing":
            for i in range(8):
                p.preprocessing[i] = v[i]
        else:
            setattr(p, name, v)
    return p


def ml_npz_params(ilupp, config, thr, fill):
    """the parameters of a case of tests/golden/ml.npz (oracle/ref_shim.cpp ref_ilupp_apply: default-constructed for config -1, else
    default_configuration(config); then set_threshold, and set_fill_in if fill >= 0)"""
    p = ilupp.iluplusplus_precond_parameter()
    if config >= 0:
        p.default_configuration(config)
    p.threshold = thr
    if fill >= 0:
        p.fill_in = fill
    return p


def ml_npz_cases(z):
    """(key of the matrix, tag, config, threshold, fill_in) of every case of ml.npz, failed ones included"""
    out = []
    for k in sorted(z.files):
        if not (k.endswith("_info") or k.endswith("_error")) or not k.startswith("ml_"):
            continue
        key, tag = k.rsplit("_", 1)[0].split("/")
        c, t, f = tag.split("_")
        out.append((key, tag, int(c[1:]), float(t[1:]), int(f[1:])))
    return out


# ---- `solve` (binding.cpp:200-230): matrices with right-hand sides, and (tag, threshold, preprocessing, knobs, rtol, atol, max_iter) ----
def solve_matrices():
    A = laplace2d_matrix(900)                                                       # test/tests.py:344-383
    yield "laplace2d_900", A, A @ np.ones(900)
    d, i, p = matgen.poisson3d(12, 11, 10)
    A = (sp.csr_matrix((d, i, p), shape=(1320, 1320)) + 0.5 * sp.diags([np.ones(1319)], [1], shape=(1320, 1320))).tocsr()
    yield "p3d_shift_1320", A, rhs(1320)
    A = sp.csr_matrix(matgen.random_dd(300, k=7, diag=3.0), shape=(300, 300))
    yield "rdd_300", A, A @ np.linspace(1.0, 2.0, 300)
    A = (sp.random(50, 50, density=0.1, random_state=39273) + 10.0 * sp.eye(50)).tocsr()       # tests.py:27-30 (stored as arrays in the fixture)
    yield "random_50", A, A @ np.ones(50)


_NPQ = ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING")
SOLVE_PARAMS = [
    ("t0.01_pq", 1e-2, _NPQ, {}, 1e-8, 1e-8, 500),
    ("t0.1_pq_loose", 0.1, _NPQ, {}, 1e-4, 1e-4, 500),
    ("t1_pq", 1.0, _NPQ, {}, 1e-8, 1e-8, 500),
    ("t0.01_mwm", 1e-2, ("MAX_WEIGHTED_MATCHING_ORDERING",), {}, 1e-8, 1e-8, 500),
    ("t0.01_mwm_spq", 1e-2, ("MAX_WEIGHTED_MATCHING_ORDERING", "SYMM_PQ"), {}, 1e-8, 1e-8, 500),
    ("t0.01_pq_2it", 1e-2, _NPQ, {}, 1e-14, 1e-14, 2),                               # stops at max_iter: "did not converge"
    ("p_t0.01_pq", 1e-2, ("PQ_ORDERING",), pivoting(), 1e-8, 1e-8, 500),             # the default-constructed parameters
    ("p_t0.01_mwm_spq", 1e-2, ("MAX_WEIGHTED_MATCHING_ORDERING", "SYMM_PQ"), pivoting(), 1e-8, 1e-8, 500),    # tests.py:362-369
]
