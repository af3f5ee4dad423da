"""CPU tests of SURVEY section 8(f) rank 3, the multilevel ILU++ preconditioner without pivoting (precon_parameter 10 family):

* the oracle's restatement (oracle/ilupp_oracle.c: orc_ml_*, of preconditioner_implementation.h:1350-1665 over ILUCDP.hpp:1405-2231)
  against the golden vectors the REAL reference emitted (tests/golden/ml10.npz from tests/golden/make_golden_ml10.py): number and
  sizes of the levels, total_nnz, every level's factors / diagonal / permutations / scalings (sha256), apply and apply_trans -- bit for bit;
* where oracle/_ref is present (the build container), the same comparison live on further matrices;
* the parameter objects of the package (ilupp_amd/params.py against parameters_implementation.h:430-609, :872-934) and the refusal
  of everything outside the built family -- no GPU involved.
"""
import hashlib
import os

import numpy as np
import pytest
import scipy.sparse as sp

import ml_cases as C
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "ml10.npz")


def _digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _keys(gold):
    return sorted({k.rsplit("/", 1)[0] for k in gold.files if k.endswith("/info")})


def test_fixture_is_complete(gold):
    names = [n for n, _ in C.matrices()]
    refused = [k for k in gold.files if k.endswith("/refused")]
    assert len(_keys(gold)) + len(refused) == len(names) * 2 * len(C.PARAMS)
    assert 0 < len(refused) < 2 * len(names)        # the move-to-corner ordering: defined on some of the matrices, undefined on others
    multi = sum(1 for k in _keys(gold) if gold[k + "/info"][0] > 1)
    assert multi >= 25, "the fixture must hold cases with several levels"


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("name", [n for n, _ in C.matrices()])
def test_oracle_against_reference_vectors(gold, name, fmt):
    key = "%s_%s" % (name, fmt)
    a = (gold[key + "/data"], gold[key + "/indices"], gold[key + "/indptr"], fmt == "csr")
    n = a[2].shape[0] - 1
    b = C.rhs(n)
    for tag, thr, pre, knobs in C.PARAMS:
        k2 = "%s/%s" % (key, tag)
        if k2 + "/refused" in gold.files:
            with pytest.raises(O.OracleError):
                O.orc().ml(a, C.oracle_params(O, thr, pre, knobs))
            continue
        P = O.orc().ml(a, C.oracle_params(O, thr, pre, knobs))
        info = gold[k2 + "/info"]
        assert P.levels() == info[0] and P.total_nnz() == info[1], k2
        assert [P.level(k)["n"] for k in range(P.levels())] == list(info[2:]), k2
        for k in range(P.levels()):
            assert np.array_equal(_digest(C.level_arrays(P.level(k))), gold[k2 + "/levels_sha"][k]), (k2, k)
        assert np.array_equal(P.apply(b), gold[k2 + "/apply"], equal_nan=True), k2
        assert np.array_equal(P.apply(b, O.TRANSPOSE), gold[k2 + "/apply_trans"], equal_nan=True), k2


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_against_reference_live():
    rng = np.random.default_rng(17 + int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")))
    for n, dens, dg in ((80, 0.08, 0.1), (500, 0.01, 0.4), (900, 0.006, 1.5)):
        A = (sp.random(n, n, density=dens, random_state=rng, format="csr") + sp.eye(n) * dg).tocsr()
        for fmt in ("csr", "csc"):
            a = O.from_scipy(A.asformat(fmt))
            for thr, rules in ((0.0, O.DROP_ERR_PROP), (0.03, O.DROP_ERR_PROP), (0.3, O.DROP_ERR_PROP), (0.05, O.DROP_INVERSE),
                               (0.02, O.DROP_INVERSE | O.DROP_STANDARD)):            # (inverse-based dropping of partialILUC, ILUCDP.hpp:1676-1710, :1856-1892)
                p = O.ml_params(thr, drop_rules=rules)
                R, P = O.ref().ml(a, p), O.orc().ml(a, p)
                assert R.levels() == P.levels() and R.total_nnz() == P.total_nnz()
                for k in range(R.levels()):
                    for x, y in zip(C.level_arrays(R.level(k)), C.level_arrays(P.level(k))):
                        assert np.array_equal(x, y, equal_nan=True)
                b = C.rhs(n)
                assert np.array_equal(R.apply(b), P.apply(b), equal_nan=True)
                assert np.array_equal(R.apply(b, O.TRANSPOSE), P.apply(b, O.TRANSPOSE), equal_nan=True)


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_against_reference_fuzz():
    """random small matrices of all kinds with random parameter sets (tests/fuzz_ml.py): every level and both applies, bit for bit"""
    import fuzz_ml
    for seed in range(int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")), int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")) + 150):
        A, (thr, pre, knobs) = fuzz_ml.case(seed)
        a = O.from_scipy(A)
        p = C.oracle_params(O, thr, pre, knobs)
        R, P = O.ref().ml(a, p), O.orc().ml(a, p)
        assert R.levels() == P.levels() and R.total_nnz() == P.total_nnz(), seed
        for k in range(R.levels()):
            for x, y in zip(C.level_arrays(R.level(k)), C.level_arrays(P.level(k))):
                assert x.shape == y.shape and np.array_equal(x, y, equal_nan=True), (seed, k)
        b = C.rhs(A.shape[0])
        assert np.array_equal(R.apply(b), P.apply(b), equal_nan=True), seed
        assert np.array_equal(R.apply(b, O.TRANSPOSE), P.apply(b, O.TRANSPOSE), equal_nan=True), seed


def test_oracle_refuses_unknown_preprocessing():
    A = sp.eye(5, format="csr")
    with pytest.raises(O.OracleError):
        O.orc().ml(O.from_scipy(A), O.ml_params(0.1, preprocessing=(9,)))      # no such step


# ---- the parameter objects (no GPU) -------------------------------------------------------------------------------------------
def test_default_parameters_and_presets():
    import ilupp_amd as ilupp
    p = ilupp.iluplusplus_precond_parameter()
    # default_parameters(), parameters_implementation.h:430-501
    assert (p.fill_in, p.threshold, p.piv_tol, p.PRECON_PARAMETER, p.MAX_LEVELS, p.PERMUTE_ROWS, p.TOTAL_PIV) == (10000, 0.0, 1.0, 0, 100, 3, 1)
    assert p.MAX_FILLIN_IS_INF and p.USE_ERR_PROP_DROPPING and not p.SMALL_PIVOT_TERMINATES and p.MIN_PIVOT == 1e-2 and p.MEM_FACTOR == 3.0
    assert p.PREPROCESSING.to_names() == ["NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"]
    p.default_configuration(1)                      # :546-549 + init case 10 (:927-934)
    assert (p.PRECON_PARAMETER, p.PERMUTE_ROWS, p.TOTAL_PIV, p.piv_tol, p.MIN_ELIM_FACTOR) == (10, 0, 0, 0.0, 0.0) and p.SMALL_PIVOT_TERMINATES
    assert p.PREPROCESSING.to_names() == ["NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"]
    p.default_configuration(11)
    assert p.PRECON_PARAMETER == 10 and p.PREPROCESSING.to_names() == ["MAX_WEIGHTED_MATCHING_ORDERING", "DD_SYMM_MOVE_CORNER_ORDERING_IM"]
    p.default_configuration(1001)                   # :580-583 + init case 1010 (:1592-1601)
    assert (p.PRECON_PARAMETER, p.THRESHOLD_SHIFT_SCHUR, p.MAX_FILLIN_IS_INF, p.fill_in) == (1010, 1e-3, False, 500)
    b = p._to_ml_params()                           # the bounded-fill family is built
    assert (b.max_fill_in, b.threshold_shift_schur) == (500, 1e-3)
    p.init(ilupp.preprocessing_sequence(["PQ_ORDERING"]), 13)       # dual threshold dropping (:951-959): built
    assert p.USE_STANDARD_DROPPING and not p.USE_ERR_PROP_DROPPING and p._to_ml_params().drop_rules == 1
    p.default_configuration(10)
    assert p.PRECON_PARAMETER == 0 and p.PERMUTE_ROWS == 3 and p.PREPROCESSING.to_names() == ["MAX_WEIGHTED_MATCHING_ORDERING"]
    with pytest.raises(NotImplementedError):
        p.default_configuration(-4)
    p.PREPROCESSING.set_normalize()
    assert p.PREPROCESSING.to_names() == ["NORMALIZE_COLUMNS", "NORMALIZE_ROWS"]
    p.PREPROCESSING.set_none()
    assert p.PREPROCESSING.to_names() == []


def test_parameter_block_of_the_built_family():
    import ilupp_amd as ilupp
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(1)
    p.threshold = 0.25
    p.MIN_PIVOT = 0.05
    p.MAX_LEVELS = 7
    b = p._to_ml_params()
    assert (b.threshold, b.n_preprocessing, list(b.preprocessing)[:3], b.max_levels, b.min_pivot, b.small_pivot_terminates) == (0.25, 3, [1, 2, 3], 7, 0.05, 1)
    assert (b.min_elim_factor, b.threshold_shift_schur, b.vary_threshold_factor, b.use_final_threshold) == (0.0, 0.0, 1.0, 0)


def test_everything_outside_the_built_family_is_refused():
    """no silent substitution: dropping rules that carry estimates over the steps, the improved Schur complement, positional dropping,
    rows ordered by weights, preprocessing steps without a kernel -- each raises NotImplementedError before anything runs"""
    import ilupp_amd as ilupp

    def refused(change, start=1):
        p = ilupp.iluplusplus_precond_parameter()
        if start is not None:
            p.default_configuration(start)
        change(p)
        with pytest.raises(NotImplementedError):
            p._to_ml_params()
    refused(lambda p: setattr(p, "SCALE_WGT_MAXINVDIAG", True))
    refused(lambda p: setattr(p, "SCHUR_COMPLEMENT", 1))
    refused(lambda p: setattr(p, "DROP_TYPE_L", 1))
    refused(lambda p: p.PREPROCESSING.append("SYMM_MOVE_CORNER_ORDERING_IM"))
    refused(lambda p: setattr(p, "EXTERNAL_FINAL_ROW", True))
    refused(lambda p: setattr(p, "PRECON_PARAMETER", -1))
    refused(lambda p: setattr(p, "FINAL_ROW_CRIT", -2), start=None)       # rows by accumulated weights (a sorted container): with pivoting only
    refused(lambda p: setattr(p, "SCHUR_COMPLEMENT", 1), start=None)
    with pytest.raises(ValueError):
        p = ilupp.iluplusplus_precond_parameter(); p.PERMUTE_ROWS = 4; p._to_ml_params()


def test_inverse_based_and_weighted_dropping():
    """USE_INVERSE_DROPPING (precon_parameter 1: parameters_implementation.h:872-876) accumulates estimates over the steps in their order:
    both factorisations then run as sequential chains (pilucdp.hip: k_pilucdp_lds, k_piluc_chain)"""
    import ilupp_amd as ilupp
    p = ilupp.iluplusplus_precond_parameter()                  # default-constructed: the pivoting family
    p.use_only_inverse_dropping()
    b = p._to_ml_params()
    assert b.drop_rules == 32 and b.weight_inverse_drop == 1.0 and not p._uses_partial_iluc()
    p.USE_ERR_PROP_DROPPING = True
    p.WEIGHT_INVERSE_DROP = 0.5
    b = p._to_ml_params()
    assert b.drop_rules == 36 and b.weight_inverse_drop == 0.5
    p.use_only_weighted_dropping1()                            # precon_parameter 2
    p.INIT_WEIGHTS_LU = 0.5
    b = p._to_ml_params()
    assert b.drop_rules == 64 and b.weight_weighted_drop == 1.0 and b.init_weights_lu == 0.5
    # ... and without pivoting (precon_parameter 11, 12): the factorisation then runs as a chain as well
    q = ilupp.iluplusplus_precond_parameter()
    q.init(ilupp.preprocessing_sequence(["PQ_ORDERING"]), 11)
    assert q._uses_partial_iluc() and q._to_ml_params().drop_rules == 32
    q.init(ilupp.preprocessing_sequence(["PQ_ORDERING"]), 12)
    assert q._uses_partial_iluc() and q._to_ml_params().drop_rules == 64


def test_parameter_block_of_the_pivoting_family():
    """the reference's default-constructed parameters (parameters_implementation.h:430-501) select partialILUCDP
    (preconditioner_implementation.h:1376-1382): the block carries its windows and tolerances"""
    import ilupp_amd as ilupp
    p = ilupp.iluplusplus_precond_parameter()
    assert not p._uses_partial_iluc()
    b = p._to_ml_params()
    assert (b.piv_tol, b.permute_rows, b.total_piv, b.begin_total_piv, b.final_row_crit, b.move_level_factor, b.row_u_max) == (1.0, 3, 1, 1, -1, 2.0, 1.5)
    assert (b.small_pivot_terminates, b.min_elim_factor, b.n_preprocessing, list(b.preprocessing)[:3], b.drop_rules) == (0, 0.5, 3, [1, 2, 3], 4)   # set_PQ: normalisation + PQ
    p.default_configuration(10)                     # BASELINE config 5: maximum weighted matching + the pivoting factorisation
    b = p._to_ml_params()
    assert (b.permute_rows, b.total_piv, b.piv_tol, list(b.preprocessing)[:b.n_preprocessing]) == (3, 1, 1.0, [4])
    p.default_configuration(1)
    assert p._uses_partial_iluc()
    p.piv_tol = 0.5
    assert not p._uses_partial_iluc() and p._to_ml_params().piv_tol == 0.5


def test_solve_fixture_is_complete():
    """tests/golden/solve.npz (make_golden_solve.py: the REAL reference's `_ilupp.solve`): every case of ml_cases.SOLVE_PARAMS on every
    matrix, CSR and CSC; converging cases and the one that stops at max_iter"""
    g = np.load(os.path.join(HERE, "golden", "solve.npz"))
    names = [n for n, _, _ in C.solve_matrices()]
    infos = [k for k in g.files if k.endswith("/info")]
    assert len(infos) == len(names) * 2 * len(C.SOLVE_PARAMS)
    ok = [bool(g[k][0]) for k in infos]
    assert any(ok) and not all(ok)
    for name, A, b in C.solve_matrices():
        if name == "random_50":
            continue                                            # (scipy's sampler is version dependent: the fixture's arrays are the case)
        M = A.tocsr(); M.sort_indices()
        assert np.array_equal(g[name + "_csr/data"], M.data) and np.array_equal(g[name + "_csr/b"], b)
