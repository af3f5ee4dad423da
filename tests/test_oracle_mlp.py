"""CPU tests of the oracle's restatement of the multilevel factorisation WITH pivoting (oracle/ilupp_oracle.c partial_ilucdp; reference
partialILUCDP, ILUCDP.hpp:268-1404, selected by the reference's default-constructed parameters):

* against tests/golden/mlp.npz (make_golden_mlp.py: the REAL reference on tests/ml_cases.py PIVOT_PARAMS, CSR and CSC): levels, sizes,
  total_nnz, a digest of every level's arrays, apply and apply_trans in full;
* against tests/golden/ml.npz (make_golden_ml.py): the reference's presets -- default-constructed, default_configuration(0 / 1 / 10 / 11) -- on
  its own test matrices, through the package's parameter object (ilupp_amd/params.py) and its C-ABI block;
* live against oracle/_ref on random cases with random parameters, where the reference build is present."""
import hashlib
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import ml_cases as C  # noqa: E402
from oracle import oracle as O  # noqa: E402


def _digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "mlp.npz"))


def test_fixture_is_complete(gold):
    names = [n for n, _ in C.matrices()]
    infos = [k for k in gold.files if k.endswith("/info")]
    assert len(infos) == len(names) * 2 * len(C.PIVOT_PARAMS)
    assert sum(1 for k in infos if gold[k][0] > 1) >= 60, "the fixture must hold cases with several levels"
    assert max(int(gold[k][0]) for k in infos) == 100                         # MAX_LEVELS reached (levels of a few rows each)


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("name", [n for n, _ in C.matrices()])
def test_oracle_against_reference_vectors(gold, name, fmt):
    key = "%s_%s" % (name, fmt)
    a = (gold[key + "/data"], gold[key + "/indices"], gold[key + "/indptr"], fmt == "csr")
    b = C.rhs(a[2].shape[0] - 1)
    for tag, thr, pre, knobs in C.PIVOT_PARAMS:
        k2 = "%s/%s" % (key, tag)
        P = O.orc().ml(a, C.oracle_params(O, thr, pre, knobs))
        info = gold[k2 + "/info"]
        assert P.levels() == info[0] and P.total_nnz() == info[1], k2
        assert [P.level(k)["n"] for k in range(P.levels())] == list(info[2:]), k2
        for k in range(P.levels()):
            assert np.array_equal(_digest(C.level_arrays(P.level(k))), gold[k2 + "/levels_sha"][k]), (k2, k)
        assert np.array_equal(P.apply(b), gold[k2 + "/apply"], equal_nan=True), k2
        assert np.array_equal(P.apply(b, O.TRANSPOSE), gold[k2 + "/apply_trans"], equal_nan=True), k2


def test_oracle_against_the_presets_of_the_reference():
    """ml.npz: ILUppPreconditioner(A, threshold, fill_in) with default-constructed parameters and default_configuration(0, 1, 10, 11), on the
    reference's test matrices (test/tests.py:9-36): the package's parameter object -> its C-ABI block -> the oracle"""
    import ilupp_amd as ilupp
    z = np.load(os.path.join(ROOT, "tests", "golden", "ml.npz"))
    seen = refused = 0
    for key, tag, config, thr, fill in C.ml_npz_cases(z):
        a = (z[key + "/A_data"], z[key + "/A_indices"], z[key + "/A_indptr"], bool(z[key + "/A_is_csr"]))
        n = a[2].shape[0] - 1
        blk = C.ml_npz_params(ilupp, config, thr, fill)._to_ml_params()
        name = "%s/%s" % (key, tag)
        try:
            P = O.orc().ml(a, C.block_to_oracle(O, blk))
        except O.OracleError as e:
            assert e.code == O.ERR_UNSUPPORTED and config == 11, name          # the move-to-corner ordering rejects an index: undefined in the reference
            refused += 1
            continue
        assert name + "_info" in z.files, name
        assert (P.levels(), P.total_nnz()) == tuple(int(v) for v in z[name + "_info"]), name
        b = C.rhs(n)
        assert np.array_equal(P.apply(b), z[name + "_apply"], equal_nan=True), name
        assert np.array_equal(P.apply(b, O.TRANSPOSE), z[name + "_apply_trans"], equal_nan=True), name
        seen += 1
    assert seen + refused == 120 and seen >= 100


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_against_reference_live():
    rng = np.random.default_rng(2024 + int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")))
    for it in range(40):
        n = int(rng.integers(5, 250))
        A = (sp.random(n, n, min(1.0, rng.uniform(2, 8) / n), random_state=rng, data_rvs=lambda k: rng.standard_normal(k))
             + sp.eye(n) * float(rng.choice([0.0, 0.5, 3.0]))).tocsr()
        A.sort_indices()
        a = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True)
        kw = dict(O.PIVOTING_DEFAULTS)
        kw.update(piv_tol=float(rng.choice([1.0, 0.5, 0.1, 0.0])), permute_rows=int(rng.integers(0, 4)), total_piv=int(rng.integers(0, 3)),
                  begin_total_piv=int(rng.integers(0, 2)), final_row_crit=int(rng.integers(-1, 10)), min_elim_factor=float(rng.choice([0.0, 0.3, 0.5])),
                  small_pivot_terminates=int(rng.integers(0, 2)), move_level_factor=float(rng.choice([0.5, 2.0])))
        if rng.integers(0, 3) == 0:
            kw["max_fill_in"] = int(rng.integers(1, 8))
        if rng.integers(0, 3) == 0:
            kw["drop_rules"] = int(rng.integers(1, 32))
        pre = [(O.PRE_PQ_ORDERING,), (O.PRE_MAX_WEIGHTED_MATCHING_ORDERING,), (1, 2, 3), (7,), ()][int(rng.integers(0, 5))]
        prm = O.ml_params(float(rng.choice([0.0, 1e-3, 1e-2, 0.1, 1.0])), pre, **kw)
        P, R = O.orc().ml(a, prm), O.ref().ml(a, prm)
        assert P.levels() == R.levels() and P.total_nnz() == R.total_nnz(), it
        for k in range(P.levels()):
            for x, y in zip(C.level_arrays(P.level(k)), C.level_arrays(R.level(k))):
                assert np.array_equal(x, y, equal_nan=(np.asarray(x).dtype.kind == "f")), (it, k)
        b = C.rhs(n)
        assert np.array_equal(P.apply(b), R.apply(b), equal_nan=True) and np.array_equal(P.apply(b, O.TRANSPOSE), R.apply(b, O.TRANSPOSE), equal_nan=True), it
