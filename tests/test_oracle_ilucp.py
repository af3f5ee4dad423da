"""CPU tests of the oracle's restatement of ILUCP (first step: the oracle; the GPU side is not built yet -- the class
ilupp_amd.ILUCPPreconditioner still refuses): oracle/ilupp_oracle.c orc_ilucp / orc_apply_ilucp against

* tests/golden/ilucp.npz (make_golden_ilucp.py: ILUCPPreconditioner of the REAL reference on its own test matrices and config-shaped ones,
  CSR and CSC, five parameter sets: both factors, the permutation, apply, apply_trans);
* oracle/_ref live on random matrices with random parameters, where the reference build is present (incl. the reservation error)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

import ml_cases as C  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [(100, 0.1, 0.1), (100, 0.0, 0.0), (3, 1e-3, 1.0), (8, 1e-2, 0.5), (1, 0.1, 0.1)]
NAMES = ["laplace2d", "random", "rdd_300", "weak_200", "offdiag_150"]


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("name", NAMES)
def test_oracle_against_reference_vectors(name, fmt):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ilucp.npz"))
    key = "%s_%s" % (name, fmt)
    a = (gold[key + "/data"], gold[key + "/indices"], gold[key + "/indptr"], fmt == "csr")
    b = C.rhs(a[2].shape[0] - 1)
    pivoted = 0
    for fill, thr, tol in CASES:
        tag = "%s/f%d_t%g_p%g" % (key, fill, thr, tol)
        P = O.ILUCP(O.orc(), a, fill_in=fill, threshold=thr, piv_tol=tol)
        for nm, arr in zip(("L_data", "L_indices", "L_indptr", "U_data", "U_indices", "U_indptr"), P.L + P.U):
            assert np.array_equal(arr, gold[tag + "/" + nm], equal_nan=(arr.dtype.kind == "f")), (tag, nm)
        assert np.array_equal(P.perm, gold[tag + "/perm"]), tag
        assert np.array_equal(P.apply(b), gold[tag + "/apply"], equal_nan=True), tag
        assert np.array_equal(P.apply(b, O.TRANSPOSE), gold[tag + "/apply_trans"], equal_nan=True), tag
        pivoted += int((P.perm != np.arange(len(P.perm))).sum())
    assert pivoted > 0 or name not in ("weak_200", "offdiag_150")            # (the diagonally dominant ones never pivot)


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_against_reference_live():
    rng = np.random.default_rng(99 + int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")))
    failures = 0
    for it in range(80):
        n = int(rng.integers(2, 250))
        A = (sp.random(n, n, min(1.0, rng.uniform(2, 9) / n), random_state=rng, data_rvs=lambda k: rng.standard_normal(k))
             + sp.eye(n) * float(rng.choice([0.0, 0.3, 3.0]))).asformat("csr" if it % 2 else "csc")
        A.sort_indices()
        a = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), bool(it % 2))
        kw = dict(fill_in=int(rng.choice([1, 2, 5, 100])), threshold=float(rng.choice([0.0, 1e-3, 0.1, 0.5])), piv_tol=float(rng.choice([0.0, 0.1, 1.0])),
                  rp=int(rng.choice([-1, 0, n // 2])), mem_factor=float(rng.choice([10.0, 10.0, 1.0])))
        try:
            R = O.ILUCP(O.ref(), a, **kw)
        except O.OracleError as e:
            with pytest.raises(O.OracleError) as e2:
                O.ILUCP(O.orc(), a, **kw)
            assert e2.value.code == e.code == O.ERR_MEMORY
            failures += 1
            continue
        P = O.ILUCP(O.orc(), a, **kw)
        assert np.array_equal(P.perm, R.perm), it
        for x, y in zip(P.L + P.U, R.L + R.U):
            assert np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")), it
        b = C.rhs(n)
        assert np.array_equal(P.apply(b), R.apply(b), equal_nan=True) and np.array_equal(P.apply(b, O.TRANSPOSE), R.apply(b, O.TRANSPOSE), equal_nan=True), it
    assert failures >= 1            # "ILUCP4: Insufficient memory reserved" (mem_factor 1) is among the cases


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_mem_factor_is_truncated_before_the_product():
    """ILUC.hpp:229 reserves min(max_fill_in n, (Integer) mem_factor * nnz): the factor is truncated FIRST, so 1.9 reserves what 1.0
    does (the restatement and the HIP kernel computed (Integer)(mem_factor * nnz))."""
    rng = np.random.default_rng(4242)
    differs = 0
    for it in range(24):
        n = int(rng.integers(30, 120))
        A = (sp.random(n, n, min(1.0, 6.0 / n), random_state=rng, data_rvs=lambda k: rng.standard_normal(k)) + sp.eye(n) * 3.0).asformat("csr")
        A.sort_indices()
        a = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True)
        outcome = {}
        for mf in (1.0, 1.5, 1.9, 2.0, 2.5):
            kw = dict(fill_in=100, threshold=0.0, piv_tol=0.1, rp=-1, mem_factor=mf)
            res = []
            for lib in (O.ref(), O.orc()):
                try:
                    P = O.ILUCP(lib, a, **kw)
                    res.append(("ok", P))
                except O.OracleError as e:
                    res.append(("err", e.code))
            assert res[0][0] == res[1][0], (it, mf, res[0][0], res[1][0])
            if res[0][0] == "ok":
                for x, y in zip(res[0][1].L + res[0][1].U, res[1][1].L + res[1][1].U):
                    assert np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")), (it, mf)
            else:
                assert res[0][1] == res[1][1] == O.ERR_MEMORY
            outcome[mf] = res[0][0]
        assert outcome[1.5] == outcome[1.9] == outcome[1.0] and outcome[2.5] == outcome[2.0]
        differs += int(outcome[1.9] != outcome[2.0])
    assert differs >= 1          # (some matrix fits 2 nnz but not 1 nnz: there 1.9 must fail as 1.0 does)
