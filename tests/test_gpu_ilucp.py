"""GPU parity tests of SURVEY section 8 (f4), first half: ILUCPPreconditioner (ilupp_amd/csrc/ilucp.hip; reference ILUCP4, ILUC.hpp:212-370,
the permuted solves sparse_implementation.h:4166-4253, binding.cpp:343-356).  Bit for bit:
* against tests/golden/ilucp.npz (the REAL reference on its own test matrices and config-shaped ones, CSR and CSC, five parameter sets):
  both factors, the permutation, apply, apply_trans;
* against the oracle on random matrices with random parameters (bounded fill, pivot tolerances, the row from which every step pivots,
  the reservation error), and through the Python class (factors(), permutations(), LinearOperator protocol) like the reference's
  generic tests (test/tests.py:258-330)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import ml_cases as C

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(100, 0.1, 0.1), (100, 0.0, 0.0), (3, 1e-3, 1.0), (8, 1e-2, 0.5), (1, 0.1, 0.1)]
NAMES = ["laplace2d", "random", "rdd_300", "weak_200", "offdiag_150"]


def _native_cp(a, fill, thr, tol, rp=-1, mem=10.0):
    from ilupp_amd import _native
    return _native.ILUCPPreconditioner(a[0], a[1], a[2], a[3], fill, thr, tol, rp, mem)


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("name", NAMES)
def test_reference_vectors(name, fmt):
    gold = np.load(os.path.join(HERE, "golden", "ilucp.npz"))
    key = "%s_%s" % (name, fmt)
    a = (gold[key + "/data"], gold[key + "/indices"], gold[key + "/indptr"], fmt == "csr")
    b = C.rhs(a[2].shape[0] - 1)
    for fill, thr, tol in CASES:
        tag = "%s/f%d_t%g_p%g" % (key, fill, thr, tol)
        P = _native_cp(a, fill, thr, tol)
        L, U, perm = P.raw()
        for nm, arr in zip(("L_data", "L_indices", "L_indptr", "U_data", "U_indices", "U_indptr"), L + U):
            assert np.array_equal(arr, gold[tag + "/" + nm], equal_nan=(arr.dtype.kind == "f")), (tag, nm)
        assert np.array_equal(perm, gold[tag + "/perm"]), tag
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, gold[tag + "/apply"], equal_nan=True), tag
        x = b.copy(); P.apply_trans(x)
        assert np.array_equal(x, gold[tag + "/apply_trans"], equal_nan=True), tag


def test_fuzz_against_the_oracle():
    from oracle import oracle as O
    off = int(os.environ.get("ILUPP_FUZZ_OFFSET", "0"))             # other seeds: profiles/tools/fuzz_more.sh
    rng = np.random.default_rng(4242 + off)
    failures = 0
    for it in range(70):
        n = int(rng.integers(2, 400))
        A = (sp.random(n, n, min(1.0, rng.uniform(2, 9) / n), random_state=rng, data_rvs=lambda k: rng.standard_normal(k))
             + sp.eye(n) * float(rng.choice([0.0, 0.3, 3.0]))).asformat("csr" if it % 2 else "csc")
        A.sort_indices()
        a = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), bool(it % 2))
        kw = dict(fill_in=int(rng.choice([1, 2, 5, 100])), threshold=float(rng.choice([0.0, 1e-3, 0.1, 0.5])), piv_tol=float(rng.choice([0.0, 0.1, 1.0])),
                  rp=int(rng.choice([-1, 0, n // 2])), mem_factor=float(rng.choice([10.0, 10.0, 1.0])))
        try:
            Q = O.ILUCP(O.orc(), a, **kw)
        except O.OracleError as e:
            assert e.code == O.ERR_MEMORY
            with pytest.raises(RuntimeError, match="Insufficient memory reserved"):
                _native_cp(a, kw["fill_in"], kw["threshold"], kw["piv_tol"], kw["rp"], kw["mem_factor"])
            failures += 1
            continue
        P = _native_cp(a, kw["fill_in"], kw["threshold"], kw["piv_tol"], kw["rp"], kw["mem_factor"])
        L, U, perm = P.raw()
        assert np.array_equal(perm, Q.perm), (it, kw)
        for x, y in zip(L + U, Q.L + Q.U):
            assert np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")), (it, kw)
        assert P.total_nnz == len(Q.L[0]) + len(Q.U[0]) and P.zero_pivots == Q.zero_pivots
        b = C.rhs(n)
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, Q.apply(b), equal_nan=True), (it, kw)
        x = b.copy(); P.apply_trans(x)
        assert np.array_equal(x, Q.apply(b, O.TRANSPOSE), equal_nan=True), (it, kw)
    assert failures >= 1 or off


def test_class_like_the_reference_tests():
    """test/tests.py:258-330: every preconditioner class solves the reference's test problems (exactly, with threshold 0 and unbounded fill),
    offers factors() and -- the pivoting ones -- permutations()"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    n = 60
    A = (sp.random(n, n, density=5 / n, random_state=39273) + 10.0 * sp.eye(n)).tocsc()
    x_exact = np.ones(n)
    b = A @ x_exact
    for fmt in ("csc", "csr"):
        M = A.asformat(fmt)
        P = ilupp.ILUCPPreconditioner(M, threshold=0.0, fill_in=n)
        assert np.allclose(P @ b, x_exact) and np.allclose(P.T @ (M.T @ x_exact), x_exact)
        x = b.copy(); P.apply(x)
        assert np.allclose(x, x_exact)
        fl, fr = P.factors()
        assert fl.shape == fr.shape == (n, n) and P.total_nnz == fl.nnz + fr.nnz
        pl, pr = P.permutations()
        perm = pr if fmt == "csc" else pl
        assert (pl is None) != (pr is None) and sorted(perm) == list(range(n))
        # L U = A P for COLUMN input (U's columns as stored: original indices), P^T U^T L^T = A for ROW input
        if fmt == "csc":
            assert np.allclose((fl @ fr).toarray(), M.toarray())
        else:
            assert np.allclose((fl @ fr).toarray(), M.toarray())
    W = C.weak_random(150, 0.05, 0.2, 11).tocsr()
    P = ilupp.ILUCPPreconditioner(W, piv_tol=1.0)
    Q = O.ILUCP(O.orc(), (W.data, W.indices.astype(np.int32), W.indptr.astype(np.int32), True), piv_tol=1.0)
    assert np.array_equal(P.permutations()[0], Q.perm) and P.permutations()[1] is None
    assert np.array_equal(P @ C.rhs(150), Q.apply(C.rhs(150)))
    assert repr(P).startswith("<150x150 ILUCPPreconditioner with nnz=%d" % P.total_nnz)
    with pytest.raises(RuntimeError, match="vector has wrong size for preconditioner!"):
        P.pr.apply(np.ones(149))
