"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(ilupp_amd -> libilupp_hip.so), against (1) golden vectors emitted by the real reference and
(2) the pinned CPU oracle on seeded inputs.

Bar (BASELINE.json north_star): integer index arrays bit-exact; fp64 factor values and solve
vectors within 1e-12 relative.  The kernels follow the reference's operation order without FMA
contraction, so we additionally assert BIT-EXACT values wherever that is expected to hold.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import golden_util as G
import matgen

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def _scipy(M):
    d, i, p, is_csr = M
    n = p.shape[0] - 1
    cls = sp.csr_matrix if is_csr else sp.csc_matrix
    return cls((d.copy(), i.copy(), p.copy()), shape=(n, n))


def _fac(Ms):
    return (Ms.data, Ms.indices, Ms.indptr, isinstance(Ms, sp.csr_matrix))


def _vec_close(x, y):
    x, y = np.asarray(x), np.asarray(y)
    if np.array_equal(x, y, equal_nan=True):
        return True
    return bool(np.all(np.abs(x - y) <= RTOL * np.abs(y)))


def _check_ilu0(z, key, M, exact=True):
    import ilupp_amd as ilupp
    A = _scipy(M)
    n = A.shape[0]
    b = G.rhs(n)
    P = ilupp.ILU0Preconditioner(A)
    L, U = [_fac(F) for F in P.factors()]
    if G.has_mat(z, key + "/ilu0_L"):
        Lg, Ug = G.get_mat(z, key + "/ilu0_L"), G.get_mat(z, key + "/ilu0_U")
        assert L[1].dtype == np.int32 and L[2].dtype == np.int32
        assert G.mat_close(L, Lg, RTOL) and G.mat_close(U, Ug, RTOL)
        if exact:
            assert G.mat_equal(L, Lg) and G.mat_equal(U, Ug)
    x = b.copy(); P.apply(x)
    assert _vec_close(x, z[key + "/ilu0_apply"])
    xt = b.copy(); P.apply_trans(xt)
    assert _vec_close(xt, z[key + "/ilu0_apply_trans"])
    if exact:
        assert np.array_equal(x, z[key + "/ilu0_apply"], equal_nan=True)
        assert np.array_equal(xt, z[key + "/ilu0_apply_trans"], equal_nan=True)
    assert P.total_nnz == int(z[key + "/ilu0_total_nnz"])
    # LinearOperator protocol (ilupp/__init__.py:134-150)
    assert np.array_equal(P @ b, x, equal_nan=True)
    assert np.array_equal(P.T @ b, xt, equal_nan=True)
    X2 = np.stack([b, 2 * b], axis=1)
    assert (P @ X2).shape == (n, 2)
    # stand-alone function (binding.cpp:421-430)
    L2, U2 = ilupp.ilu0(A)
    assert G.mat_equal(_fac(L2), L) and G.mat_equal(_fac(U2), U)
    return P


@pytest.mark.parametrize("name", ["laplace", "laplace2d", "random"])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_ilu0_reference_test_matrices(name, fmt):
    z = G.load("reftests.npz")
    key = "%s_%s" % (name, fmt)
    P = _check_ilu0(z, key, G.get_mat(z, key + "/A"))
    n = P.shape[0]
    assert repr(P) == "<%dx%d ILU0Preconditioner with nnz=%d, dtype=float64>" % (n, n, P.total_nnz)


@pytest.mark.parametrize("name", sorted(G.CONFIG_CASES))
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_ilu0_config_shaped(name, fmt):
    z = G.load("configs.npz")
    M, _ = G.config_inputs(name, fmt)
    _check_ilu0(z, "%s_%s" % (name, fmt), M)


def test_ilu0_edges():
    z = G.load("edges.npz")
    _check_ilu0(z, "one", G.get_mat(z, "one/A"))
    _check_ilu0(z, "zeros", G.get_mat(z, "zeros/A"))


def test_reference_unit_tests_ilu0():
    """the reference's own assertions for ILU0 (test/tests.py:197-207, 234-259, 286-306)"""
    import ilupp_amd as ilupp
    n = 50
    d, i, p = matgen.laplace1d(n)
    for fmt in ("csr", "csc"):
        A = sp.csr_matrix((d, i, p), shape=(n, n)).asformat(fmt)
        b = np.ones(n)
        X = np.linspace(0, 1, n + 2)[1:-1]
        x_exact = X * (1 - X) / 2
        P = ilupp.ILU0Preconditioner(A)
        x = b.copy(); P.apply(x)
        assert np.allclose(x, x_exact)
        assert np.allclose(P.T @ (A.T @ x_exact), x_exact)
        L, U = P.factors()
        assert all(r >= c for r, c in zip(*L.nonzero())) and all(r <= c for r, c in zip(*U.nonzero()))
        assert np.allclose(A.toarray(), L.dot(U).toarray())
        assert P.total_nnz == 2 * (2 * n - 1)


def test_ilu0_medium_digests():
    """C1 (2-D 5-point 200x200) and 64^3 against sha256 digests of the reference's output"""
    import ilupp_amd as ilupp
    dg = G.load("digests.json")
    for name, gen in (("poisson2d_200", lambda: matgen.poisson2d(200)), ("poisson3d_64", lambda: matgen.poisson3d(64))):
        e = dg[name]
        d, i, p = gen()
        n = p.shape[0] - 1
        A = sp.csr_matrix((d, i, p), shape=(n, n))
        P = ilupp.ILU0Preconditioner(A)
        L, U = [_fac(F) for F in P.factors()]
        assert G.digest_of(L) == e["ilu0_L"] and G.digest_of(U) == e["ilu0_U"]
        x = np.ones(n); P.apply(x)
        assert G.sha(x) == e["ilu0_apply_ones"]
        xt = np.ones(n); P.apply_trans(xt)
        assert G.sha(xt) == e["ilu0_apply_trans_ones"]
        Pc = ilupp.ILU0Preconditioner(A.tocsc())
        Lc, Uc = [_fac(F) for F in Pc.factors()]
        assert G.digest_of(Lc) == e["csc_ilu0_L"] and G.digest_of(Uc) == e["csc_ilu0_U"]
        xc = np.ones(n); Pc.apply(xc)
        assert G.sha(xc) == e["csc_ilu0_apply_ones"]
    e = dg["random_dd_50000"]
    d, i, p = matgen.random_dd(50000, 19, 25.0, 12345)
    A = sp.csr_matrix((d, i, p), shape=(50000, 50000))
    P = ilupp.ILU0Preconditioner(A)
    L, U = [_fac(F) for F in P.factors()]
    assert G.digest_of(L) == e["ilu0_L"] and G.digest_of(U) == e["ilu0_U"]
    x = np.ones(50000); P.apply(x)
    assert G.sha(x) == e["ilu0_apply_ones"]


_VARIANT_SCRIPT = r"""
import sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np, scipy.sparse as sp
import golden_util as G, matgen
import ilupp_amd as ilupp
dg = G.load("digests.json")
for name, gen in (("poisson2d_200", lambda: matgen.poisson2d(200)), ("poisson3d_64", lambda: matgen.poisson3d(64))):
    e = dg[name]
    d, i, p = gen()
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    P = ilupp.ILU0Preconditioner(A)
    x = np.ones(n); P.apply(x)
    assert G.sha(x) == e["ilu0_apply_ones"], name
    L, U = [(F.data, F.indices, F.indptr, isinstance(F, sp.csr_matrix)) for F in P.factors()]
    assert G.digest_of(L) == e["ilu0_L"] and G.digest_of(U) == e["ilu0_U"], name
    x = np.ones(n); P.apply(x)
    assert G.sha(x) == e["ilu0_apply_ones"], name
    xt = np.ones(n); P.apply_trans(xt)
    assert G.sha(xt) == e["ilu0_apply_trans_ones"], name
print("variant ok")
"""


@pytest.mark.parametrize("env", [{"ILUPP_CLASSIC_ANALYSIS": "1"}, {"ILUPP_NO_PACKED": "1"}, {"ILUPP_NO_TILES": "1"}],
                         ids=["csr_factor_kernel_packed_sweeps", "csr_streaming_kernels_only", "identity_placement"])
def test_ilu0_fallback_kernel_generations(env):
    """the short-row matrices normally take the level-major kernels; the CSR-streaming generation (what any
    matrix the level-major analysis rejects runs on) must give the same bits.  The switches are read once
    per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _VARIANT_SCRIPT % {"root": root, "tests": os.path.join(root, "tests")}
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout + r.stderr


def test_ilu0_poisson128_bitexact_and_refactor():
    """128^3 (64 patches of 16x16 lines, level-major kernels): factors and apply bit-identical to the C restatement
    of the reference; a second numeric factorisation on the same pattern (other values) as well"""
    from oracle import oracle as O
    from ilupp_amd import _native
    import torch
    g = 128
    d, i, p = matgen.poisson3d(g)
    n = p.shape[0] - 1
    orc = O.orc()
    Lo, Uo = orc.ilu0((d, i, p, True))
    xo = orc.apply_lu(Lo, Uo, np.ones(n), O.ID)
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    tx = torch.ones(n, dtype=torch.float64, device=dev)
    P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
    assert np.array_equal(tx.cpu().numpy(), xo)
    (Ld, Li, Lp, _, _, _), (Ud, Ui, Up, _, _, _) = P.factors_info()
    assert np.array_equal(Li, Lo[1]) and np.array_equal(Lp, Lo[2]) and np.array_equal(Ld, Lo[0])
    assert np.array_equal(Ui, Uo[1]) and np.array_equal(Up, Uo[2]) and np.array_equal(Ud, Uo[0])
    # same pattern, new values
    d2 = d * (1.0 + 0.25 * np.cos(np.arange(d.shape[0], dtype=np.float64)))
    d2[d > 0] = d[d > 0] * 1.5
    td2 = torch.from_numpy(d2).to(dev)
    P.refactor_device(td2.data_ptr(), ti.data_ptr(), tp.data_ptr())
    L2, U2 = orc.ilu0((d2, i, p, True))
    x2 = orc.apply_lu(L2, U2, np.ones(n), O.ID)
    tx.fill_(1.0)
    P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
    assert np.array_equal(tx.cpu().numpy(), x2)
    (Ld, _, _, _, _, _), (Ud, _, _, _, _, _) = P.factors_info()
    assert np.array_equal(Ld, L2[0]) and np.array_equal(Ud, U2[0])


@pytest.mark.parametrize("case", ["rand_k7", "rand_k30_long_rows", "rand_dense_rows", "grid_ragged",
                                  "grid_ragged_patches", "grid_2d_ragged", "grid_perturbed_values"])
def test_ilu0_vs_oracle_seeded(case):
    """fresh seeded inputs against the pinned CPU oracle, incl. rows longer than the LDS working row"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    orc = O.orc()
    if case == "rand_k7":
        d, i, p = matgen.random_dd(3000, 7, 5.0, 7)
    elif case == "rand_k30_long_rows":
        d, i, p = matgen.random_dd(1500, 30, 40.0, 11)
    elif case == "rand_dense_rows":
        d, i, p = matgen.random_dd(400, 90, 120.0, 13)        # > 64 entries per row: global working row
    elif case == "grid_ragged_patches":
        d, i, p = matgen.poisson3d(70, 45, 37)       # level-major kernels, patches cut by the grid's edges
    elif case == "grid_2d_ragged":
        d, i, p = matgen.poisson2d(301, 173) if matgen.poisson2d.__code__.co_argcount > 1 else matgen.poisson2d(301)
    elif case == "grid_perturbed_values":
        d, i, p = matgen.poisson3d(40, 40, 40)
        d = d * (1.0 + 0.3 * np.sin(np.arange(d.shape[0], dtype=np.float64) * 0.37))
        d[d > 0] += 2.0
    else:
        d, i, p = matgen.poisson3d(9, 4, 11)
    n = p.shape[0] - 1
    for fmt in ("csr", "csc"):
        M = (d, i, p, True) if fmt == "csr" else matgen.to_csc(d, i, p) + (False,)
        Lo, Uo = orc.ilu0(M)
        P = ilupp.ILU0Preconditioner(_scipy(M))
        L, U = [_fac(F) for F in P.factors()]
        assert G.mat_equal(L, Lo) and G.mat_equal(U, Uo)
        b = G.rhs(n)
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, orc.apply_lu(Lo, Uo, b, O.ID), equal_nan=True)
        xt = b.copy(); P.apply_trans(xt)
        assert np.array_equal(xt, orc.apply_lu(Lo, Uo, b, O.TRANSPOSE), equal_nan=True)


def test_repeated_apply_and_wrong_size():
    import ilupp_amd as ilupp
    d, i, p = matgen.poisson2d(30)
    n = p.shape[0] - 1
    P = ilupp.ILU0Preconditioner(sp.csr_matrix((d, i, p), shape=(n, n)))
    b = G.rhs(n)
    x1 = b.copy(); P.apply(x1)
    for _ in range(5):
        x2 = b.copy(); P.apply(x2)
        assert np.array_equal(x1, x2)
        xt = b.copy(); P.apply_trans(xt)
    with pytest.raises(RuntimeError, match="vector has wrong size for preconditioner!"):
        P.apply(np.ones(n + 1))
    with pytest.raises(RuntimeError, match=r"Expected d \(d\) array for b, got f!"):
        P.apply(np.ones(n, dtype=np.float32))
    assert P.pr.exists and P.pr.special_info == "" and P.pr.memory == 0.0
    assert P.pr.memory_used_calculations == 0.0 and P.pr.memory_allocated_calculations == 0.0


def test_missing_diagonal_is_reported():
    import ilupp_amd as ilupp
    A = sp.csr_matrix(np.array([[2.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 2.0]]))
    with pytest.raises(RuntimeError, match="missing diagonal entry in row 1"):
        ilupp.ILU0Preconditioner(A)


def test_full_size_properties_128():
    """size-independent properties at a size the oracle does not need to touch: L*U reproduces A on
    A's pattern (ILU(0) defining property), apply solves (LU)x=b, transposed apply solves (LU)^T x=b"""
    import ilupp_amd as ilupp
    g = 96
    d, i, p = matgen.poisson3d(g)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    P = ilupp.ILU0Preconditioner(A)
    L, U = P.factors()
    LU = (L @ U).tocsr()
    # on the pattern of A, (LU)_ij == A_ij
    mask = A.copy(); mask.data[:] = 1.0
    R = (LU.multiply(mask) - A)
    assert abs(R).max() < 1e-12
    b = G.rhs(n)
    x = b.copy(); P.apply(x)
    assert np.linalg.norm(LU @ x - b) <= 1e-12 * np.linalg.norm(b) * 50
    xt = b.copy(); P.apply_trans(xt)
    assert np.linalg.norm(LU.T @ xt - b) <= 1e-12 * np.linalg.norm(b) * 50


# ------------------------------------------------------------------------------------------------
# IChol(0)  (SURVEY section 8a, A9): factor bit-exact, LL^T apply == apply_trans
# ------------------------------------------------------------------------------------------------
def _check_ichol0(z, key, S):
    import ilupp_amd as ilupp
    A = _scipy(S)
    n = A.shape[0]
    b = G.rhs(n)
    P = ilupp.IChol0Preconditioner(A)
    (Lm,) = P.factors()
    L = _fac(Lm)
    if G.has_mat(z, key + "/ichol0_L"):
        Lg = G.get_mat(z, key + "/ichol0_L")
        assert G.mat_close(L, Lg, RTOL)
        assert G.mat_equal(L, Lg)
    x = b.copy(); P.apply(x)
    xt = b.copy(); P.apply_trans(xt)
    assert np.array_equal(x, z[key + "/ichol0_apply"], equal_nan=True)
    assert np.array_equal(xt, z[key + "/ichol0_apply_trans"], equal_nan=True)
    assert P.total_nnz == L[2][-1]
    L2 = ilupp.ichol0(A)
    assert G.mat_equal(_fac(L2), L)
    return P


@pytest.mark.parametrize("name", ["laplace", "laplace2d", "random"])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_ichol0_reference_test_matrices(name, fmt):
    z = G.load("reftests.npz")
    key = "%s_%s" % (name, fmt)
    P = _check_ichol0(z, key, G.get_mat(z, key + "/S"))
    n = P.shape[0]
    assert repr(P) == "<%dx%d IChol0Preconditioner with nnz=%d, dtype=float64>" % (n, n, P.total_nnz)


@pytest.mark.parametrize("name", sorted(G.CONFIG_CASES))
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_ichol0_config_shaped(name, fmt):
    z = G.load("configs.npz")
    _, S = G.config_inputs(name, fmt)
    _check_ichol0(z, "%s_%s" % (name, fmt), S)


def test_reference_unit_tests_ichol0():
    """test/tests.py:286-306 for IChol0: one-step solve on the 1-D Laplacian, L L^T = A, total_nnz = 2n-1"""
    import ilupp_amd as ilupp
    n = 50
    d, i, p = matgen.laplace1d(n)
    for fmt in ("csr", "csc"):
        A = sp.csr_matrix((d, i, p), shape=(n, n)).asformat(fmt)
        X = np.linspace(0, 1, n + 2)[1:-1]
        x_exact = X * (1 - X) / 2
        P = ilupp.IChol0Preconditioner(A)
        x = np.ones(n); P.apply(x)
        assert np.allclose(x, x_exact)
        assert np.allclose(P.T @ (A.T @ x_exact), x_exact)
        (L,) = P.factors()
        assert np.allclose(A.toarray(), L.dot(L.T).toarray())
        assert P.total_nnz == 2 * n - 1


def test_ichol0_medium_digests():
    import ilupp_amd as ilupp
    dg = G.load("digests.json")
    for name, gen in (("poisson2d_200", lambda: matgen.poisson2d(200)), ("poisson3d_64", lambda: matgen.poisson3d(64))):
        e = dg[name]
        d, i, p = gen()
        n = p.shape[0] - 1
        P = ilupp.IChol0Preconditioner(sp.csr_matrix((d, i, p), shape=(n, n)))
        (L,) = P.factors()
        assert G.digest_of(_fac(L)) == e["ichol0_L"]
        x = np.ones(n); P.apply(x)
        assert G.sha(x) == e["ichol0_apply_ones"]


# ------------------------------------------------------------------------------------------------
# ILUT: value-dependent patterns -- indices AND values bit-exact
# ------------------------------------------------------------------------------------------------
def _check_ilut(z, key, M, params):
    import ilupp_amd as ilupp
    A = _scipy(M)
    n = A.shape[0]
    b = G.rhs(n)
    for (p, t) in params:
        tag = "ilut_%d_%g" % (p, t)
        P = ilupp.ILUTPreconditioner(A, fill_in=p, threshold=t)
        L, U = [_fac(F) for F in P.factors()]
        Lg, Ug = G.get_mat(z, key + "/" + tag + "_L"), G.get_mat(z, key + "/" + tag + "_U")
        assert G.mat_close(L, Lg, RTOL) and G.mat_close(U, Ug, RTOL), tag
        assert G.mat_equal(L, Lg) and G.mat_equal(U, Ug), tag
        x = b.copy(); P.apply(x)
        xt = b.copy(); P.apply_trans(xt)
        assert np.array_equal(x, z[key + "/" + tag + "_apply"], equal_nan=True), tag
        assert np.array_equal(xt, z[key + "/" + tag + "_apply_trans"], equal_nan=True), tag
        assert P.total_nnz == int(z[key + "/" + tag + "_total_nnz"]), tag
        L2, U2 = ilupp.ilut(A, fill_in=p, threshold=t)
        assert G.mat_equal(_fac(L2), L) and G.mat_equal(_fac(U2), U)


@pytest.mark.parametrize("name", ["laplace", "laplace2d", "random"])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_ilut_reference_test_matrices(name, fmt):
    z = G.load("reftests.npz")
    key = "%s_%s" % (name, fmt)
    _check_ilut(z, key, G.get_mat(z, key + "/A"), G.REFTEST_ILUT)


@pytest.mark.parametrize("name", sorted(G.CONFIG_CASES))
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_ilut_config_shaped(name, fmt):
    z = G.load("configs.npz")
    M, _ = G.config_inputs(name, fmt)
    _check_ilut(z, "%s_%s" % (name, fmt), M, G.CONFIG_ILUT)


def test_ilut_edges_and_zero_pivot():
    import ilupp_amd as ilupp
    z = G.load("edges.npz")
    _check_ilut(z, "one", G.get_mat(z, "one/A"), ((5, 0.1),))
    _check_ilut(z, "zeros", G.get_mat(z, "zeros/A"), ((5, 0.1), (100, 0.0)))     # explicit zeros are stripped (compress(0.0))
    Zm = G.get_mat(z, "zeropivot/A")
    with pytest.raises(RuntimeError, match="ILUT_heap: encountered zero pivot in row %d" % int(z["zeropivot/err_row"])):
        ilupp.ILUTPreconditioner(_scipy(Zm), fill_in=100, threshold=0.0)


def test_ilut_topk_ties():
    """equal magnitudes at the top-k cut: the kept set follows libstdc++'s std::sort, re-run on the GPU"""
    import ilupp_amd as ilupp
    z = G.load("edges.npz")
    d, i, p = matgen.poisson3d(12)
    A = sp.csr_matrix((d, i, p), shape=(1728, 1728))
    for (pp, t) in ((10, 1e-4), (4, 0.0), (20, 1e-6)):
        L, U = ilupp.ilut(A, fill_in=pp, threshold=t)
        assert G.mat_equal(_fac(L), G.get_mat(z, "ties/ilut_%d_%g_L" % (pp, t)))
        assert G.mat_equal(_fac(U), G.get_mat(z, "ties/ilut_%d_%g_U" % (pp, t)))


def test_reference_unit_tests_ilut():
    """test/tests.py:263-283: with threshold=0 ILUT is an exact LU -> one-step solve, L.U = A, total_nnz bound"""
    import ilupp_amd as ilupp
    z = G.load("reftests.npz")
    for name in ("laplace", "random"):
        for fmt in ("csr", "csc"):
            A = _scipy(G.get_mat(z, "%s_%s/A" % (name, fmt)))
            n = A.shape[0]
            x_exact = np.ones(n)
            b = A @ x_exact
            P = ilupp.ILUTPreconditioner(A, threshold=0.0)
            x = b.copy(); P.apply(x)
            assert np.allclose(x, x_exact)
            assert np.allclose(P.T @ (A.T @ x_exact), x_exact)
            L, U = P.factors()
            assert all(r >= c for r, c in zip(*L.nonzero())) and all(r <= c for r, c in zip(*U.nonzero()))
            assert np.allclose(A.toarray(), L.dot(U).toarray())
    d, i, p = matgen.laplace1d(50)
    P = ilupp.ILUTPreconditioner(sp.csr_matrix((d, i, p), shape=(50, 50)), threshold=0.0)
    assert P.total_nnz <= 2 * (2 * 50 - 1)


def test_ilut_medium_digests():
    import ilupp_amd as ilupp
    dg = G.load("digests.json")
    e = dg["poisson2d_200"]
    d, i, p = matgen.poisson2d(200)
    A = sp.csr_matrix((d, i, p), shape=(40000, 40000))
    for (pp, t) in ((10, 1e-4), (5, 0.1)):
        L, U = ilupp.ilut(A, fill_in=pp, threshold=t)
        assert G.digest_of(_fac(L)) == e["ilut_%d_%g_L" % (pp, t)]
        assert G.digest_of(_fac(U)) == e["ilut_%d_%g_U" % (pp, t)]
    e = dg["random_dd_50000"]
    d, i, p = matgen.random_dd(50000, 19, 25.0, 12345)
    A = sp.csr_matrix((d, i, p), shape=(50000, 50000))
    for (pp, t) in ((10, 1e-4), (5, 0.1)):
        P = ilupp.ILUTPreconditioner(A, fill_in=pp, threshold=t)
        L, U = [_fac(F) for F in P.factors()]
        assert G.digest_of(L) == e["ilut_%d_%g_L" % (pp, t)]
        assert G.digest_of(U) == e["ilut_%d_%g_U" % (pp, t)]
        x = np.ones(50000); P.apply(x)
        assert G.sha(x) == e["ilut_%d_%g_apply_ones" % (pp, t)]


# ------------------------------------------------------------------------------------------------
# ICholT: CSC lower factor, diagonal first; apply == apply_trans
# ------------------------------------------------------------------------------------------------
def _check_icholt(z, key, S, params, with_apply=True):
    import ilupp_amd as ilupp
    A = _scipy(S)
    n = A.shape[0]
    b = G.rhs(n)
    for (a, t) in params:
        tag = "icholt_%d_%g" % (a, t)
        P = ilupp.ICholTPreconditioner(A, add_fill_in=a, threshold=t)
        (Lm,) = P.factors()
        L = _fac(Lm)
        Lg = G.get_mat(z, key + "/" + tag + "_L")
        assert not L[3]                     # csc for either input orientation
        assert G.mat_close(L, Lg, RTOL), tag
        assert G.mat_equal(L, Lg), tag
        if with_apply:
            x = b.copy(); P.apply(x)
            assert np.array_equal(x, z[key + "/" + tag + "_apply"], equal_nan=True), tag
            xt = b.copy(); P.apply_trans(xt)
            assert np.array_equal(xt, x, equal_nan=True), tag       # LL^T: apply_trans == apply
        assert P.total_nnz == L[2][-1]
        L2 = ilupp.icholt(A, add_fill_in=a, threshold=t)
        assert G.mat_equal(_fac(L2), L)


@pytest.mark.parametrize("name", ["laplace", "laplace2d", "random"])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_icholt_reference_test_matrices(name, fmt):
    z = G.load("reftests.npz")
    key = "%s_%s" % (name, fmt)
    _check_icholt(z, key, G.get_mat(z, key + "/S"), ((0, 0.0), (5, 1e-3)))


@pytest.mark.parametrize("name", sorted(G.CONFIG_CASES))
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_icholt_config_shaped(name, fmt):
    z = G.load("configs.npz")
    _, S = G.config_inputs(name, fmt)
    _check_icholt(z, "%s_%s" % (name, fmt), S, G.ICHOLT)


def test_icholt_topk_ties_and_unit_tests():
    import ilupp_amd as ilupp
    z = G.load("edges.npz")
    d, i, p = matgen.poisson3d(12)
    S = matgen.symmetrize(d, i, p) + (True,)
    for (a, t) in ((5, 1e-3), (3, 0.0), (12, 0.0)):
        L = ilupp.icholt(_scipy(S), add_fill_in=a, threshold=t)
        assert G.mat_equal(_fac(L), G.get_mat(z, "ties/icholt_%d_%g_L" % (a, t)))
    # test/tests.py:286-306 for ICholT on the 1-D Laplacian
    n = 50
    d, i, p = matgen.laplace1d(n)
    for fmt in ("csr", "csc"):
        A = sp.csr_matrix((d, i, p), shape=(n, n)).asformat(fmt)
        X = np.linspace(0, 1, n + 2)[1:-1]
        x_exact = X * (1 - X) / 2
        P = ilupp.ICholTPreconditioner(A)
        x = np.ones(n); P.apply(x)
        assert np.allclose(x, x_exact)
        assert np.allclose(P.T @ (A.T @ x_exact), x_exact)
        (L,) = P.factors()
        assert np.allclose(A.toarray(), L.dot(L.T).toarray())
        assert P.total_nnz == 2 * n - 1
        assert repr(P) == "<%dx%d ICholTPreconditioner with nnz=%d, dtype=float64>" % (n, n, P.total_nnz)


def test_icholt_medium_digests():
    import ilupp_amd as ilupp
    dg = G.load("digests.json")
    e = dg["poisson2d_200"]
    d, i, p = matgen.poisson2d(200)
    A = sp.csr_matrix((d, i, p), shape=(40000, 40000))
    for (a, t) in ((0, 0.0), (5, 1e-3)):
        P = ilupp.ICholTPreconditioner(A, add_fill_in=a, threshold=t)
        (L,) = P.factors()
        assert G.digest_of(_fac(L)) == e["icholt_%d_%g_L" % (a, t)]
        x = np.ones(40000); P.apply(x)
        assert G.sha(x) == e["icholt_%d_%g_apply_ones" % (a, t)]


# ---------------------------------------------------------------------------------------------
# the dataflow ICholT kernel (icholt_df.hip) and the wave-parallel ILUT kernel (ilut_wp.hip): fresh seeded inputs against
# the pinned oracle, the sequential kernels as A/B, and the capacity fall-backs
# ---------------------------------------------------------------------------------------------
def _spd_from(d, i, p, shift):
    ds, is_, ps = matgen.symmetrize(d, i, p)
    n = ps.shape[0] - 1
    S = sp.csr_matrix((ds, is_, ps), shape=(n, n)) + shift * sp.identity(n, format="csr")
    S = S.tocsr(); S.sort_indices()
    return S.data.astype(np.float64), S.indices.astype(np.int32), S.indptr.astype(np.int32)


def _with_env(name, value, f):
    import os
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return f()
    finally:
        if old is None:
            del os.environ[name]
        else:
            os.environ[name] = old


@pytest.mark.parametrize("case", ["grid_40", "grid_ragged", "rand_spd", "rand_spd_dense_rows", "chain"])
def test_icholt_dataflow_vs_oracle_seeded(case):
    """value-dependent patterns, reaches from dropped entries, linked-list order, ties: bit-exact against the oracle, for
    the LDS capacity classes and the largest capacity class (ILUPP_ICHOLT_CLASS=3) and both input orientations"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    orc = O.orc()
    if case == "grid_40":
        d, i, p = matgen.poisson3d(40)
        params = ((0, 0.0), (5, 1e-3), (2, 0.05), (12, 1e-5))
    elif case == "grid_ragged":
        d, i, p = matgen.poisson3d(23, 9, 31)
        params = ((0, 0.0), (3, 1e-2), (7, 0.0))
    elif case == "rand_spd":
        d, i, p = _spd_from(*matgen.random_dd(4000, 9, 0.0, 21), 12.0)
        params = ((0, 0.0), (4, 1e-3), (10, 1e-6))
    elif case == "rand_spd_dense_rows":
        d, i, p = _spd_from(*matgen.random_dd(600, 40, 0.0, 5), 60.0)       # long working columns: capacity fall-back
        params = ((0, 0.0), (30, 1e-8))
    else:
        d, i, p = matgen.laplace1d(3000)
        params = ((0, 0.0), (2, 0.0))
    n = p.shape[0] - 1
    b = G.rhs(n)
    for fmt in ("csr", "csc"):
        M = (d, i, p, True) if fmt == "csr" else matgen.to_csc(d, i, p) + (False,)
        for a, t in params:
            Lo = orc.icholt(M, a, t)
            P = ilupp.ICholTPreconditioner(_scipy(M), add_fill_in=a, threshold=t)
            L, = P.factors()
            assert G.mat_equal(_fac(L), Lo), (case, fmt, a, t)
            x = b.copy(); P.apply(x)
            assert np.array_equal(x, orc.apply_llt(Lo, b, O.ID), equal_nan=True)
            if fmt == "csr":
                # the largest capacity class of the same kernel (one wave per CU, the whole LDS for its working arrays)
                Ps = _with_env("ILUPP_ICHOLT_CLASS", "3",
                               lambda: ilupp.ICholTPreconditioner(_scipy(M), add_fill_in=a, threshold=t))
                Ls, = Ps.factors()
                assert G.mat_equal(_fac(Ls), Lo), (case, "largest capacity class", a, t)


@pytest.mark.parametrize("case", ["grid_24", "rand_20k", "rand_long_rows", "wide_budget", "huge_budget", "budget_one", "long_pieces", "mass_forget"])
def test_ilut_wave_kernel_vs_oracle_seeded(case):
    """pool/kept/U-slot working row, validated U-row fetches, top-k with ties, rows that outgrow LDS, U rows longer than one
    wave (fill_in > 64), fill budgets beyond the LDS selection queue (fill_in > 256: the largest capacity class): bit-exact
    against the oracle in the default classes and in the largest one (ILUPP_ILUT_BIG=1)"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    orc = O.orc()
    if case == "grid_24":
        d, i, p = matgen.poisson3d(24)
        params = ((10, 1e-4), (5, 0.1), (30, 0.0))
    elif case == "rand_20k":
        d, i, p = matgen.random_dd(20000, 19, 25.0, 99)      # some rows outgrow the LDS pieces
        params = ((10, 1e-4), (3, 1e-2))
    elif case == "rand_long_rows":
        d, i, p = matgen.random_dd(500, 90, 120.0, 13)
        params = ((20, 1e-6), (100, 0.0))
    elif case == "wide_budget":
        d, i, p = matgen.poisson3d(12)
        params = ((100, 0.0), (200, 1e-12))                  # U rows of more than 64 entries
    elif case == "huge_budget":
        d, i, p = matgen.poisson3d(9)
        params = ((300, 0.0), (729, 0.0))                    # budgets beyond kWpSel: exact LU in the largest capacity class
    elif case == "long_pieces":
        # working rows with more than 1 024 U slots (the dropping step's path for pieces that do not fit its registers) and pools that
        # outgrow LDS in the middle of a row (the pool moves and the row goes on)
        d, i, p = matgen.random_dd(2200, 30, 40.0, 5)
        params = ((40, 0.0),)
    elif case == "mass_forget":
        # long left parts with a high threshold: dozens of entries are forgotten with one elimination (more than the list of freed
        # places holds: the pool is rewritten in place instead)
        d, i, p = matgen.random_dd(1500, 160, 200.0, 6)
        params = ((10, 0.3), (10, 0.05))
    else:
        d, i, p = matgen.poisson3d(10)
        params = ((1, 0.1), (2, 0.0))
    n = p.shape[0] - 1
    b = G.rhs(n)
    for fmt in ("csr", "csc"):
        M = (d, i, p, True) if fmt == "csr" else matgen.to_csc(d, i, p) + (False,)
        for fill, t in params:
            Lo, Uo = orc.ilut(M, fill, t)
            P = ilupp.ILUTPreconditioner(_scipy(M), fill_in=fill, threshold=t)
            L, U = [_fac(F) for F in P.factors()]
            assert G.mat_equal(L, Lo) and G.mat_equal(U, Uo), (case, fmt, fill, t)
            x = b.copy(); P.apply(x)
            assert np.array_equal(x, orc.apply_lu(Lo, Uo, b, O.ID), equal_nan=True)
            xt = b.copy(); P.apply_trans(xt)              # (rows > 4 entries: the row-parallel sweep kernel, all three sweep kinds)
            assert np.array_equal(xt, orc.apply_lu(Lo, Uo, b, O.TRANSPOSE), equal_nan=True)
            if fmt == "csr":
                Ps = _with_env("ILUPP_ILUT_BIG", "1", lambda: ilupp.ILUTPreconditioner(_scipy(M), fill_in=fill, threshold=t))
                Ls, Us = [_fac(F) for F in Ps.factors()]
                assert G.mat_equal(Ls, Lo) and G.mat_equal(Us, Uo), (case, "largest capacity class", fill, t)


def test_icholt_full_size_properties_128():
    """128^3 (2.1 M columns): the dataflow kernel against the reference's own C++ when it travelled (oracle/_ref), else
    the C restatement; plus L L^T = A on A's pattern for the no-fill variant's kept entries"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    ref = O.ref() if O.ref_available() else O.orc()
    d, i, p = matgen.poisson3d(128)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    for a, t in ((0, 0.0), (5, 1e-3)):
        P = ilupp.ICholTPreconditioner(A, add_fill_in=a, threshold=t)
        L, = P.factors()
        Lo = ref.icholt((d, i, p, True), a, t)
        assert G.mat_equal(_fac(L), Lo), (a, t)
        assert P.total_nnz == L.nnz
    # apply == apply_trans for LL^T objects (preconditioner_implementation.h:381-394), both equal to the oracle's two sweeps
    b = np.ones(n)
    x = b.copy(); P.apply(x)
    xt = b.copy(); P.apply_trans(xt)
    assert np.array_equal(x, xt)
    assert np.array_equal(x, O.orc().apply_llt(Lo, b, O.ID))


def test_icholt_config_c4_full_size():
    """BASELINE config C4 at its full size (256^3, 16.8 M columns, ICholT(0, 0)): bit-identical to the reference's own C++
    when it travelled (oracle/_ref), else to the C restatement"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    ref = O.ref() if O.ref_available() else O.orc()
    d, i, p = matgen.poisson3d(256)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    P = ilupp.ICholTPreconditioner(A)
    L, = P.factors()
    Lo = ref.icholt((d, i, p, True), 0, 0.0)
    assert G.mat_equal(_fac(L), Lo)
    assert P.total_nnz == L.nnz == 66912256
    # ... by the speculative static kernel for box grids (round 5), and its apply by the wave-exchange sweeps with the vector wave: the two
    # triangular solves at full size against the oracle's (sparse_implementation.h:4040-4087: T2 / T3 on the column-stored factor)
    assert P.pr.path() == "icholt:grid-static"
    b = G.rhs(n)
    x = b.copy(); P.pr.apply(x)
    assert np.array_equal(x, O.orc().apply_llt(Lo, b))
    assert P.pr.kernel_names() == ("k_icholt_grid", "k_sptrsv_wv<1, true>", "k_sptrsv_wv<-1, true>"), P.pr.kernel_names()


def test_icholt_fill_and_threshold_full_size():
    """SURVEY section 8(d)'s other C4 variant at its full size: ICholT(add_fill_in=5, threshold=1e-3) on the 256^3 mesh (the largest capacity
    class of the dataflow kernel, the level-ordered sweeps): factor and apply bit-identical to the reference's own C++ when it travelled
    (oracle/_ref), else to the C restatement"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    ref = O.ref() if O.ref_available() else O.orc()
    d, i, p = matgen.poisson3d(256)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    P = ilupp.ICholTPreconditioner(A, add_fill_in=5, threshold=1e-3)
    L, = P.factors()
    Lo = ref.icholt((d, i, p, True), 5, 1e-3)
    assert G.mat_equal(_fac(L), Lo)
    assert P.total_nnz == L.nnz == len(Lo[0])
    b = np.ones(n)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, O.orc().apply_llt(Lo, b, O.ID))
    x2 = b.copy(); P.apply(x2)                                      # (the second apply runs on the renumbered copy the first one built)
    assert np.array_equal(x, x2)


def test_fuzz_new_kernels():
    """60 random matrices x both orientations x random parameters (budgets 1..100, thresholds 0..0.3, equal magnitudes at the
    top-k cut, indefinite matrices with NaN columns) through ILUT and ICholT: everything bit-identical to the oracle"""
    import fuzz_util
    assert fuzz_util.run(60, first_seed=1000 + int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")), verbose=False) == 0
