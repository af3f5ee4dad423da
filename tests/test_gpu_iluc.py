"""ILUC on the GPU through the C ABI: factors, apply and apply_trans array-equal to what the REAL reference produced
(tests/golden/iluc.npz) and to the reference / the oracle run live on seeded matrices; the reference's error cases."""
import numpy as np
import pytest
import scipy.sparse as sp

import golden_util as G
import matgen

pytestmark = pytest.mark.gpu

PARAMS = ((5, 0.1), (100, 0.0), (3, 1e-3), (1, 0.0), (20, 1e-2))
CFG = {"p2d_12": lambda: matgen.poisson2d(12), "p3d_7": lambda: matgen.poisson3d(7), "p3d_5_9_4": lambda: matgen.poisson3d(5, 9, 4),
       "rdd_300": lambda: matgen.random_dd(300, k=9), "rdd_600": lambda: matgen.random_dd(600, k=11, diag=3.0)}


def _scipy(M):
    n = M[2].shape[0] - 1
    return (sp.csr_matrix if M[3] else sp.csc_matrix)((M[0], M[1], M[2]), shape=(n, n))


def _fac(F):
    return (F.data, F.indices, F.indptr, isinstance(F, sp.csr_matrix))


def check(z, key, M, params=PARAMS):
    import ilupp_amd as ilupp
    A = _scipy(M)
    n = A.shape[0]
    b = G.rhs(n)
    for (p, t) in params:
        tag = "%s/iluc_%d_%g" % (key, p, t)
        if (tag + "_error") in z.files:
            code, row = (int(v) for v in z[tag + "_error"])
            with pytest.raises(RuntimeError) as ei:
                ilupp.ILUCPreconditioner(A, fill_in=p, threshold=t)
            if code == 1:
                assert "zero pivot" in str(ei.value) and ("k=%d" % row) in str(ei.value)
            else:
                assert "insufficient memory reserved" in str(ei.value)
            continue
        P = ilupp.ILUCPreconditioner(A, fill_in=p, threshold=t)
        L, U = P.factors()
        assert G.mat_equal(_fac(L), G.get_mat(z, tag + "_L")) and G.mat_equal(_fac(U), G.get_mat(z, tag + "_U")), tag
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, z[tag + "_apply"]), tag
        xt = b.copy(); P.apply_trans(xt)
        assert np.array_equal(xt, z[tag + "_apply_trans"]), tag
        assert P.total_nnz == L.nnz + U.nnz
        L2, U2 = ilupp.iluc(A, fill_in=p, threshold=t)
        assert G.mat_equal(_fac(L2), _fac(L)) and G.mat_equal(_fac(U2), _fac(U))


@pytest.mark.parametrize("name", ["laplace", "laplace2d", "random"])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_reference_test_matrices(name, fmt):
    z = G.load("iluc.npz")
    key = "ref_%s_%s" % (name, fmt)
    check(z, key, G.get_mat(z, key + "/A"))


@pytest.mark.parametrize("name", sorted(CFG))
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_config_shaped(name, fmt):
    d, i, p = CFG[name]()
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    M = A if fmt == "csr" else A.tocsc()
    check(G.load("iluc.npz"), "cfg_%s_%s" % (name, fmt), (M.data, M.indices, M.indptr, fmt == "csr"))


def test_edges():
    z = G.load("iluc.npz")
    check(z, "edge_one", (np.array([2.5]), np.array([0], dtype=np.int32), np.array([0, 1], dtype=np.int32), True), ((5, 0.1),))
    check(z, "edge_nopivot", G.get_mat(z, "edge_nopivot/A"), ((5, 0.1),))
    d, i, p = matgen.poisson3d(6)
    check(z, "edge_ties", (d, i, p, True), ((2, 0.0), (3, 0.0), (4, 0.0)))


@pytest.mark.parametrize("case", ["mesh", "random", "random_wide"])
def test_medium_against_the_oracle(case):
    """sizes where thousands of steps are in flight at once"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    ref = O.ref() if O.ref_available() else O.orc()
    if case == "mesh":
        d, i, p = matgen.poisson3d(24); params = ((4, 1e-3), (8, 1e-2))
    elif case == "random":
        d, i, p = matgen.random_dd(30000, k=7, diag=4.0); params = ((6, 1e-3), (3, 0.0))
    else:
        d, i, p = matgen.random_dd(8000, k=25, diag=2.0); params = ((12, 1e-4), (40, 1e-2))
    n = p.shape[0] - 1
    rng = np.random.default_rng(5)
    d = d * (1.0 + 0.3 * rng.random(d.shape[0]))
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    b = G.rhs(n)
    for M, is_csr in ((A, True), (A.tocsc(), False)):
        for (fill, tau) in params:
            Lo, Uo = ref.iluc((M.data, M.indices, M.indptr, is_csr), fill, tau)
            P = ilupp.ILUCPreconditioner(M, fill_in=fill, threshold=tau)
            L, U = P.factors()
            assert G.mat_equal(_fac(L), Lo) and G.mat_equal(_fac(U), Uo), (case, is_csr, fill, tau)
            x = b.copy(); P.apply(x)
            assert np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, O.ID))
            xt = b.copy(); P.apply_trans(xt)
            assert np.array_equal(xt, O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE))


def test_error_path_on_reused_slabs():
    """the step without a pivot writes its row of U: the steps it reaches read that row, not what an older factorisation left in a slab the
    pool hands out again (round 2's abort in this file: a GPU memory fault that depended on what the pool had kept).  A factorisation of
    the same shape with large indices runs first, is released, and its blocks come back for the error case"""
    import ilupp_amd as ilupp
    z = G.load("iluc.npz")
    M = G.get_mat(z, "edge_nopivot/A")
    n = M[2].shape[0] - 1
    rng = np.random.default_rng(2)
    for _ in range(3):
        B = (sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rng, format="csr") + sp.eye(n) * 4.0).tocsr()
        B.sort_indices()
        P = ilupp.ILUCPreconditioner(B, fill_in=5, threshold=0.1)
        del P
        check(z, "edge_nopivot", M, ((5, 0.1),))
