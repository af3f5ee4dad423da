"""oracle (plain-C restatement of ILUC2, oracle/ilupp_oracle.c) against the golden vectors the REAL reference produced
(tests/golden/iluc.npz, tests/golden/make_golden_iluc.py) and against the algorithm as the reference's own tests state it
(test/tests.py:164-192: a dense Crout ILU with the same dropping rule), bit-exact / to 1e-12."""
import numpy as np
import pytest
import scipy.sparse as sp

import golden_util as G
import matgen
from oracle import oracle as O

PARAMS = ((5, 0.1), (100, 0.0), (3, 1e-3), (1, 0.0), (20, 1e-2))
CFG = {"p2d_12": lambda: matgen.poisson2d(12), "p3d_7": lambda: matgen.poisson3d(7), "p3d_5_9_4": lambda: matgen.poisson3d(5, 9, 4),
       "rdd_300": lambda: matgen.random_dd(300, k=9), "rdd_600": lambda: matgen.random_dd(600, k=11, diag=3.0)}


def check(z, key, M, params=PARAMS, f=None):
    f = f or O.orc().iluc
    n = M[2].shape[0] - 1
    b = G.rhs(n)
    for (p, t) in params:
        tag = "%s/iluc_%d_%g" % (key, p, t)
        if (tag + "_error") in z.files:
            code, row = (int(v) for v in z[tag + "_error"])
            with pytest.raises(O.OracleError) as ei:
                f(M, p, t)
            assert ei.value.code == code and (code != O.ERR_ZERO_PIVOT or ei.value.row == row)
            continue
        L, U = f(M, p, t)
        assert G.mat_equal(L, G.get_mat(z, tag + "_L")) and G.mat_equal(U, G.get_mat(z, tag + "_U")), tag
        assert np.array_equal(O.orc().apply_lu(L, U, b, O.ID), z[tag + "_apply"])
        assert np.array_equal(O.orc().apply_lu(L, U, b, O.TRANSPOSE), z[tag + "_apply_trans"])


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


def test_restatement_vs_reference_live():
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (no reference in this environment)")
    ref = O.ref()
    for seed in range(6):
        d, i, p = matgen.random_dd(400 + 37 * seed, k=5 + seed, diag=2.0 + seed, seed=100 + seed)
        n = p.shape[0] - 1
        A = sp.csr_matrix((d, i, p), shape=(n, n))
        for M, is_csr in ((A, True), (A.tocsc(), False)):
            for (fill, tau) in ((4, 1e-2), (12, 1e-4), (2, 0.0)):
                a = (M.data, M.indices, M.indptr, is_csr)
                assert all(G.mat_equal(x, y) for x, y in zip(O.orc().iluc(a, fill, tau), ref.iluc(a, fill, tau)))
