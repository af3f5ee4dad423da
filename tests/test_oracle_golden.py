"""CPU tests: the plain-C oracle (oracle/ilupp_oracle.c) against golden vectors emitted by the real
reference (tests/golden/make_golden.py).  Bit-exact on every array.  This is what PINS the oracle."""
import os

import numpy as np
import pytest

import golden_util as G
from oracle import oracle as O

orc = O.orc()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_family(z, key, M, S, ilut_params, icholt_params):
    n = M[2].shape[0] - 1
    b = G.rhs(n)
    L, U = orc.ilu0(M)
    if G.has_mat(z, key + "/ilu0_L"):
        assert G.mat_equal(L, G.get_mat(z, key + "/ilu0_L"))
        assert G.mat_equal(U, G.get_mat(z, key + "/ilu0_U"))
    assert np.array_equal(orc.apply_lu(L, U, b, O.ID), z[key + "/ilu0_apply"], equal_nan=True)
    assert np.array_equal(orc.apply_lu(L, U, b, O.TRANSPOSE), z[key + "/ilu0_apply_trans"], equal_nan=True)
    assert int(L[2][-1]) + int(U[2][-1]) == int(z[key + "/ilu0_total_nnz"])
    for (p, t) in ilut_params:
        tag = "ilut_%d_%g" % (p, t)
        L, U = orc.ilut(M, p, t)
        assert G.mat_equal(L, G.get_mat(z, key + "/" + tag + "_L")), tag
        assert G.mat_equal(U, G.get_mat(z, key + "/" + tag + "_U")), tag
        assert np.array_equal(orc.apply_lu(L, U, b, O.ID), z[key + "/" + tag + "_apply"], equal_nan=True)
        assert np.array_equal(orc.apply_lu(L, U, b, O.TRANSPOSE), z[key + "/" + tag + "_apply_trans"], equal_nan=True)
        # ILUT convention: unit diagonal not counted (preconditioner_implementation.h:1035-1039)
        assert int(L[2][-1]) + int(U[2][-1]) - n == int(z[key + "/" + tag + "_total_nnz"])
    if S is not None:
        L = orc.ichol0(S)
        assert G.mat_equal(L, G.get_mat(z, key + "/ichol0_L"))
        assert np.array_equal(orc.apply_llt(L, b, O.ID), z[key + "/ichol0_apply"], equal_nan=True)
        assert np.array_equal(orc.apply_llt(L, b, O.TRANSPOSE), z[key + "/ichol0_apply_trans"], equal_nan=True)
        for (a, t) in icholt_params:
            tag = "icholt_%d_%g" % (a, t)
            L = orc.icholt(S, a, t)
            assert G.mat_equal(L, G.get_mat(z, key + "/" + tag + "_L")), tag
            assert np.array_equal(orc.apply_llt(L, b, O.ID), z[key + "/" + tag + "_apply"], equal_nan=True)


@pytest.mark.parametrize("name", ["laplace", "laplace2d", "random"])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_reference_test_matrices(name, fmt):
    z = G.load("reftests.npz")
    key = "%s_%s" % (name, fmt)
    _check_family(z, key, G.get_mat(z, key + "/A"), G.get_mat(z, key + "/S"), G.REFTEST_ILUT, ((0, 0.0), (5, 1e-3)))


@pytest.mark.parametrize("name", sorted(G.CONFIG_CASES))
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_config_shaped(name, fmt):
    z = G.load("configs.npz")
    M, S = G.config_inputs(name, fmt)
    _check_family(z, "%s_%s" % (name, fmt), M, S, G.CONFIG_ILUT, G.ICHOLT)


def test_edges():
    z = G.load("edges.npz")
    M = G.get_mat(z, "one/A")
    _check_family(z, "one", M, M, ((5, 0.1),), ((0, 0.0),))
    M = G.get_mat(z, "zeros/A")
    _check_family(z, "zeros", M, None, ((5, 0.1), (100, 0.0)), ())
    Zm = G.get_mat(z, "zeropivot/A")
    with pytest.raises(O.OracleError) as e:
        orc.ilut(Zm, 100, 0.0)
    assert e.value.code == O.ERR_ZERO_PIVOT and e.value.row == int(z["zeropivot/err_row"])
    assert "zero pivot in row %d" % e.value.row in str(e.value)


def test_topk_ties():
    """structured grids produce equal magnitudes at the top-k cut; the kept set is defined by
    libstdc++'s std::sort (dropping.hpp:25-26), restated in the oracle."""
    import matgen
    z = G.load("edges.npz")
    d, i, p = matgen.poisson3d(12)
    M = (d, i, p, True)
    for (pp, t) in ((10, 1e-4), (4, 0.0), (20, 1e-6)):
        L, U = orc.ilut(M, pp, t)
        assert G.mat_equal(L, G.get_mat(z, "ties/ilut_%d_%g_L" % (pp, t)))
        assert G.mat_equal(U, G.get_mat(z, "ties/ilut_%d_%g_U" % (pp, t)))
    S = matgen.symmetrize(d, i, p) + (True,)
    for (a, t) in ((5, 1e-3), (3, 0.0), (12, 0.0)):
        L = orc.icholt(S, a, t)
        assert G.mat_equal(L, G.get_mat(z, "ties/icholt_%d_%g_L" % (a, t)))


def test_medium_digests():
    import matgen
    dg = G.load("digests.json")
    for name, gen in (("poisson2d_200", lambda: matgen.poisson2d(200)), ("poisson3d_64", lambda: matgen.poisson3d(64))):
        e = dg[name]
        d, i, p = gen()
        M = (d, i, p, True)
        n = p.shape[0] - 1
        assert G.digest_of(M) == e["input"]
        L, U = orc.ilu0(M)
        assert G.digest_of(L) == e["ilu0_L"] and G.digest_of(U) == e["ilu0_U"]
        assert G.sha(orc.apply_lu(L, U, np.ones(n), O.ID)) == e["ilu0_apply_ones"]
        assert G.sha(orc.apply_lu(L, U, np.ones(n), O.TRANSPOSE)) == e["ilu0_apply_trans_ones"]
        Lc = orc.ichol0(M)
        assert G.digest_of(Lc) == e["ichol0_L"]
        assert G.sha(orc.apply_llt(Lc, np.ones(n), O.ID)) == e["ichol0_apply_ones"]
        for (a, t) in ((0, 0.0), (5, 1e-3)):
            Lt = orc.icholt(M, a, t)
            assert G.digest_of(Lt) == e["icholt_%d_%g_L" % (a, t)]
    e = dg["random_dd_50000"]
    d, i, p = matgen.random_dd(50000, 19, 25.0, 12345)
    M = (d, i, p, True)
    assert G.digest_of(M) == e["input"]
    for (pp, t) in ((10, 1e-4), (5, 0.1)):
        L, U = orc.ilut(M, pp, t)
        assert G.digest_of(L) == e["ilut_%d_%g_L" % (pp, t)]
        assert G.digest_of(U) == e["ilut_%d_%g_U" % (pp, t)]
        assert G.sha(orc.apply_lu(L, U, np.ones(50000), O.ID)) == e["ilut_%d_%g_apply_ones" % (pp, t)]


def test_sort_restatement_matches_insertion_regime():
    """<=16 candidates: libstdc++ std::sort degenerates to a stable insertion sort (stl_algo.h:1855)."""
    rng = np.random.default_rng(0)
    for n in range(0, 17):
        k = rng.integers(0, 3, n).astype(np.float64) * rng.choice([-1.0, 1.0], n)
        got = orc.sort_slots_by_abs_desc(k)
        want = np.argsort(-np.abs(k), kind="stable").astype(np.int32)
        assert np.array_equal(got, want)


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_vs_reference_live():
    """Where the real reference is built (build container), compare live on fresh inputs too."""
    import matgen
    ref = O.ref()
    for seed in (1, 2, 3):
        d, i, p = matgen.random_dd(400, 7, 3.0 + seed, seed)
        M = (d, i, p, True)
        S = matgen.symmetrize(d, i, p) + (True,)
        for A in (M, matgen.to_csc(d, i, p) + (False,)):
            assert all(G.mat_equal(a, b) for a, b in zip(orc.ilu0(A), ref.ilu0(A)))
            assert all(G.mat_equal(a, b) for a, b in zip(orc.ilut(A, 6, 1e-3), ref.ilut(A, 6, 1e-3)))
        assert G.mat_equal(orc.ichol0(S), ref.ichol0(S))
        assert G.mat_equal(orc.icholt(S, 3, 1e-2), ref.icholt(S, 3, 1e-2))
    rng = np.random.default_rng(5)
    for n in (17, 33, 100, 1000):
        k = rng.integers(0, 5, n).astype(np.float64)
        assert np.array_equal(orc.sort_slots_by_abs_desc(k), ref.sort_slots_by_abs_desc(k))


def test_restatement_vs_reference_on_fuzz_matrices():
    """the matrices of the GPU fuzz (tests/fuzz_util.py: ties, budgets 1..100, indefinite matrices whose ICholT factors lose
    diagonals or whole columns) through the C restatement and through the reference's own C++: factors and applies identical,
    so the fuzz's oracle is pinned on these inputs too.  Needs oracle/_ref (built from /root/reference where that exists)."""
    from oracle import oracle as O
    if not O.ref_available():
        pytest.skip("oracle/_ref not built")
    import scipy.sparse as sp
    orc, ref = O.orc(), O.ref()
    nseeds = 120
    bad = 0; deg = 0; tot = 0
    for seed in range(2000, 2000 + nseeds):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 2, 3, 7, 20, 64, 65, 150, 400, 1500, 2500]))
        dens = float(rng.choice([0.5, 2.0, 5.0, 12.0])) / max(n, 1)
        R = sp.random(n, n, density=min(1.0, dens), random_state=rng, format='csr')
        if rng.random() < 0.3:
            R.data = np.round(R.data * 4) / 4
        A = (R + sp.identity(n) * float(rng.choice([1.0, 4.0, 25.0]))).tocsr(); A.sort_indices()
        S = ((A + A.T) * 0.5 + sp.identity(n) * float(rng.choice([2.0, 2.0, 0.0, -3.0]))).tocsr(); S.sort_indices()
        for fmt in ('csr', 'csc'):
            Af = A if fmt == 'csr' else A.tocsc(); Sf = S if fmt == 'csr' else S.tocsc()
            Af.sort_indices(); Sf.sort_indices()
            Mi = (Af.data.astype(np.float64), Af.indices.astype(np.int32), Af.indptr.astype(np.int32), fmt == 'csr')
            Ms = (Sf.data.astype(np.float64), Sf.indices.astype(np.int32), Sf.indptr.astype(np.int32), fmt == 'csr')
            fill = int(rng.choice([1, 2, 3, 5, 10, 17, 70, 100])); tau = float(rng.choice([0.0, 1e-6, 1e-3, 0.05, 0.3]))
            add = int(rng.choice([0, 1, 2, 5, 9, 40])); tau2 = float(rng.choice([0.0, 1e-6, 1e-3, 0.05, 0.3]))
            b = np.cos(np.arange(n, dtype=np.float64)) + 1.5
            tot += 1
            try:
                Lo, Uo = orc.ilut(Mi, fill, tau); e1 = None
            except O.OracleError as e: e1 = e
            try:
                Lr, Ur = ref.ilut(Mi, fill, tau); e2 = None
            except O.OracleError as e: e2 = e
            if (e1 is None) != (e2 is None): bad += 1; print('ILUT error mismatch', seed, fmt)
            elif e1 is None:
                ok = all(np.array_equal(a, c, equal_nan=True) for a, c in zip(Lo[:3] + Uo[:3], Lr[:3] + Ur[:3]))
                ok = ok and np.array_equal(orc.apply_lu(Lo, Uo, b, O.TRANSPOSE), ref.apply_lu(Lr, Ur, b, O.TRANSPOSE), equal_nan=True)
                if not ok: bad += 1; print('ILUT mismatch', seed, fmt)
            Lo = orc.icholt(Ms, add, tau2); Lr = ref.icholt(Ms, add, tau2)
            ok = all(np.array_equal(a, c, equal_nan=True) for a, c in zip(Lo[:3], Lr[:3]))
            d = bool(np.any(np.diff(Lo[2]) == 0)) or any(Lo[1][Lo[2][j]] != j for j in range(n) if Lo[2][j+1] > Lo[2][j])
            deg += d
            trailing_empty = n > 0 and Lo[2][-1] == Lo[2][-2]
            if ok and not trailing_empty:
                ok = np.array_equal(orc.apply_llt(Lo, b, O.ID), ref.apply_llt(Lr, b, O.ID), equal_nan=True)
            if not ok: bad += 1; print('ICHOLT mismatch', seed, fmt, 'degenerate', d)
    assert bad == 0
    assert deg > 0            # the degenerate cases are in the sample


def test_golden_ml_fixture_is_complete():
    """tests/golden/ml.npz (make_golden_ml.py: BiCGstab iterates and multilevel ILU++ applies from the REAL reference): every case has
    its apply, apply_trans and (levels, total_nnz); where oracle/_ref travelled, a sample is re-derived from it live"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "ml.npz"))
    infos = [k for k in z.files if k.endswith("_info")]
    assert len(infos) == 120 and all(k[:-5] + "_apply" in z.files and k[:-5] + "_apply_trans" in z.files for k in infos)
    assert all(("bicg/x_%d" % k) in z.files for k in range(1, 7))
    assert {int(z[k][0]) for k in infos} >= {1, 2, 3}            # one-, two- and three-level factorisations are pinned
