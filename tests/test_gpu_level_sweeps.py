"""The level-ordered one-lane-per-row sweeps (ilupp_amd/csrc/sptrsv_lvl.hip): factors whose rows are too long for the level-major
records -- ILU(0) of 9- and 27-point stencils, ILUT / ILUC factors, ICholT with fill -- are renumbered by dependency level at their
first sweep.  apply and apply_trans, first call (builds the renumbered copy) and later calls, must have the bits of the reference's
triangular solves (matrix_sparse::triangular_solve, sparse.hpp:4040-4075, through the oracle) on the same factors; the factors
themselves are compared with the reference / the oracle too."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import golden_util as G
import matgen

pytestmark = pytest.mark.gpu


def _fac(Ms):
    return (Ms.data, Ms.indices, Ms.indptr, isinstance(Ms, sp.csr_matrix))


def _oracle():
    from oracle import oracle as O
    return O, (O.ref() if O.ref_available() else O.orc())


def _applies(P, want_id, want_tr, b):
    for rep in range(3):
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, want_id), ("apply", rep)
        xt = b.copy(); P.apply_trans(xt)
        assert np.array_equal(xt, want_tr), ("apply_trans", rep)


@pytest.mark.parametrize("dims", [(48, 48), (13, 13, 13), (3, 700), (200, 7)])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_ilu0_box_stencils(dims, fmt):
    import ilupp_amd as ilupp
    O, ref = _oracle()
    d, i, p = matgen.box_stencil(dims)
    n = p.shape[0] - 1
    rng = np.random.default_rng(11)
    d = d * (1.0 + 0.2 * rng.random(d.shape[0]))
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    if fmt == "csc":
        A = A.tocsc()
    P = ilupp.ILU0Preconditioner(A)
    Lo, Uo = ref.ilu0((A.data, A.indices, A.indptr, fmt == "csr"))
    L, U = P.factors()
    assert G.mat_equal(_fac(L), Lo) and G.mat_equal(_fac(U), Uo)
    b = G.rhs(n)
    _applies(P, O.orc().apply_lu(Lo, Uo, b, O.ID), O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE), b)


@pytest.mark.parametrize("case", ["mesh", "random", "random_wide"])
def test_ilut_iluc_factors(case):
    import ilupp_amd as ilupp
    O, ref = _oracle()
    if case == "mesh":
        d, i, p = matgen.poisson3d(20); fill, tau = 8, 1e-3
    elif case == "random":
        d, i, p = matgen.random_dd(20000, k=9, diag=6.0); fill, tau = 10, 1e-4
    else:
        d, i, p = matgen.random_dd(5000, k=25, diag=3.0); fill, tau = 30, 1e-5
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    b = G.rhs(n)
    for M, is_csr in ((A, True), (A.tocsc(), False)):
        P = ilupp.ILUTPreconditioner(M, fill_in=fill, threshold=tau)
        Lo, Uo = ref.ilut((M.data, M.indices, M.indptr, is_csr), fill, tau)
        L, U = P.factors()
        assert G.mat_equal(_fac(L), Lo) and G.mat_equal(_fac(U), Uo)
        _applies(P, O.orc().apply_lu(Lo, Uo, b, O.ID), O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE), b)
        del P
        P = ilupp.ILUCPreconditioner(M, fill_in=fill, threshold=tau)
        Lo, Uo = ref.iluc((M.data, M.indices, M.indptr, is_csr), fill, tau)
        L, U = P.factors()
        assert G.mat_equal(_fac(L), Lo) and G.mat_equal(_fac(U), Uo)
        _applies(P, O.orc().apply_lu(Lo, Uo, b, O.ID), O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE), b)
        del P


@pytest.mark.parametrize("g", [14, 24])
def test_icholt_with_fill(g):
    import ilupp_amd as ilupp
    O, ref = _oracle()
    d, i, p = matgen.poisson3d(g)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    b = G.rhs(n)
    for (a, t) in ((5, 1e-3), (12, 1e-5)):
        P = ilupp.ICholTPreconditioner(A, add_fill_in=a, threshold=t)
        Lo = ref.icholt((A.data, A.indices, A.indptr, True), a, t)
        (L,) = P.factors()
        assert G.mat_equal(_fac(L), Lo)
        want = O.orc().apply_llt(Lo, b, O.ID)
        _applies(P, want, want, b)


def test_refactor_drops_the_renumbered_copy():
    """ILU(0) on a 9-point stencil, new values on the same pattern: the sweeps must use the new factors"""
    import torch
    from ilupp_amd import _native
    O, ref = _oracle()
    d, i, p = matgen.box_stencil((40, 40))
    n = p.shape[0] - 1
    dev = torch.device("cuda", 0)
    ti, tp = torch.from_numpy(i).to(dev), torch.from_numpy(p).to(dev)
    td = torch.from_numpy(d).to(dev)
    torch.cuda.synchronize()
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    b = G.rhs(n)
    x = b.copy(); P.apply(x)
    Lo, Uo = ref.ilu0((d, i, p, True))
    assert np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, O.ID))
    d2 = d * (1.0 + 0.3 * np.random.default_rng(3).random(d.shape[0]))
    td2 = torch.from_numpy(d2).to(dev)
    torch.cuda.synchronize()
    P.refactor_device(td2.data_ptr(), ti.data_ptr(), tp.data_ptr())
    Lo, Uo = ref.ilu0((d2, i, p, True))
    for use, f in ((O.ID, P.apply), (O.TRANSPOSE, P.apply_trans)):
        x = b.copy(); f(x)
        assert np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, use))


def test_ilu0_27_point_96_cubed():
    """ILU(0) on a 27-point 96^3 matrix (n = 884 736, 23 M entries; eliminations meet off-diagonal entries: the general merge of
    ILU0.hpp:8-23) through the device-resident constructor: L, U, apply and apply_trans array-equal to the reference"""
    import torch
    from ilupp_amd import _native
    O, ref = _oracle()
    d, i, p = matgen.box_stencil((96, 96, 96))
    n = p.shape[0] - 1
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    torch.cuda.synchronize()
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    assert P.path() == "ilu0:level-order"
    Lo, Uo = ref.ilu0((d, i, p, True))
    L, U = P.factors_info()
    assert G.mat_equal(tuple(L[:4]), Lo) and G.mat_equal(tuple(U[:4]), Uo)
    b = G.rhs(n)
    for use in (O.ID, O.TRANSPOSE):
        want = O.orc().apply_lu(Lo, Uo, b, use)
        for rep in range(2):
            x = torch.from_numpy(b.copy()).to(dev)
            torch.cuda.synchronize()
            P.apply_device(x.data_ptr(), n, transpose=(use == O.TRANSPOSE), sync=True)
            assert np.array_equal(x.cpu().numpy(), want), (use, rep)


def test_fuzz_level_order():
    """random matrices (5 to 40 entries per row, some with a band: chains) through ILU(0), IChol0 and ILUC and their applies"""
    import fuzz_lvl
    assert fuzz_lvl.run(6, first_seed=int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")), verbose=False) == 0
