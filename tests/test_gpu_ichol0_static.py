"""IChol(0) on the static level-major form (st.hip: k_ichol0_st): lower triangles of 5-/7-point stencils, whose rows only ever divide
by earlier diagonals (IChol.hpp:33-59 with empty dot products).  The factor, apply and apply_trans array-equal to the reference;
matrices the form declines (wider stencils) keep the dataflow kernel over chains."""
import numpy as np
import pytest
import scipy.sparse as sp

import golden_util as G
import matgen

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as O
    return O, (O.ref() if O.ref_available() else O.orc())


def _check(A, expect_static):
    import ilupp_amd as ilupp
    O, ref = _oracle()
    n = A.shape[0]
    P = ilupp.IChol0Preconditioner(A)
    if expect_static is not None:
        assert (P.pr.path() == "ichol0:static-level-major") == expect_static, P.pr.path()
    Lo = ref.ichol0((A.data, A.indices, A.indptr, isinstance(A, sp.csr_matrix)))
    (L,) = P.factors()
    assert G.mat_equal((L.data, L.indices, L.indptr, isinstance(L, sp.csr_matrix)), Lo)
    b = G.rhs(n)
    want = O.orc().apply_llt(Lo, b, O.ID)
    for rep in range(2):
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, want)
        xt = b.copy(); P.apply_trans(xt)
        assert np.array_equal(xt, want)
    if expect_static:
        # (the pair's sweeps: round 4's wave-exchange kernels where every lane fits their classes -- the backward one accumulating in
        # descending column order --, round 2's otherwise)
        names = P.pr.kernel_names()
        assert names[0] == "k_ichol0_st" and names[1:] in ((("k_sptrsv_wv<1, true>", "k_sptrsv_wv<-1, true, true>")), ("k_sptrsv_wx<1, true>", "k_sptrsv_wx<-1, true, true>"),
                                                             ("k_sptrsv_st<1, true>", "k_sptrsv_st<-1, true>")), names


@pytest.mark.parametrize("shape", [(40, 40, 40), (64, 24, 16), (17, 33, 65), (300, 300), (128, 128, 128)])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_ichol0_static_meshes(shape, fmt):
    if len(shape) == 2:
        d, i, p = matgen.poisson2d(*shape)
    else:
        d, i, p = matgen.poisson3d(*shape)
    n = p.shape[0] - 1
    rng = np.random.default_rng(7)
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    # symmetric scaling of the values (keeps the matrix positive definite)
    D = sp.diags(1.0 + 0.5 * rng.random(n))
    A = (D @ A @ D).tocsr()
    A.sort_indices()
    if fmt == "csc":
        A = A.tocsc()
    # (64, 24, 16): the forward-only tiling puts some in-workgroup dependencies more than seven steps back -- declined, chain kernel
    _check(A, None if shape == (64, 24, 16) else True)


def test_ichol0_more_lines_than_lanes():
    d, i, p = matgen.poisson3d(48, 280, 280)
    n = p.shape[0] - 1
    _check(sp.csr_matrix((d, i, p), shape=(n, n)), True)


@pytest.mark.parametrize("dims", [(48, 48), (13, 13, 13)])
def test_ichol0_declined_patterns_keep_the_chain_kernel(dims):
    """9-/27-point stencils: a row shares columns with the rows it divides by -- not 'simple'"""
    d, i, p = matgen.box_stencil(dims)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    A = ((A + A.T) * 0.5).tocsr()
    A.sort_indices()
    _check(A, False)


def test_ichol0_indefinite_gives_nans_where_the_reference_does():
    """a negative pivot: sqrt of a negative number, NaN from there on -- in the same places as the reference"""
    import ilupp_amd as ilupp
    O, ref = _oracle()
    d, i, p = matgen.poisson3d(20)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n)).tolil()
    A[4000, 4000] = -1.0
    A = A.tocsr(); A.sort_indices()
    P = ilupp.IChol0Preconditioner(A)
    Lo = ref.ichol0((A.data, A.indices, A.indptr, True))
    (L,) = P.factors()
    assert np.array_equal(L.indices, Lo[1]) and np.array_equal(L.indptr, Lo[2])
    assert np.array_equal(np.isnan(L.data), np.isnan(Lo[0]))
    ok = ~np.isnan(Lo[0])
    assert np.array_equal(L.data[ok], Lo[0][ok])
