"""Which generation of ILU(0) kernels a matrix class gets WITHOUT any switch (every generation that stays compiled
serves some matrix no newer kernel accepts).  Each case is also compared, as arrays, with the oracle's restatement of the reference
(ILU0.hpp:26-106, sparse_implementation.h:4040-4087)."""
import numpy as np
import pytest
import scipy.sparse as sp

import matgen

pytestmark = pytest.mark.gpu


def _csr(A):
    A = sp.csr_matrix(A); A.sort_indices()
    return A.data.astype(np.float64), A.indices.astype(np.int32), A.indptr.astype(np.int32)


def _mesh_missing_upper():
    d, i, p = matgen.poisson3d(32)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n)).tolil()
    for r in np.random.default_rng(11).integers(40, n - 40, size=300):
        if A[r, r + 32] != 0:
            A[r, r + 32] = 0                                  # some transposed entries are missing
    A = A.tocsr(); A.eliminate_zeros()
    return _csr(A)


def _mesh_with_holes():
    d, i, p = matgen.poisson3d(32)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    keep = np.flatnonzero(np.random.default_rng(5).random(n) > 0.03)
    return _csr(A[keep][:, keep])


def _tridiagonal_blocks():
    b = sp.csr_matrix(matgen.laplace1d(37), shape=(37, 37))
    return _csr(sp.block_diag([b] * 3000))


def _arrow():
    n = 4000
    A = sp.lil_matrix((n, n)); A.setdiag(4.0); A[n - 1, :] = -0.001; A[:, n - 1] = -0.001; A[n - 1, n - 1] = 4.0
    return _csr(A)


CASES = [
    ("box grid 64x48x40", lambda: matgen.poisson3d(64, 48, 40), "ilu0:static-direct", "k_ilu0_wa<0, 4, 4>;k_sptrsv_wv<1, false, false, true>;k_sptrsv_wv<-1, true>"),
    ("2-D grid below the grid analysis' size", lambda: matgen.poisson2d(300, 300), "ilu0:static-direct", "k_ilu0_wa<0, 4, 4>;k_sptrsv_wv<1, false, false, true>;k_sptrsv_wv<-1, true>"),
    ("chains without side neighbours", _tridiagonal_blocks, "ilu0:static-direct", "k_ilu0_sd;k_sptrsv_st<1, false>;k_sptrsv_st<-1, false>"),
    ("mesh with missing transposed entries", _mesh_missing_upper, "ilu0:static-level-major", "k_ilu0_st;k_sptrsv_st<1, false>;k_sptrsv_st<-1, false>"),
    ("mesh with holes", _mesh_with_holes, "ilu0:level-order", ""),          # (round 6: short rows, but 94 levels of 340 rows)
    ("small mesh 20^3", lambda: matgen.poisson3d(20), "ilu0:csr-program", ""),
    ("1-D chain", lambda: matgen.laplace1d(200000), "ilu0:csr", ""),
    ("arrow matrix", _arrow, "ilu0:csr", ""),
    ("27-point box stencil", lambda: matgen.box_stencil((32, 32, 32)), "ilu0:level-order", ""),
    ("random rows, 8 per row", lambda: matgen.random_dd(100000, 8, 25.0, 7), "ilu0:level-order", ""),
]


@pytest.mark.parametrize("name,make,path,kernels", CASES, ids=[c[0] for c in CASES])
def test_path_without_switches(name, make, path, kernels):
    from oracle import oracle as O
    from ilupp_amd import _native
    d, i, p = make()
    d = d * (1.0 + 0.25 * np.random.default_rng(3).random(d.shape[0]))
    n = p.shape[0] - 1
    P = _native.ILU0Preconditioner(d, i, p, True)
    assert P.path() == path, (name, P.path())
    assert ";".join(P.kernel_names()) == kernels, (name, P.kernel_names())
    L, U = O.orc().ilu0((d, i, p, True))
    (ld, li, lp, _, _, _), (ud, ui, up, _, _, _) = P.factors_info()
    assert np.array_equal(lp, L[2]) and np.array_equal(li, L[1]) and np.array_equal(up, U[2]) and np.array_equal(ui, U[1])
    assert np.array_equal(ld, L[0]) and np.array_equal(ud, U[0])
    b = np.random.default_rng(1).random(n)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, O.orc().apply_lu(L, U, b, O.ID))
    xt = b.copy(); P.apply_trans(xt)
    assert np.array_equal(xt, O.orc().apply_lu(L, U, b, O.TRANSPOSE))
