"""ilupp_amd -- MI355X-native incomplete LU / incomplete Cholesky preconditioners.

Drop-in for the hot path of c-f-h/ilupp: the same Python surface as the reference's
``ilupp/__init__.py`` (class names, argument names and defaults, LinearOperator protocol,
``apply``/``apply_trans`` in place, ``factors()``, ``total_nnz``, ``repr``, exception types), with the
factorisation and the L/U triangular solves running as hand-written HIP kernels on the GPU through
the C ABI in ``include/ilupp_hip.h``.  There is no CPU fallback.

    import ilupp_amd as ilupp
    P = ilupp.ILU0Preconditioner(A)          # scipy CSR/CSC in
    y = P @ x                                # or P.apply(x) in place, P.T @ x
"""
import numpy as np
import scipy.sparse
import scipy.sparse.linalg

from . import _native as _ilupp

__version__ = "0.1.0"

_index_size = _ilupp.index_size()        # 4: int32 build (reference default, declarations.h:49-53)
if _index_size == 4:
    _index_dtype = np.dtype(np.int32)
elif _index_size == 8:
    _index_dtype = np.dtype(np.int64)
else:
    raise RuntimeError('invalid index type size %d' % _index_size)


def _upcast_indices(idx):
    """Same contract as the reference (ilupp/__init__.py:38-53): smaller ints are widened, larger
    ones are refused rather than silently truncated."""
    sz, target_sz = idx.dtype.itemsize, _index_size
    if sz == target_sz:
        return idx
    elif sz < target_sz:
        return idx.astype(_index_dtype)
    else:
        raise TypeError(
            'Index array has %d bytes per index, but the library '
            'is compiled for %d bytes per index. Downcasting might '
            'lead to integer overflow. Please compile ilupp with a '
            'larger index type if you need to use very large matrices.'
            % (sz, target_sz))


def _matrix_fields(A):
    """(data, indices, indptr, is_csr) of a square scipy CSR/CSC matrix; sorts A's indices IN PLACE
    exactly like the reference does (ilupp/__init__.py:55-71)."""
    if isinstance(A, scipy.sparse.csr_matrix):
        is_csr = True
    elif isinstance(A, scipy.sparse.csc_matrix):
        is_csr = False
    else:
        raise TypeError("A must be a csr_matrix or a csc_matrix")
    if A.shape[0] != A.shape[1]:
        raise ValueError("A must be a square matrix!")
    A.sort_indices()
    return A.data, _upcast_indices(A.indices), _upcast_indices(A.indptr), is_csr


def _matrix_from_info(data, indices, indptr, is_csr, rows, cols):
    """ilupp/__init__.py:73-82"""
    if is_csr:
        A = scipy.sparse.csr_matrix((data, indices, indptr), shape=(rows, cols), copy=False)
    else:
        A = scipy.sparse.csc_matrix((data, indices, indptr), shape=(rows, cols), copy=False)
    A.has_sorted_indices = True
    return A


class _BaseWrapper(scipy.sparse.linalg.LinearOperator):
    """Members common to all preconditioners (reference: ilupp/__init__.py:122-168)."""

    def _matvec(self, x):
        y = x.copy().ravel()
        self.pr.apply(y)
        return y

    def _rmatvec(self, x):
        y = x.copy().ravel()
        self.pr.apply_trans(y)
        return y

    def apply(self, x):
        """Apply the preconditioner to the vector `x` in-place."""
        self.pr.apply(x.ravel())

    def apply_trans(self, x):
        """Apply the transposed preconditioner to the vector `x` in-place."""
        self.pr.apply_trans(x.ravel())

    @property
    def total_nnz(self):
        """The total number of nonzeros stored in the factor matrices of the preconditioner."""
        return self.pr.total_nnz

    def factors(self):
        """Return all matrix factors (usually (L,U) or just (L,)) as a list of sparse matrices."""
        return [_matrix_from_info(*info) for info in self.pr.factors_info()]

    def __repr__(self):
        M, N = self.shape
        if self.dtype is None:
            dt = 'unspecified dtype'
        else:
            dt = 'dtype=' + str(self.dtype)
        return '<%dx%d %s with nnz=%d, %s>' % (M, N, self.__class__.__name__, self.total_nnz, dt)


class ILUTPreconditioner(_BaseWrapper):
    """ILUT (Saad) preconditioner: ``fill_in`` nonzeros per row of L/U, relative ``threshold``.
    Reference: ilupp/__init__.py:205-216."""

    def __init__(self, A, fill_in=100, threshold=0.1):
        Ad, Ai, Ap, Ao = _matrix_fields(A)
        self.pr = _ilupp.ILUTPreconditioner(Ad, Ai, Ap, Ao, fill_in, threshold)
        scipy.sparse.linalg.LinearOperator.__init__(self, shape=A.shape, dtype=A.dtype)


class ILU0Preconditioner(_BaseWrapper):
    """ILU(0) preconditioner (no fill-in).  Reference: ilupp/__init__.py:272-281."""

    def __init__(self, A):
        Ad, Ai, Ap, Ao = _matrix_fields(A)
        self.pr = _ilupp.ILU0Preconditioner(Ad, Ai, Ap, Ao)
        scipy.sparse.linalg.LinearOperator.__init__(self, shape=A.shape, dtype=A.dtype)


class IChol0Preconditioner(_BaseWrapper):
    """IChol(0) preconditioner for a symmetric positive definite matrix.  Reference: :283-293."""

    def __init__(self, A):
        Ad, Ai, Ap, Ao = _matrix_fields(A)
        self.pr = _ilupp.IChol0Preconditioner(Ad, Ai, Ap, Ao)
        scipy.sparse.linalg.LinearOperator.__init__(self, shape=A.shape, dtype=A.dtype)


class ICholTPreconditioner(_BaseWrapper):
    """Incomplete Cholesky with ``add_fill_in`` extra nonzeros per column and relative ``threshold``
    (Lin-More for threshold=0).  Reference: ilupp/__init__.py:295-310."""

    def __init__(self, A, add_fill_in=0, threshold=0.0):
        Ad, Ai, Ap, Ao = _matrix_fields(A)
        self.pr = _ilupp.ICholTPreconditioner(Ad, Ai, Ap, Ao, add_fill_in, threshold)
        scipy.sparse.linalg.LinearOperator.__init__(self, shape=A.shape, dtype=A.dtype)


def ichol0(A):
    """L factor of an incomplete Cholesky decomposition without fill-in (reference :314-316)."""
    return _matrix_from_info(*_ilupp.ichol0(*_matrix_fields(A)))


def icholt(A, add_fill_in=0, threshold=0.0):
    """L factor of an incomplete Cholesky decomposition with thresholding (reference :318-320)."""
    return _matrix_from_info(*_ilupp.icholt(*_matrix_fields(A), add_fill_in, threshold))


def ilu0(A):
    """(L, U) factors of an incomplete LU decomposition without fill-in (reference :322-324)."""
    return tuple(_matrix_from_info(*mtx) for mtx in _ilupp.ilu0(*_matrix_fields(A)))


def ilut(A, fill_in=100, threshold=0.1):
    """(L, U) factors of an incomplete LU decomposition with thresholding (reference :326-328)."""
    return tuple(_matrix_from_info(*mtx) for mtx in _ilupp.ilut(*_matrix_fields(A), fill_in, threshold))
