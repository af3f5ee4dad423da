"""ilupp_amd -- incomplete LU / incomplete Cholesky preconditioners on MI355X.

A drop-in for the hot path of c-f-h/ilupp behind that package's Python surface: the preconditioner classes are
scipy ``LinearOperator``s built from a square scipy CSR/CSC matrix, ``P @ x`` / ``P.T @ x`` / in-place
``apply`` / ``apply_trans`` run the triangular solves, ``factors()`` hands the factors back as scipy matrices.
What the reference does in C++ on one core (src/ilupp via src/binding.cpp) runs here as hand-written HIP kernels
through the C ABI of ``include/ilupp_hip.h``; there is no CPU fallback.

    import ilupp_amd as ilupp
    P = ilupp.ILU0Preconditioner(A)
    y = P @ x

Observable behaviour kept from the reference (ilupp/__init__.py): class and function names, argument names and
defaults, ``A.sort_indices()`` on the caller's matrix, exception types, ``total_nnz`` conventions, ``repr``.
"""
import collections

import numpy as np
import scipy.sparse as _sp
from scipy.sparse.linalg import LinearOperator as _LinearOperator

import os as _os

from . import _native
from .params import iluplusplus_precond_parameter, preprocessing_sequence  # noqa: F401  (reference: ilupp/__init__.py:27-28)

__version__ = "0.2.0"

# Which binding of the C ABI builds the objects: the ctypes one (`_native`, default: it also carries the device-pointer entry
# points) or the compiled pybind11 shim (`_ilupp_hip`, ilupp_amd/csrc/pybind_module.cpp: the module a maintainer of the reference
# would ship in place of `ilupp._ilupp`).  Same factory signatures, same object members.
if _os.environ.get("ILUPP_AMD_BINDING", "ctypes") == "pybind":
    from . import _ilupp_hip as _backend
else:
    _backend = _native

# index width of the compiled engine (the reference's `Integer`; binding.cpp:279)
_INDEX_DTYPES = {4: np.dtype(np.int32), 8: np.dtype(np.int64)}
try:
    _INDEX = _INDEX_DTYPES[_native.index_size()]
except KeyError:
    raise RuntimeError("invalid index type size %d" % _native.index_size())

_Borrowed = collections.namedtuple("_Borrowed", "data indices indptr is_csr")


def _engine_indices(a):
    """an index array in the engine's width: narrower ones are widened, wider ones refused (never truncated)"""
    if a.dtype.itemsize > _INDEX.itemsize:
        raise TypeError("the matrix uses %d-byte indices but this build of the engine works with %d-byte ones; "
                        "narrowing them could overflow" % (a.dtype.itemsize, _INDEX.itemsize))
    return a if a.dtype.itemsize == _INDEX.itemsize else a.astype(_INDEX)


def _borrow(A):
    """the buffers of a square CSR/CSC matrix as the engine wants them; sorts A's indices in place like the reference"""
    if isinstance(A, _sp.csr_matrix):
        row_major = True
    elif isinstance(A, _sp.csc_matrix):
        row_major = False
    else:
        raise TypeError("A must be a csr_matrix or a csc_matrix")
    rows, cols = A.shape
    if rows != cols:
        raise ValueError("A must be a square matrix!")
    A.sort_indices()
    return _Borrowed(A.data, _engine_indices(A.indices), _engine_indices(A.indptr), row_major)


def _as_scipy(info):
    """one entry of factors_info() -> scipy matrix sharing the arrays"""
    data, indices, indptr, row_major, rows, cols = info
    kind = _sp.csr_matrix if row_major else _sp.csc_matrix
    M = kind((data, indices, indptr), shape=(rows, cols), copy=False)
    M.has_sorted_indices = True
    return M


class _HipPreconditioner(_LinearOperator):
    """A factorisation held in HBM, seen as a linear operator.  `pr` is the native object (the role of the
    reference's pybind11 preconditioner objects)."""

    def __init__(self, A, make):
        self.pr = make(_borrow(A))
        super().__init__(dtype=A.dtype, shape=A.shape)

    # -- LinearOperator protocol: out-of-place, so that P @ x, P.dot(x), P.T @ x and scipy's solvers work
    def _solve_copy(self, x, solve):
        y = np.array(x, copy=True).ravel()
        solve(y)
        return y

    def _matvec(self, x):
        return self._solve_copy(x, self.pr.apply)

    def _rmatvec(self, x):
        return self._solve_copy(x, self.pr.apply_trans)

    # -- the reference's in-place members
    def apply(self, x):
        """x <- M^-1 x, overwriting the caller's array (any shape with n elements; it is flattened as a view)."""
        flat = x.ravel()
        self.pr.apply(flat)

    def apply_trans(self, x):
        """x <- M^-T x, overwriting the caller's array."""
        flat = x.ravel()
        self.pr.apply_trans(flat)

    total_nnz = property(lambda self: self.pr.total_nnz,
                         doc="Stored entries of all factors, counted the way the reference counts them for this kind of object.")

    def factors(self):
        """All matrix factors ((L, U) or just (L,)) as a list of sparse matrices."""
        return [_as_scipy(f) for f in self.pr.factors_info()]

    def __repr__(self):
        rows, cols = self.shape
        what = "unspecified dtype" if self.dtype is None else "dtype=%s" % (self.dtype,)
        return "<%dx%d %s with nnz=%d, %s>" % (rows, cols, type(self).__name__, self.total_nnz, what)


def _ml_parameters(threshold, fill_in, params):
    """the parameter object of a multilevel construction: the caller's, or default-constructed ones carrying the two numbers"""
    if params is not None:
        return params
    fresh = iluplusplus_precond_parameter()
    fresh.threshold = threshold
    if fill_in is not None:
        fresh.fill_in = fill_in
    return fresh


class ILUppPreconditioner(_HipPreconditioner):
    """Multilevel ILU++ (reference: ilupp/__init__.py:171-203 over binding.cpp:284-298).  With default-constructed parameters (the
    family WITH pivoting) ONE matrix builds about 2.5x slower than the reference does on one host core -- that factorisation is a
    sequential chain; use the family without pivoting (``default_configuration(1)``) or :meth:`batch` where construction time matters.

    `A`: scipy CSR or CSC matrix.  `threshold` / `fill_in`: the two numbers most callers tune (relative size below which an entry
    is dropped; bound on the entries kept per row).  `params`: a complete :class:`iluplusplus_precond_parameter`; when given, the two
    numbers are ignored.

    Default-constructed parameters select, as in the reference, the factorisation WITH pivoting (partialILUCDP: the column of a step is the
    largest entry of its working row, the next row the one with the fewest entries in L so far) -- a chain of n steps that one wave of the
    GPU walks (ilupp_amd/csrc/pilucdp.hip).  Parameters without row reordering, total pivoting and pivot tolerance
    (``params.default_configuration(1)``, precon_parameter 10) fix rows and columns beforehand and run as a dataflow computation over all
    CUs (piluc_df.hip): the fast path for large matrices.  Both are bit-identical to the reference; what is not built (the improved Schur complement, FINAL_ROW_CRIT < -1, a few preprocessing steps) raises
    NotImplementedError."""

    def __init__(self, A, threshold=1.0, fill_in=None, params=None):
        params = _ml_parameters(threshold, fill_in, params)
        super().__init__(A, lambda m: _backend.MultilevelILUCDPPreconditioner(*m, params))

    @classmethod
    def batch(cls, matrices, threshold=1.0, fill_in=None, params=None):
        """One preconditioner per matrix of `matrices` (all CSR or all CSC), built side by side on the GPU: the objects
        ``[ILUppPreconditioner(A, ...) for A in matrices]`` gives, bit for bit, in about the time of the slowest one -- the sequential
        chains of the factorisation with pivoting (one wave each) share one launch (``ilupp_hip_ml_create_batch``)."""
        params = _ml_parameters(threshold, fill_in, params)
        matrices = list(matrices)
        if not matrices:
            return []
        ms = [_borrow(A) for A in matrices]
        if any(m.is_csr != ms[0].is_csr for m in ms):
            raise TypeError("a batch holds matrices of one format")
        natives = _backend.MultilevelILUCDPPreconditioner_batch([(m.data, m.indices, m.indptr) for m in ms], ms[0].is_csr, params)
        out = []
        for A, pr in zip(matrices, natives):
            P = cls.__new__(cls)
            P.pr = pr
            _LinearOperator.__init__(P, dtype=A.dtype, shape=A.shape)
            out.append(P)
        return out

    # the three memory figures of the reference's object (binding.cpp:257-259), passed through from the native one
    memory = property(lambda self: self.pr.memory)
    memory_used_calculations = property(lambda self: self.pr.memory_used_calculations)
    memory_allocated_calculations = property(lambda self: self.pr.memory_allocated_calculations)


class ILUTPreconditioner(_HipPreconditioner):
    """ILUT (Saad): at most `fill_in` entries per row of L and of U, relative drop `threshold`."""

    def __init__(self, A, fill_in=100, threshold=0.1):
        super().__init__(A, lambda m: _backend.ILUTPreconditioner(*m, fill_in, threshold))


class ILUCPreconditioner(_HipPreconditioner):
    """ILUC, the Crout ILU of Li, Saad and Chow: at most `fill_in` entries per column of L and row of U, relative drop `threshold`."""

    def __init__(self, A, fill_in=100, threshold=0.1):
        super().__init__(A, lambda m: _backend.ILUCPreconditioner(*m, fill_in, threshold))


class ILUTPPreconditioner(_HipPreconditioner):
    """ILUT with column pivoting -- on this engine a SEQUENTIAL chain that one wave walks: bit-identical to the reference and
    10-30x slower than the reference on one host core for a single matrix (profiles/r04_chains.txt).  (Reference:
    ilupp/__init__.py:218-236 over ILUTP2, ILUTP.hpp:13-140.)

    `A`: scipy CSR or CSC matrix; `fill_in`: entries kept per row of L and of U; `threshold`: relative size below which an entry is
    dropped; `piv_tol`: 0 pivots only away from a zero, 1 always takes the largest entry of the row, values between compare the
    diagonal with the largest entry; `mem_factor`: storage reserved per factor, in multiples of nnz(A).

    Every row sees the column permutation the rows before it made (ilupp_amd/csrc/ilutp.hip)."""

    def __init__(self, A, fill_in=100, threshold=0.1, piv_tol=0.1, mem_factor=10.0):
        super().__init__(A, lambda m: _backend.ILUTPPreconditioner(*m, fill_in, threshold, piv_tol, -1, mem_factor))

    def permutations(self):
        """(left, right): the two index arrays the pivoting produced (rows stay where they are for this factorisation)."""
        left, right = self.pr.permutations()
        return left, right


class ILUCPPreconditioner(_HipPreconditioner):
    """Crout ILU with column pivoting (Mayer 2005) -- on this engine a SEQUENTIAL chain that one wave walks: bit-identical to the
    reference and 10-30x slower than the reference on one host core for a single matrix (profiles/r04_chains.txt).  (Reference:
    ilupp/__init__.py:252-270 over ILUCP4, ILUC.hpp:212-370.)

    `A`: scipy CSR or CSC matrix; `fill_in`: entries kept per column of L and per row of U; `threshold`, `piv_tol`, `mem_factor`: as
    for :class:`ILUTPPreconditioner`.

    The pivot of a step decides which entries of all later rows are alive (ilupp_amd/csrc/ilucp.hip)."""

    def __init__(self, A, fill_in=100, threshold=0.1, piv_tol=0.1, mem_factor=10.0):
        super().__init__(A, lambda m: _backend.ILUCPPreconditioner(*m, fill_in, threshold, piv_tol, -1, mem_factor))

    def permutations(self):
        """(left, right): the two index arrays the pivoting produced (rows stay where they are for this factorisation)."""
        left, right = self.pr.permutations()
        return left, right


class ILU0Preconditioner(_HipPreconditioner):
    """ILU(0): incomplete LU in the pattern of A."""

    def __init__(self, A):
        super().__init__(A, lambda m: _backend.ILU0Preconditioner(*m))


class IChol0Preconditioner(_HipPreconditioner):
    """IChol(0) of a symmetric positive definite matrix, in the pattern of its lower triangle."""

    def __init__(self, A):
        super().__init__(A, lambda m: _backend.IChol0Preconditioner(*m))


class ICholTPreconditioner(_HipPreconditioner):
    """Incomplete Cholesky with `add_fill_in` extra entries per column and relative drop `threshold`."""

    def __init__(self, A, add_fill_in=0, threshold=0.0):
        super().__init__(A, lambda m: _backend.ICholTPreconditioner(*m, add_fill_in, threshold))


def solve(A, b, rtol=1e-4, atol=1e-4, max_iter=500, threshold=0.1, fill_in=None, params=None, info=False):
    """Solve the linear system Ax=b using a multilevel ILU++ preconditioner and BiCGStab (reference: ilupp/__init__.py:85-119 over
    binding.cpp:199-231 -> solve_with_multilevel_preconditioner, BiCGstab with SPLIT preconditioning from the zero vector).

    The matrix, the preconditioner and every vector of the iteration live in HBM; only the residual norm comes back per iteration --
    the loop itself runs inside the library (``ilupp_hip_solve`` of the C ABI, ilupp_amd/csrc/krylov.hip).
    Returns the solution (with info=True also (iterations, relative reduction reached, residual norm reached)); raises
    RuntimeError("did not converge") like the reference.  As for :class:`ILUppPreconditioner`, default-constructed parameters select the
    factorisation with pivoting (sequential: one wave); parameters of the family without pivoting use the whole GPU."""
    params = _ml_parameters(threshold, fill_in, params)
    m = _borrow(A)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if b.shape[0] != A.shape[1]:
        raise RuntimeError("right-hand side has wrong size!")
    sol, it, rel, res = _backend.solve(*m, b, rtol, atol, max_iter, params)
    return (sol, (it, rel, res)) if info else sol


# ---- stand-alone factor functions ---------------------------------------------------------------
def ichol0(A):
    """L of an incomplete Cholesky decomposition without fill-in."""
    return _as_scipy(_backend.ichol0(*_borrow(A)))


def icholt(A, add_fill_in=0, threshold=0.0):
    """L of an incomplete Cholesky decomposition with thresholding."""
    return _as_scipy(_backend.icholt(*_borrow(A), add_fill_in, threshold))


def ilu0(A):
    """(L, U) of an incomplete LU decomposition without fill-in."""
    return tuple(_as_scipy(f) for f in _backend.ilu0(*_borrow(A)))


def ilut(A, fill_in=100, threshold=0.1):
    """(L, U) of an incomplete LU decomposition with thresholding."""
    return tuple(_as_scipy(f) for f in _backend.ilut(*_borrow(A), fill_in, threshold))


def iluc(A, fill_in=100, threshold=0.1):
    """(L, U) of an incomplete Crout LU decomposition with thresholding."""
    return tuple(_as_scipy(f) for f in _backend.iluc(*_borrow(A), fill_in, threshold))
