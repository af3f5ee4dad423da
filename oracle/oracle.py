"""oracle/oracle.py -- ctypes front-end to the parity oracle.  TEST INFRASTRUCTURE ONLY.

Loads ``oracle/liborc.so`` (plain-C restatement, prefix ``orc_``) and, when present,
``oracle/_ref/libilupp_ref.so`` (the real reference behind the same C ABI, prefix ``ref_``).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the product package ``ilupp_amd`` never does.

Every function takes/returns plain numpy arrays:  a matrix is ``(data, indices, indptr, is_csr)``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

LOWER, UPPER = 0, 1
ID, TRANSPOSE = 0, 1

OK, ERR_ZERO_PIVOT, ERR_NOT_TRIANGULAR, ERR_MEMORY = 0, 1, 2, 3


class _Mat(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("nnz", ctypes.c_int32),
                ("ptr", ctypes.POINTER(ctypes.c_int32)), ("idx", ctypes.POINTER(ctypes.c_int32)),
                ("val", ctypes.POINTER(ctypes.c_double)), ("is_csr", ctypes.c_int)]


_I32P = ctypes.POINTER(ctypes.c_int32)
_F64P = ctypes.POINTER(ctypes.c_double)

# steps of a preprocessing sequence (ilupp_oracle.h)
DROP_STANDARD, DROP_STANDARD2, DROP_ERR_PROP, DROP_ERR_PROP2, DROP_PIVOT, DROP_INVERSE = 1, 2, 4, 8, 16, 32
DROP_WEIGHTED, DROP_WEIGHTED2 = 64, 128
PRE_NORMALIZE_COLUMNS, PRE_NORMALIZE_ROWS, PRE_PQ_ORDERING, PRE_MAX_WEIGHTED_MATCHING_ORDERING, PRE_DD_SYMM_MOVE_CORNER_ORDERING_IM = 1, 2, 3, 4, 5
PRE_UNIT_OR_ZERO_DIAGONAL_SCALING, PRE_SPARSE_FIRST_ORDERING, PRE_SYMM_PQ = 6, 7, 8
ERR_UNSUPPORTED = 4


class MLParams(ctypes.Structure):
    """orc_ml_params: the knobs of the multilevel preconditioner without pivoting (precon_parameter 10 family)"""
    _fields_ = [("threshold", ctypes.c_double), ("n_preprocessing", ctypes.c_int), ("preprocessing", ctypes.c_int * 8),
                ("pq_threshold", ctypes.c_double), ("max_levels", ctypes.c_int), ("min_ml_size", ctypes.c_int32),
                ("small_pivot_terminates", ctypes.c_int), ("min_pivot", ctypes.c_double), ("min_elim_factor", ctypes.c_double),
                ("threshold_shift_schur", ctypes.c_double), ("vary_threshold_factor", ctypes.c_double),
                ("use_final_threshold", ctypes.c_int), ("final_threshold", ctypes.c_double), ("max_fill_in", ctypes.c_int32),
                ("drop_rules", ctypes.c_int), ("weight_standard_drop", ctypes.c_double), ("weight_standard_drop2", ctypes.c_double),
                ("weight_err_prop_drop", ctypes.c_double), ("weight_err_prop_drop2", ctypes.c_double), ("weight_pivot_drop", ctypes.c_double),
                ("combine_factor", ctypes.c_int), ("neutral_element", ctypes.c_double), ("min_weight", ctypes.c_double),
                ("scale_weight_invdiag", ctypes.c_int),
                ("piv_tol", ctypes.c_double), ("permute_rows", ctypes.c_int), ("total_piv", ctypes.c_int), ("begin_total_piv", ctypes.c_int),
                ("final_row_crit", ctypes.c_int), ("move_level_factor", ctypes.c_double), ("row_u_max", ctypes.c_double),
                ("weight_inverse_drop", ctypes.c_double), ("weight_weighted_drop", ctypes.c_double), ("init_weights_lu", ctypes.c_double)]


# the reference's default-constructed parameters (precon_parameter 0: the factorisation WITH pivoting), parameters_implementation.h:430-501
PIVOTING_DEFAULTS = dict(small_pivot_terminates=0, min_elim_factor=0.5, piv_tol=1.0, permute_rows=3, total_piv=1, begin_total_piv=1)


class _MLView(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("L", _Mat), ("U", _Mat), ("D", _F64P), ("perm_rows", _I32P), ("perm_cols", _I32P),
                ("inv_perm_rows", _I32P), ("inv_perm_cols", _I32P), ("D_l", _F64P), ("D_r", _F64P), ("zero_pivots", ctypes.c_int32)]


def ml_params(threshold=0.0, preprocessing=(PRE_NORMALIZE_COLUMNS, PRE_NORMALIZE_ROWS, PRE_PQ_ORDERING), **kw):
    """default_configuration(1) (NORMALIZE_COLUMNS, NORMALIZE_ROWS, PQ + precon_parameter 10) with single knobs changed"""
    p = MLParams()
    orc().lib.orc_ml_default_params(ctypes.byref(p))
    p.threshold = threshold
    p.n_preprocessing = len(preprocessing)
    for i, s in enumerate(preprocessing):
        p.preprocessing[i] = s
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class ML:
    """a multilevel preconditioner object of either library"""

    def __init__(self, lib, A, params):
        self.libobj = lib
        self.lib = lib.lib
        self.pre = lib.prefix
        args, self._keep = lib._in(A)
        self.h = ctypes.c_void_p()
        rc = getattr(self.lib, self.pre + "ml_create")(*args, ctypes.byref(params), ctypes.byref(self.h))
        if rc:
            raise OracleError(rc)
        self.n = args[0]

    def levels(self):
        return int(getattr(self.lib, self.pre + "ml_levels")(self.h))

    def total_nnz(self):
        return int(getattr(self.lib, self.pre + "ml_total_nnz")(self.h))

    def level(self, k):
        v = _MLView()
        rc = getattr(self.lib, self.pre + "ml_level")(self.h, int(k), ctypes.byref(v))
        if rc:
            raise OracleError(rc)
        n = v.n

        def mat(m):
            nnz = m.nnz
            return (np.ctypeslib.as_array(m.val, shape=(max(nnz, 1),))[:nnz].copy(), np.ctypeslib.as_array(m.idx, shape=(max(nnz, 1),))[:nnz].copy(),
                    np.ctypeslib.as_array(m.ptr, shape=(n + 1,)).copy(), bool(m.is_csr))

        def vec(p):
            return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()
        return {"n": n, "L": mat(v.L), "U": mat(v.U), "D": vec(v.D), "perm_rows": vec(v.perm_rows), "perm_cols": vec(v.perm_cols),
                "inv_perm_rows": vec(v.inv_perm_rows), "inv_perm_cols": vec(v.inv_perm_cols), "D_l": vec(v.D_l), "D_r": vec(v.D_r),
                "zero_pivots": int(v.zero_pivots)}

    def apply(self, x, use=ID):
        x = np.array(x, dtype=np.float64, copy=True).ravel()
        getattr(self.lib, self.pre + "ml_apply")(self.h, int(use), _p_f64(x))
        return x

    def __del__(self):
        if getattr(self, "h", None) is not None and self.h:
            getattr(self.lib, self.pre + "ml_free")(self.h)
            self.h = None


def _p_i32(a):
    return a.ctypes.data_as(_I32P)


def _p_f64(a):
    return a.ctypes.data_as(_F64P)


def build(ref=False):
    """(Re)build liborc.so, and _ref/libilupp_ref.so when ``ref`` and /root/reference exist."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])
    if ref and os.path.isdir("/root/reference/src/ilupp"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class OracleError(RuntimeError):
    def __init__(self, code, row=-1):
        self.code, self.row = code, row
        msg = {ERR_ZERO_PIVOT: "ILUT_heap: encountered zero pivot in row %d" % row,
               ERR_NOT_TRIANGULAR: "matrix not in triangular form",
               ERR_MEMORY: "append_row: insufficient memory reserved"}.get(code, "error %d" % code)
        super().__init__(msg)


class _Lib:
    """One of the two libraries behind the common ABI."""

    def __init__(self, path, prefix):
        self.path, self.prefix = path, prefix
        self.lib = ctypes.CDLL(path)
        f = self._f
        mat_in = [ctypes.c_int32, _I32P, _I32P, _F64P, ctypes.c_int]
        f("ilu0").argtypes = mat_in + [ctypes.POINTER(_Mat), ctypes.POINTER(_Mat)]
        f("ilut").argtypes = mat_in + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_Mat),
                                       ctypes.POINTER(_Mat), _I32P]
        f("iluc").argtypes = mat_in + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_Mat),
                                       ctypes.POINTER(_Mat), _I32P]
        f("ichol0").argtypes = mat_in + [ctypes.POINTER(_Mat)]
        f("icholt").argtypes = mat_in + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_Mat)]
        f("trisolve").argtypes = mat_in + [ctypes.c_int, ctypes.c_int, _F64P]
        f("trisolve").restype = None
        f("apply_lu").argtypes = [ctypes.POINTER(_Mat), ctypes.POINTER(_Mat), ctypes.c_int, _F64P]
        f("apply_lu").restype = None
        f("apply_llt").argtypes = [ctypes.POINTER(_Mat), ctypes.c_int, _F64P]
        f("apply_llt").restype = None
        f("free_mat").argtypes = [ctypes.POINTER(_Mat)]
        f("free_mat").restype = None
        f("sort_slots_by_abs_desc").argtypes = [_I32P, ctypes.c_int32, _F64P]
        f("sort_slots_by_abs_desc").restype = None
        f("ml_create").argtypes = mat_in + [ctypes.POINTER(MLParams), ctypes.POINTER(ctypes.c_void_p)]
        f("ml_levels").argtypes = [ctypes.c_void_p]
        f("ml_total_nnz").argtypes = [ctypes.c_void_p]
        f("ml_total_nnz").restype = ctypes.c_int32
        f("ml_level").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(_MLView)]
        f("ml_apply").argtypes = [ctypes.c_void_p, ctypes.c_int, _F64P]
        f("ml_apply").restype = None
        f("ml_free").argtypes = [ctypes.c_void_p]
        f("ml_free").restype = None

    def ml(self, A, params):
        """multilevel ILU++ without pivoting: an ML object (levels, total_nnz, level(k), apply)"""
        return ML(self, A, params)

    def solve(self, A, b, params, rtol=1e-4, atol=1e-4, max_iter=500):
        """_ilupp.solve as the reference's binding runs it (binding.cpp:200-230; the real reference only): (x, converged, iterations,
        10^-rel_tol, 10^-abs_tol)"""
        if self.prefix != "ref_":
            raise RuntimeError("solve: only the real reference has it (the engine's iteration is checked against its outputs)")
        f = self.lib.ref_solve
        f.restype = ctypes.c_int
        args, keep = self._in(A)
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(args[0])
        it, rel, res = ctypes.c_int32(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        ok = f(*args, ctypes.byref(params), _p_f64(b), ctypes.c_double(rtol), ctypes.c_double(atol), ctypes.c_int32(max_iter), _p_f64(x), ctypes.byref(it),
               ctypes.byref(rel), ctypes.byref(res))
        if ok < 0:
            raise OracleError(ERR_UNSUPPORTED)
        return x, bool(ok), it.value, rel.value, res.value

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    # -- helpers ----------------------------------------------------------------------------
    @staticmethod
    def _in(A):
        data, indices, indptr, is_csr = A
        data = np.ascontiguousarray(data, dtype=np.float64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        n = indptr.shape[0] - 1
        return (n, _p_i32(indptr), _p_i32(indices), _p_f64(data), int(bool(is_csr))), (data, indices, indptr)

    def _out(self, m):
        n, nnz = m.n, m.nnz
        ptr = np.ctypeslib.as_array(m.ptr, shape=(n + 1,)).copy()
        idx = np.ctypeslib.as_array(m.idx, shape=(max(nnz, 1),))[:nnz].copy()
        val = np.ctypeslib.as_array(m.val, shape=(max(nnz, 1),))[:nnz].copy()
        is_csr = bool(m.is_csr)
        self._f("free_mat")(ctypes.byref(m))
        return (val, idx, ptr, is_csr)

    @staticmethod
    def _as_mat(M, keep):
        data, indices, indptr, is_csr = M
        data = np.ascontiguousarray(data, dtype=np.float64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        keep.extend([data, indices, indptr])
        m = _Mat()
        m.n = indptr.shape[0] - 1
        m.nnz = int(indptr[-1])
        m.ptr, m.idx, m.val = _p_i32(indptr), _p_i32(indices), _p_f64(data)
        m.is_csr = int(bool(is_csr))
        return m

    # -- factorisations ---------------------------------------------------------------------
    def ilu0(self, A):
        args, keep = self._in(A)
        L, U = _Mat(), _Mat()
        rc = self._f("ilu0")(*args, ctypes.byref(L), ctypes.byref(U))
        if rc:
            raise OracleError(rc)
        return self._out(L), self._out(U)

    def ilut(self, A, fill_in=100, threshold=0.1):
        args, keep = self._in(A)
        L, U = _Mat(), _Mat()
        row = ctypes.c_int32(-1)
        rc = self._f("ilut")(*args, int(fill_in), float(threshold), ctypes.byref(L), ctypes.byref(U),
                             ctypes.byref(row))
        if rc:
            raise OracleError(rc, row.value)
        return self._out(L), self._out(U)

    def iluc(self, A, fill_in=100, threshold=0.1):
        """(first, second) as ilupp.iluc returns them: ILUC.hpp:112-207, binding.cpp:449-460"""
        args, keep = self._in(A)
        L, U = _Mat(), _Mat()
        row = ctypes.c_int32(-1)
        rc = self._f("iluc")(*args, int(fill_in), float(threshold), ctypes.byref(L), ctypes.byref(U),
                             ctypes.byref(row))
        if rc:
            raise OracleError(rc, row.value)
        return self._out(L), self._out(U)

    def ichol0(self, A):
        args, keep = self._in(A)
        L = _Mat()
        rc = self._f("ichol0")(*args, ctypes.byref(L))
        if rc:
            raise OracleError(rc)
        return self._out(L)

    def icholt(self, A, add_fill_in=0, threshold=0.0):
        args, keep = self._in(A)
        L = _Mat()
        rc = self._f("icholt")(*args, int(add_fill_in), float(threshold), ctypes.byref(L))
        if rc:
            raise OracleError(rc)
        return self._out(L)

    # -- solves -----------------------------------------------------------------------------
    def trisolve(self, M, form, use, x):
        args, keep = self._in(M)
        x = np.array(x, dtype=np.float64, copy=True).ravel()
        self._f("trisolve")(*args, int(form), int(use), _p_f64(x))
        return x

    def apply_lu(self, L, U, x, use=ID):
        keep = []
        Lm, Um = self._as_mat(L, keep), self._as_mat(U, keep)
        x = np.array(x, dtype=np.float64, copy=True).ravel()
        self._f("apply_lu")(ctypes.byref(Lm), ctypes.byref(Um), int(use), _p_f64(x))
        return x

    def apply_llt(self, L, x, use=ID):
        keep = []
        Lm = self._as_mat(L, keep)
        x = np.array(x, dtype=np.float64, copy=True).ravel()
        self._f("apply_llt")(ctypes.byref(Lm), int(use), _p_f64(x))
        return x

    def sort_slots_by_abs_desc(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.float64)
        lst = np.arange(keys.shape[0], dtype=np.int32)
        self._f("sort_slots_by_abs_desc")(_p_i32(lst), lst.shape[0], _p_f64(keys))
        return lst


_orc = None
_ref = None


def orc():
    """The plain-C restatement (always available; built on demand)."""
    global _orc
    if _orc is None:
        path = os.environ.get("ILUPP_ORACLE_LIBRARY") or os.path.join(_HERE, "liborc.so")       # (make -C oracle asan: the sanitizer build)
        if not os.path.exists(path):
            build()
        _orc = _Lib(path, "orc_")
    return _orc


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libilupp_ref.so"))


def ref():
    """The real reference behind the same ABI (only where oracle/_ref was built)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libilupp_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libilupp_ref.so not built (needs /root/reference: make -C oracle ref)")
        _ref = _Lib(path, "ref_")
        lib = _ref.lib
        mat_in = [ctypes.c_int32, _I32P, _I32P, _F64P, ctypes.c_int]
        lib.ref_total_nnz_lu_generic.argtypes = [ctypes.POINTER(_Mat), ctypes.POINTER(_Mat)]
        lib.ref_total_nnz_lu_generic.restype = ctypes.c_int32
        lib.ref_total_nnz_ilut.argtypes = mat_in + [ctypes.c_int32, ctypes.c_double]
        lib.ref_total_nnz_ilut.restype = ctypes.c_int32
        lib.ref_ilut_precond_apply.argtypes = mat_in + [ctypes.c_int32, ctypes.c_double, ctypes.c_int, _F64P]
    return _ref


def ref_total_nnz_ilut(A, fill_in, threshold):
    r = ref()
    args, keep = r._in(A)
    return int(r.lib.ref_total_nnz_ilut(*args, int(fill_in), float(threshold)))


def ref_ilut_precond_apply(A, fill_in, threshold, x, use=ID):
    r = ref()
    args, keep = r._in(A)
    x = np.array(x, dtype=np.float64, copy=True).ravel()
    rc = r.lib.ref_ilut_precond_apply(*args, int(fill_in), float(threshold), int(use), _p_f64(x))
    if rc:
        raise OracleError(rc)
    return x


def as_scipy(M):
    """(data, indices, indptr, is_csr) -> scipy matrix (test convenience)."""
    import scipy.sparse as sp
    data, indices, indptr, is_csr = M
    n = indptr.shape[0] - 1
    cls = sp.csr_matrix if is_csr else sp.csc_matrix
    return cls((data, indices, indptr), shape=(n, n))


def from_scipy(A):
    """scipy csr/csc -> (data, indices, indptr, is_csr) with sorted int32 indices."""
    import scipy.sparse as sp
    A = A.copy()
    A.sort_indices()
    return (A.data.astype(np.float64), A.indices.astype(np.int32), A.indptr.astype(np.int32),
            isinstance(A, sp.csr_matrix))


class ILUCP:
    """ILUCPPreconditioner of either library: factors for the major-order view, the permutation, apply"""
    _name = "ilucp"

    def __init__(self, lib, A, fill_in=100, threshold=0.1, piv_tol=0.1, rp=-1, mem_factor=10.0):
        self.lib = lib
        args, self._keep = lib._in(A)
        n = args[0]
        self.n, self.is_csr = n, bool(args[4])
        L, U = _Mat(), _Mat()
        perm = np.zeros(n, dtype=np.int32)
        zp = ctypes.c_int32(0)
        self.h = ctypes.c_void_p()
        if lib.prefix == "ref_":
            f = getattr(lib.lib, "ref_" + self._name)
            f.restype = ctypes.c_int
            rc = f(args[0], args[1], args[2], args[3], args[4], ctypes.c_int32(fill_in), ctypes.c_double(threshold), ctypes.c_double(piv_tol), ctypes.c_int32(rp),
                   ctypes.c_double(mem_factor), ctypes.byref(L), ctypes.byref(U), _p_i32(perm), ctypes.byref(zp), ctypes.byref(self.h))
        else:
            f = getattr(lib.lib, "orc_" + self._name)
            f.restype = ctypes.c_int
            rc = f(args[0], args[1], args[2], args[3], ctypes.c_int32(fill_in), ctypes.c_double(threshold), ctypes.c_double(piv_tol), ctypes.c_int32(rp),
                   ctypes.c_double(mem_factor), ctypes.byref(L), ctypes.byref(U), _p_i32(perm), ctypes.byref(zp))
        if rc:
            raise OracleError(rc)
        self.perm, self.zero_pivots = perm, int(zp.value)
        self._L, self._U = L, U
        self.L = self._copy(L)
        self.U = self._copy(U)

    @staticmethod
    def _copy(m):
        n, nnz = m.n, m.nnz
        return (np.ctypeslib.as_array(m.val, shape=(max(nnz, 1),))[:nnz].copy(), np.ctypeslib.as_array(m.idx, shape=(max(nnz, 1),))[:nnz].copy(),
                np.ctypeslib.as_array(m.ptr, shape=(n + 1,)).copy())

    def apply(self, x, use=ID):
        x = np.array(x, dtype=np.float64, copy=True).ravel()
        if self.lib.prefix == "ref_":
            getattr(self.lib.lib, "ref_%s_apply" % self._name)(self.h, ctypes.c_int32(self.n), int(use), _p_f64(x))
        else:
            getattr(self.lib.lib, "orc_apply_" + self._name)(ctypes.byref(self._L), ctypes.byref(self._U), _p_i32(self.perm), int(self.is_csr), int(use), _p_f64(x))
        return x


class ILUTP(ILUCP):
    """ILUTPPreconditioner of either library: L by rows (its 1 last, permuted column numbering), U by rows (pivot first, original
    column indices), the permutation, apply -- for the major-order view of the input"""
    _name = "ilutp"
