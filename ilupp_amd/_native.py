"""ctypes binding of include/ilupp_hip.h (the C ABI of the MI355X engine).

This module plays the role of the reference's compiled extension ``ilupp._ilupp``
(src/binding.cpp): same function names, positional signatures, return kinds and exception
types/messages for the hot-path subset.  There is NO CPU fallback: if the HIP library is missing
or no GPU is visible, construction raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ILUPP_HIP_LIBRARY") or os.path.join(_HERE, "libilupp_hip.so")

_I32P = ctypes.POINTER(ctypes.c_int32)
_F64P = ctypes.POINTER(ctypes.c_double)
_VP = ctypes.c_void_p


class Timings(ctypes.Structure):
    _fields_ = [("analysis_ms", ctypes.c_float), ("numeric_ms", ctypes.c_float),
                ("last_apply_ms", ctypes.c_float), ("numeric_kernel_ms", ctypes.c_float),
                ("lsolve_kernel_ms", ctypes.c_float), ("usolve_kernel_ms", ctypes.c_float)]


class MLParams(ctypes.Structure):
    """ilupp_ml_params of include/ilupp_hip.h"""
    _fields_ = [("threshold", ctypes.c_double), ("n_preprocessing", ctypes.c_int32), ("preprocessing", ctypes.c_int32 * 8),
                ("pq_threshold", ctypes.c_double), ("max_levels", ctypes.c_int32), ("min_ml_size", ctypes.c_int32),
                ("small_pivot_terminates", ctypes.c_int32), ("min_pivot", ctypes.c_double), ("min_elim_factor", ctypes.c_double),
                ("threshold_shift_schur", ctypes.c_double), ("vary_threshold_factor", ctypes.c_double),
                ("use_final_threshold", ctypes.c_int32), ("final_threshold", ctypes.c_double), ("max_fill_in", ctypes.c_int32),
                ("drop_rules", ctypes.c_int32), ("weight_standard_drop", ctypes.c_double), ("weight_standard_drop2", ctypes.c_double),
                ("weight_err_prop_drop", ctypes.c_double), ("weight_err_prop_drop2", ctypes.c_double), ("weight_pivot_drop", ctypes.c_double),
                ("combine_factor", ctypes.c_int32), ("neutral_element", ctypes.c_double), ("min_weight", ctypes.c_double),
                ("scale_weight_invdiag", ctypes.c_int32),
                ("piv_tol", ctypes.c_double), ("permute_rows", ctypes.c_int32), ("total_piv", ctypes.c_int32), ("begin_total_piv", ctypes.c_int32),
                ("final_row_crit", ctypes.c_int32), ("move_level_factor", ctypes.c_double), ("row_u_max", ctypes.c_double),
                ("weight_inverse_drop", ctypes.c_double), ("weight_weighted_drop", ctypes.c_double), ("init_weights_lu", ctypes.c_double)]


_lib = None
_lib_gil = None


def _lib_holding_gil():
    """the same library through ctypes.PyDLL: calls made through it keep the GIL, as the reference's `apply` / `apply_trans` do
    (binding.cpp:237-254 has no gil_scoped_release; the factorisations release it, :299-397)"""
    global _lib_gil
    if _lib_gil is None:
        lib()
        G = ctypes.PyDLL(_LIB_PATH)
        G.ilupp_hip_apply.argtypes = [_VP, _VP, ctypes.c_int64]
        G.ilupp_hip_apply_trans.argtypes = [_VP, _VP, ctypes.c_int64]
        G.ilupp_hip_ml_apply.argtypes = [_VP, _VP, ctypes.c_int64, ctypes.c_int]
        G.ilupp_hip_ilucp_apply.argtypes = [_VP, _VP, ctypes.c_int64, ctypes.c_int]
        _lib_gil = G
    return _lib_gil


def lib():
    """Load libilupp_hip.so (built in-tree by `make -C ilupp_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError("ilupp_amd: %s not found -- build the HIP extension first "
                          "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
                          % _LIB_PATH)
    L = ctypes.CDLL(_LIB_PATH)
    L.ilupp_hip_index_size.restype = ctypes.c_int
    L.ilupp_hip_last_error.restype = ctypes.c_char_p
    L.ilupp_hip_set_device.argtypes = [ctypes.c_int]
    L.ilupp_hip_device_count.restype = ctypes.c_int
    mat_host = [_VP, _VP, _VP, ctypes.c_int32, ctypes.c_int]
    L.ilupp_hip_ilu0_create.argtypes = mat_host + [ctypes.POINTER(_VP)]
    L.ilupp_hip_ilu0_create_device.argtypes = mat_host + [ctypes.POINTER(_VP)]
    L.ilupp_hip_ilu0_create_device_nnz.argtypes = [_VP, _VP, _VP, ctypes.c_int32, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(_VP)]
    L.ilupp_hip_ilu0_refactor_device.argtypes = [_VP, _VP, _VP, _VP]
    L.ilupp_hip_ilut_create.argtypes = mat_host + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_VP)]
    L.ilupp_hip_iluc_create.argtypes = mat_host + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_VP)]
    L.ilupp_hip_iluc_create_device.argtypes = mat_host + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_VP)]
    L.ilupp_hip_ichol0_create.argtypes = mat_host + [ctypes.POINTER(_VP)]
    L.ilupp_hip_icholt_create.argtypes = mat_host + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_VP)]
    L.ilupp_hip_ilut_create_device.argtypes = mat_host + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_VP)]
    L.ilupp_hip_ichol0_create_device.argtypes = mat_host + [ctypes.POINTER(_VP)]
    L.ilupp_hip_icholt_create_device.argtypes = mat_host + [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(_VP)]
    L.ilupp_hip_set_caller_stream.argtypes = [_VP, ctypes.c_int]
    L.ilupp_hip_spmv_device.argtypes = [_VP, _VP, _VP, ctypes.c_int32, ctypes.c_int64, _VP, _VP, _VP]
    L.ilupp_hip_path.argtypes = [_VP]
    L.ilupp_hip_path.restype = ctypes.c_char_p
    L.ilupp_hip_analysis_path.argtypes = [_VP]
    L.ilupp_hip_analysis_path.restype = ctypes.c_char_p
    L.ilupp_hip_debug_static_table.argtypes = [_VP, ctypes.c_int, _VP, ctypes.c_longlong]
    L.ilupp_hip_debug_static_table.restype = ctypes.c_longlong
    L.ilupp_hip_kernel_names.argtypes = [_VP]
    L.ilupp_hip_kernel_names.restype = ctypes.c_char_p
    L.ilupp_hip_destroy.argtypes = [_VP]
    L.ilupp_hip_destroy.restype = None
    L.ilupp_hip_apply.argtypes = [_VP, _VP, ctypes.c_int64]
    L.ilupp_hip_apply_trans.argtypes = [_VP, _VP, ctypes.c_int64]
    L.ilupp_hip_apply_device.argtypes = [_VP, _VP, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    L.ilupp_hip_sync.argtypes = [_VP]
    L.ilupp_hip_release_cached_memory.argtypes = []
    L.ilupp_hip_set_cache_limit.argtypes = [ctypes.c_ulonglong]
    L.ilupp_hip_cached_bytes.argtypes = []
    L.ilupp_hip_cached_bytes.restype = ctypes.c_ulonglong
    L.ilupp_hip_live_blocks.argtypes = []
    L.ilupp_hip_live_blocks.restype = ctypes.c_ulonglong
    L.ilupp_hip_total_nnz.argtypes = [_VP]
    L.ilupp_hip_total_nnz.restype = ctypes.c_int64
    for name in ("memory_used_calculations", "memory_allocated_calculations", "memory"):
        f = getattr(L, "ilupp_hip_" + name)
        f.argtypes = [_VP]
        f.restype = ctypes.c_double
    L.ilupp_hip_exists.argtypes = [_VP]
    L.ilupp_hip_special_info.argtypes = [_VP]
    L.ilupp_hip_special_info.restype = ctypes.c_char_p
    L.ilupp_hip_print_info.argtypes = [_VP]
    L.ilupp_hip_print_info.restype = None
    L.ilupp_hip_dimension.argtypes = [_VP]
    L.ilupp_hip_dimension.restype = ctypes.c_int32
    L.ilupp_hip_num_factors.argtypes = [_VP]
    L.ilupp_hip_factor_info.argtypes = [_VP, ctypes.c_int, _I32P, _I32P, ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_int)]
    L.ilupp_hip_factor_copy.argtypes = [_VP, ctypes.c_int, _VP, _VP, _VP]
    L.ilupp_hip_factor_device_ptrs.argtypes = [_VP, ctypes.c_int, ctypes.POINTER(_VP), ctypes.POINTER(_VP),
                                               ctypes.POINTER(_VP)]
    L.ilupp_hip_get_timings.argtypes = [_VP, ctypes.POINTER(Timings)]
    L.ilupp_hip_ml_default_params.argtypes = [ctypes.POINTER(MLParams)]
    L.ilupp_hip_ml_default_params.restype = None
    L.ilupp_hip_ml_create.argtypes = mat_host + [ctypes.POINTER(MLParams), ctypes.POINTER(_VP)]
    L.ilupp_hip_ml_create_device.argtypes = mat_host + [ctypes.POINTER(MLParams), ctypes.POINTER(_VP)]
    L.ilupp_hip_ml_create_batch.argtypes = [ctypes.c_int32, ctypes.POINTER(_VP), ctypes.POINTER(_VP), ctypes.POINTER(_VP), _I32P, ctypes.c_int,
                                            ctypes.POINTER(MLParams), ctypes.POINTER(_VP), _I32P]
    L.ilupp_hip_ml_destroy.argtypes = [_VP]
    L.ilupp_hip_ml_destroy.restype = None
    L.ilupp_hip_ml_apply.argtypes = [_VP, _VP, ctypes.c_int64, ctypes.c_int]
    L.ilupp_hip_ml_apply_device.argtypes = [_VP, _VP, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    L.ilupp_hip_ml_apply_part_device.argtypes = [_VP, _VP, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ilupp_hip_ml_sync.argtypes = [_VP]
    L.ilupp_hip_ilucp_create.argtypes = [_VP, _VP, _VP, ctypes.c_int32, ctypes.c_int, ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_int32,
                                         ctypes.c_double, ctypes.POINTER(_VP)]
    L.ilupp_hip_ilutp_create.argtypes = L.ilupp_hip_ilucp_create.argtypes
    L.ilupp_hip_ilucp_destroy.argtypes = [_VP]
    L.ilupp_hip_ilucp_destroy.restype = None
    L.ilupp_hip_ilucp_apply.argtypes = [_VP, _VP, ctypes.c_int64, ctypes.c_int]
    L.ilupp_hip_ilucp_total_nnz.argtypes = [_VP]
    L.ilupp_hip_ilucp_total_nnz.restype = ctypes.c_int64
    L.ilupp_hip_ilucp_zero_pivots.argtypes = [_VP]
    L.ilupp_hip_ilucp_info.argtypes = [_VP, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_float)]
    L.ilupp_hip_ilucp_copy.argtypes = [_VP] * 8
    L.ilupp_hip_ml_levels.argtypes = [_VP]
    L.ilupp_hip_ml_levels.restype = ctypes.c_int32
    L.ilupp_hip_ml_total_nnz.argtypes = [_VP]
    L.ilupp_hip_ml_total_nnz.restype = ctypes.c_int64
    L.ilupp_hip_ml_level_info.argtypes = [_VP, ctypes.c_int32, _I32P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    L.ilupp_hip_ml_level_copy.argtypes = [_VP, ctypes.c_int32] + [_VP] * 13
    L.ilupp_hip_ml_timings.argtypes = [_VP, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.ilupp_hip_solve.argtypes = mat_host + [_VP, ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.POINTER(MLParams), _VP,
                                             ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    _lib = L
    return L


# names every symbol include/ilupp_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "ilupp_hip_index_size", "ilupp_hip_last_error", "ilupp_hip_set_device", "ilupp_hip_device_count",
    "ilupp_hip_ilu0_create", "ilupp_hip_ilu0_create_device", "ilupp_hip_ilu0_create_device_nnz", "ilupp_hip_ilut_create",
    "ilupp_hip_ichol0_create", "ilupp_hip_icholt_create", "ilupp_hip_destroy",
    "ilupp_hip_apply", "ilupp_hip_apply_trans", "ilupp_hip_apply_device", "ilupp_hip_total_nnz",
    "ilupp_hip_memory_used_calculations", "ilupp_hip_memory_allocated_calculations", "ilupp_hip_memory",
    "ilupp_hip_exists", "ilupp_hip_special_info", "ilupp_hip_print_info", "ilupp_hip_dimension",
    "ilupp_hip_num_factors", "ilupp_hip_factor_info", "ilupp_hip_factor_copy",
    "ilupp_hip_factor_device_ptrs", "ilupp_hip_get_timings", "ilupp_hip_ilu0_refactor_device",
    "ilupp_hip_sync", "ilupp_hip_release_cached_memory", "ilupp_hip_set_cache_limit", "ilupp_hip_cached_bytes", "ilupp_hip_live_blocks", "ilupp_hip_ilut_create_device", "ilupp_hip_ichol0_create_device",
    "ilupp_hip_icholt_create_device", "ilupp_hip_set_caller_stream", "ilupp_hip_path", "ilupp_hip_analysis_path", "ilupp_hip_debug_static_table", "ilupp_hip_kernel_names", "ilupp_hip_spmv_device",
    "ilupp_hip_iluc_create", "ilupp_hip_iluc_create_device",
    "ilupp_hip_ml_default_params", "ilupp_hip_ml_create", "ilupp_hip_ml_create_device", "ilupp_hip_ml_destroy", "ilupp_hip_ml_apply",
    "ilupp_hip_ml_apply_device", "ilupp_hip_ml_apply_part_device", "ilupp_hip_ml_sync", "ilupp_hip_ml_levels", "ilupp_hip_ml_total_nnz", "ilupp_hip_ml_level_info",
    "ilupp_hip_ml_level_copy", "ilupp_hip_ml_timings", "ilupp_hip_solve", "ilupp_hip_ml_create_batch",
    "ilupp_hip_ilucp_create", "ilupp_hip_ilucp_destroy", "ilupp_hip_ilucp_apply", "ilupp_hip_ilucp_total_nnz", "ilupp_hip_ilucp_zero_pivots",
    "ilupp_hip_ilucp_info", "ilupp_hip_ilucp_copy", "ilupp_hip_ilutp_create",
]


def index_size():
    """binding.cpp:279"""
    return lib().ilupp_hip_index_size()


ILUPP_ERR_UNSUPPORTED = -8          # include/ilupp_hip.h


def _raise(rc):
    msg = lib().ilupp_hip_last_error().decode()
    if rc == ILUPP_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)          # pybind11 maps std::exception -> RuntimeError


# ---- buffer validation: the checks of make_matrix / make_vector (binding.cpp:33-98) ----------------
def _check_1d_contiguous(a, name):
    mv = memoryview(a)
    if mv.ndim != 1:
        raise RuntimeError("Expected 1D array for %s!" % name)
    if mv.strides[0] != mv.itemsize:
        raise RuntimeError("Expected contiguous array for %s!" % name)
    return mv


def _check_real(a, name):
    mv = _check_1d_contiguous(a, name)
    if mv.format != "d":
        raise RuntimeError("Expected d (d) array for %s, got %s!" % (name, mv.format))
    return mv


def _check_int(a, name):
    mv = _check_1d_contiguous(a, name)
    if not (len(mv.format) == 1 and mv.format in "ilq" and mv.itemsize == 4):
        raise RuntimeError("Expected integer type with length 4 for %s, got %s!" % (name, mv.format))
    return mv


def _matrix_args(A_data, A_indices, A_indptr, is_csr):
    d = _check_real(A_data, "A_data")
    i = _check_int(A_indices, "A_indices")
    p = _check_int(A_indptr, "A_indptr")
    if p.shape[0] <= 1:
        raise RuntimeError("matrix has size 0!")
    if i.shape[0] != d.shape[0]:
        raise RuntimeError("indices and data should have the same size!")
    n = p.shape[0] - 1
    da = np.frombuffer(d, dtype=np.float64) if d.shape[0] else np.zeros(0)
    ia = np.frombuffer(i, dtype=np.int32) if i.shape[0] else np.zeros(0, dtype=np.int32)
    pa = np.frombuffer(p, dtype=np.int32)
    return (da.ctypes.data, ia.ctypes.data, pa.ctypes.data, n, 1 if is_csr else 0), (da, ia, pa)


class Preconditioner:
    """Native preconditioner object: the members of wrapPreconditioner<P> (binding.cpp:233-264)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.ilupp_hip_destroy(h)

    def _vec(self, x):
        mv = _check_real(x, "b")
        if mv.readonly:
            raise RuntimeError("b must be writable")
        return np.frombuffer(mv, dtype=np.float64)

    def apply(self, x):
        a = self._vec(x)
        if a.shape[0] != lib().ilupp_hip_dimension(self._h):
            raise RuntimeError("vector has wrong size for preconditioner!")
        rc = _lib_holding_gil().ilupp_hip_apply(self._h, a.ctypes.data, a.shape[0])
        if rc:
            _raise(rc)

    def apply_trans(self, x):
        a = self._vec(x)
        if a.shape[0] != lib().ilupp_hip_dimension(self._h):
            raise RuntimeError("vector has wrong size for preconditioner!")
        rc = _lib_holding_gil().ilupp_hip_apply_trans(self._h, a.ctypes.data, a.shape[0])
        if rc:
            _raise(rc)

    def apply_device(self, dptr, n, transpose=False, sync=True):
        """apply on a device pointer (int address in this GPU's HBM)"""
        rc = lib().ilupp_hip_apply_device(self._h, dptr, n, 1 if transpose else 0, 1 if sync else 0)
        if rc:
            _raise(rc)

    def sync(self):
        rc = lib().ilupp_hip_sync(self._h)
        if rc:
            _raise(rc)

    def refactor_device(self, d_data, d_indices, d_indptr):
        rc = lib().ilupp_hip_ilu0_refactor_device(self._h, d_data, d_indices, d_indptr)
        if rc:
            _raise(rc)

    @property
    def total_nnz(self):
        return int(lib().ilupp_hip_total_nnz(self._h))

    @property
    def memory_used_calculations(self):
        return lib().ilupp_hip_memory_used_calculations(self._h)

    @property
    def memory_allocated_calculations(self):
        return lib().ilupp_hip_memory_allocated_calculations(self._h)

    @property
    def memory(self):
        return lib().ilupp_hip_memory(self._h)

    @property
    def exists(self):
        return bool(lib().ilupp_hip_exists(self._h))

    @property
    def special_info(self):
        return lib().ilupp_hip_special_info(self._h).decode()

    def print_info(self):
        lib().ilupp_hip_print_info(self._h)

    def factors_info(self):
        """list of (data, indices, indptr, is_csr, rows, cols), as wrap_matrix builds (binding.cpp:118-133)"""
        out = []
        L = lib()
        for k in range(L.ilupp_hip_num_factors(self._h)):
            rows, cols = ctypes.c_int32(), ctypes.c_int32()
            nnz, is_csr = ctypes.c_int64(), ctypes.c_int()
            rc = L.ilupp_hip_factor_info(self._h, k, ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(nnz),
                                         ctypes.byref(is_csr))
            if rc:
                _raise(rc)
            data = np.empty(nnz.value, dtype=np.float64)
            indices = np.empty(nnz.value, dtype=np.int32)
            indptr = np.empty(rows.value + 1, dtype=np.int32)
            rc = L.ilupp_hip_factor_copy(self._h, k, data.ctypes.data, indices.ctypes.data, indptr.ctypes.data)
            if rc:
                _raise(rc)
            out.append((data, indices, indptr, bool(is_csr.value), rows.value, cols.value))
        return out

    def path(self):
        """which kernel family built this object (measurement hook)"""
        return lib().ilupp_hip_path(self._h).decode()

    def analysis_path(self):
        """ILU(0): "grid" when the row blocks came from the box-grid guess (proven row by row), else "general" (measurement hook)"""
        return lib().ilupp_hip_analysis_path(self._h).decode()

    def kernel_names(self):
        """(factor kernel, forward sweep, backward sweep) of a static ILU(0) object or of an LL^T object (its sweeps once an apply has built them), else () (measurement hook)"""
        s = lib().ilupp_hip_kernel_names(self._h).decode()
        return tuple(s.split(";")) if s else ()

    def factor_device_ptrs(self, which):
        """(data, indices, indptr) device addresses of factor `which` (valid while the object lives)"""
        d, i, p = _VP(), _VP(), _VP()
        rc = lib().ilupp_hip_factor_device_ptrs(self._h, which, ctypes.byref(d), ctypes.byref(i), ctypes.byref(p))
        if rc:
            _raise(rc)
        return d.value, i.value, p.value

    def timings(self):
        t = Timings()
        lib().ilupp_hip_get_timings(self._h, ctypes.byref(t))
        return {k: getattr(t, k) for k, _ in Timings._fields_}


class MultilevelPreconditioner:
    """Native multilevel ILU++ object: the members wrapPreconditioner<_MultilevelILUCDPPreconditioner> gives it (binding.cpp:233-264,
    :284-298), plus the levels for tests and measurements."""

    def __init__(self, handle, n):
        self._h = handle
        self._n = n

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.ilupp_hip_ml_destroy(h)

    def _apply(self, x, transpose):
        mv = _check_real(x, "b")
        if mv.readonly:
            raise RuntimeError("b must be writable")
        a = np.frombuffer(mv, dtype=np.float64)
        if a.shape[0] != self._n:
            raise RuntimeError("vector has wrong size for preconditioner!")
        rc = _lib_holding_gil().ilupp_hip_ml_apply(self._h, a.ctypes.data, a.shape[0], transpose)
        if rc:
            _raise(rc)

    def apply(self, x):
        self._apply(x, 0)

    def apply_trans(self, x):
        self._apply(x, 1)

    def apply_device(self, dptr, n, transpose=False, sync=True):
        rc = lib().ilupp_hip_ml_apply_device(self._h, dptr, n, 1 if transpose else 0, 1 if sync else 0)
        if rc:
            _raise(rc)

    def apply_part_device(self, dptr, n, left, transpose=False, sync=True):
        """the left (left=True) or right half of the split preconditioner on a device pointer"""
        rc = lib().ilupp_hip_ml_apply_part_device(self._h, dptr, n, 1 if transpose else 0, 1 if left else 0, 1 if sync else 0)
        if rc:
            _raise(rc)

    def sync(self):
        rc = lib().ilupp_hip_ml_sync(self._h)
        if rc:
            _raise(rc)

    @property
    def total_nnz(self):
        return int(lib().ilupp_hip_ml_total_nnz(self._h))

    # (preconditioner.h:65, :113: bookkeeping of host allocations in the reference; nothing of the kind lives on the host here)
    memory_used_calculations = 0.0
    memory_allocated_calculations = 0.0
    memory = 0.0
    exists = True
    special_info = ""

    def factors_info(self):
        """binding.cpp:158-163: "not implemented" for the multilevel class -- an empty list"""
        return []

    def print_info(self):
        print("A multilevel incomplete LU factorisation: %d levels, %d entries" % (self.levels(), self.total_nnz))

    def levels(self):
        return int(lib().ilupp_hip_ml_levels(self._h))

    def level_sizes(self, k):
        """(n, stored entries of the left factor, of the right factor) of level k"""
        n, nl, nu = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        rc = lib().ilupp_hip_ml_level_info(self._h, k, ctypes.byref(n), ctypes.byref(nl), ctypes.byref(nu))
        if rc:
            _raise(rc)
        return n.value, nl.value, nu.value

    def level(self, k):
        """level k as a dict: L (by columns), U (by rows) as (data, indices, indptr), D, the permutations, the scalings"""
        L = lib()
        n, nl, nu = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        rc = L.ilupp_hip_ml_level_info(self._h, k, ctypes.byref(n), ctypes.byref(nl), ctypes.byref(nu))
        if rc:
            _raise(rc)
        n, nl, nu = n.value, nl.value, nu.value
        out = {"n": n,
               "L": (np.empty(nl), np.empty(nl, dtype=np.int32), np.empty(n + 1, dtype=np.int32)),
               "U": (np.empty(nu), np.empty(nu, dtype=np.int32), np.empty(n + 1, dtype=np.int32)),
               "D": np.empty(n), "perm_rows": np.empty(n, dtype=np.int32), "perm_cols": np.empty(n, dtype=np.int32),
               "inv_perm_rows": np.empty(n, dtype=np.int32), "inv_perm_cols": np.empty(n, dtype=np.int32), "D_l": np.empty(n), "D_r": np.empty(n)}
        ptrs = [a.ctypes.data for a in out["L"]] + [a.ctypes.data for a in out["U"]] + [
            out[k2].ctypes.data for k2 in ("D", "perm_rows", "perm_cols", "inv_perm_rows", "inv_perm_cols", "D_l", "D_r")]
        rc = L.ilupp_hip_ml_level_copy(self._h, k, *ptrs)
        if rc:
            _raise(rc)
        return out

    def timings(self):
        a, b, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        lib().ilupp_hip_ml_timings(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return {"construct_ms": a.value, "kernel_ms": b.value, "last_apply_ms": c.value}


def MultilevelILUCDPPreconditioner(A_data, A_indices, A_indptr, is_csr, param):
    """binding.cpp:284-298; `param`: an iluplusplus_precond_parameter (ilupp_amd.params) or an MLParams block"""
    p = param if isinstance(param, MLParams) else param._to_ml_params()
    args, keep = _matrix_args(A_data, A_indices, A_indptr, is_csr)
    h = _VP()
    rc = lib().ilupp_hip_ml_create(*args, ctypes.byref(p), ctypes.byref(h))
    if rc:
        _raise(rc)
    return MultilevelPreconditioner(h, args[3])


def MultilevelILUCDPPreconditioner_batch(matrices, is_csr, param):
    """one multilevel preconditioner per matrix of `matrices` (a list of (data, indices, indptr)), built side by side
    (ilupp_hip_ml_create_batch): the same objects the constructor gives one at a time"""
    p = param if isinstance(param, MLParams) else param._to_ml_params()
    cnt = len(matrices)
    if cnt == 0:
        return []
    keep, ns = [], []
    D, I, P = (_VP * cnt)(), (_VP * cnt)(), (_VP * cnt)()
    for k, (d, i, ptr) in enumerate(matrices):
        args, arrays = _matrix_args(d, i, ptr, is_csr)
        keep.append(arrays)
        D[k], I[k], P[k] = args[0], args[1], args[2]
        ns.append(args[3])
    N = (ctypes.c_int32 * cnt)(*ns)
    out = (_VP * cnt)()
    status = (ctypes.c_int32 * cnt)()
    rc = lib().ilupp_hip_ml_create_batch(cnt, D, I, P, N, 1 if is_csr else 0, ctypes.byref(p), out, status)
    if rc:
        first = next((k for k in range(cnt) if status[k] != 0), -1)
        for k in range(cnt):
            if out[k]:
                lib().ilupp_hip_ml_destroy(out[k])
        msg = lib().ilupp_hip_last_error().decode() + " (matrix %d of the batch, status %d)" % (first, status[first] if first >= 0 else rc)
        if rc == ILUPP_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise RuntimeError(msg)
    return [MultilevelPreconditioner(_VP(out[k]), ns[k]) for k in range(cnt)]


def solve(A_data, A_indices, A_indptr, is_csr, rhs, rtol, atol, max_iter, param):
    """_ilupp.solve, binding.cpp:200-230: (x, iterations, relative reduction reached, residual norm reached); RuntimeError("did not
    converge") as there.  The iteration runs inside the library (ilupp_hip_solve): no torch, no Python in the loop."""
    p = param if isinstance(param, MLParams) else param._to_ml_params()
    args, keep = _matrix_args(A_data, A_indices, A_indptr, is_csr)
    b = _check_real(rhs, "b")
    if b.shape[0] != args[3]:
        raise RuntimeError("right-hand side has wrong size!")
    ba = np.frombuffer(b, dtype=np.float64)
    x = np.empty(args[3], dtype=np.float64)
    it, rel, res = ctypes.c_int32(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    rc = lib().ilupp_hip_solve(*args, ba.ctypes.data, ba.shape[0], rtol, atol, max_iter, ctypes.byref(p), x.ctypes.data, ctypes.byref(it),
                               ctypes.byref(rel), ctypes.byref(res))
    if rc:
        _raise(rc)
    return x, it.value, rel.value, res.value


def MultilevelILUCDPPreconditioner_device(d_data, d_indices, d_indptr, n, is_csr, param):
    p = param if isinstance(param, MLParams) else param._to_ml_params()
    h = _VP()
    rc = lib().ilupp_hip_ml_create_device(d_data, d_indices, d_indptr, n, 1 if is_csr else 0, ctypes.byref(p), ctypes.byref(h))
    if rc:
        _raise(rc)
    return MultilevelPreconditioner(h, n)


def _create(fn, A_data, A_indices, A_indptr, is_csr, *extra):
    args, keep = _matrix_args(A_data, A_indices, A_indptr, is_csr)
    h = _VP()
    rc = fn(*args, *extra, ctypes.byref(h))
    if rc:
        _raise(rc)
    return Preconditioner(h)


# ---- factories, binding.cpp:299-310, 366-397 ------------------------------------------------------
def ILU0Preconditioner(A_data, A_indices, A_indptr, is_csr):
    return _create(lib().ilupp_hip_ilu0_create, A_data, A_indices, A_indptr, is_csr)


def ILU0Preconditioner_device(d_data, d_indices, d_indptr, n, is_csr, nnz=None):
    """ILU(0) of a matrix already resident in HBM (device pointers as ints).  nnz: the number of stored entries when the caller knows it
    (the length of its arrays): the construction then need not read indptr[n] back (ilupp_hip_ilu0_create_device_nnz)."""
    h = _VP()
    if nnz is not None:
        rc = lib().ilupp_hip_ilu0_create_device_nnz(d_data, d_indices, d_indptr, n, int(nnz), 1 if is_csr else 0, ctypes.byref(h))
    else:
        rc = lib().ilupp_hip_ilu0_create_device(d_data, d_indices, d_indptr, n, 1 if is_csr else 0, ctypes.byref(h))
    if rc:
        _raise(rc)
    return Preconditioner(h)


def _create_device(fn, d_data, d_indices, d_indptr, n, is_csr, *extra):
    h = _VP()
    rc = fn(d_data, d_indices, d_indptr, n, 1 if is_csr else 0, *extra, ctypes.byref(h))
    if rc:
        _raise(rc)
    return Preconditioner(h)


def ILUTPreconditioner_device(d_data, d_indices, d_indptr, n, is_csr, max_fill_in, threshold):
    return _create_device(lib().ilupp_hip_ilut_create_device, d_data, d_indices, d_indptr, n, is_csr,
                          ctypes.c_int32(int(max_fill_in)), ctypes.c_double(float(threshold)))


def ILUCPreconditioner_device(d_data, d_indices, d_indptr, n, is_csr, max_fill_in, threshold):
    return _create_device(lib().ilupp_hip_iluc_create_device, d_data, d_indices, d_indptr, n, is_csr,
                          ctypes.c_int32(int(max_fill_in)), ctypes.c_double(float(threshold)))


def IChol0Preconditioner_device(d_data, d_indices, d_indptr, n, is_csr):
    return _create_device(lib().ilupp_hip_ichol0_create_device, d_data, d_indices, d_indptr, n, is_csr)


def ICholTPreconditioner_device(d_data, d_indices, d_indptr, n, is_csr, add_fill_in, threshold):
    return _create_device(lib().ilupp_hip_icholt_create_device, d_data, d_indices, d_indptr, n, is_csr,
                          ctypes.c_int32(int(add_fill_in)), ctypes.c_double(float(threshold)))


def set_cache_limit(nbytes):
    """limit (bytes) of the device buffers kept for the next construction (include/ilupp_hip.h: ilupp_hip_set_cache_limit)"""
    rc = lib().ilupp_hip_set_cache_limit(ctypes.c_ulonglong(int(nbytes)))
    if rc:
        _raise(rc)


def cached_bytes():
    return int(lib().ilupp_hip_cached_bytes())


def live_blocks():
    return int(lib().ilupp_hip_live_blocks())


def release_cached_memory():
    """hand the device buffers the library keeps for the next construction back to the driver"""
    rc = lib().ilupp_hip_release_cached_memory()
    if rc:
        _raise(rc)


def set_caller_stream(stream_handle, enable=True):
    """order the *_device entry points of this thread after / before `stream_handle` (a hipStream_t as int; 0 = the
    legacy default stream); see include/ilupp_hip.h"""
    lib().ilupp_hip_set_caller_stream(_VP(stream_handle or 0), 1 if enable else 0)


def ILUTPreconditioner(A_data, A_indices, A_indptr, is_csr, max_fill_in, threshold):
    return _create(lib().ilupp_hip_ilut_create, A_data, A_indices, A_indptr, is_csr,
                   ctypes.c_int32(int(max_fill_in)), ctypes.c_double(float(threshold)))


def ILUCPreconditioner(A_data, A_indices, A_indptr, is_csr, max_fill_in, threshold):
    """binding.cpp:329-340"""
    return _create(lib().ilupp_hip_iluc_create, A_data, A_indices, A_indptr, is_csr,
                   ctypes.c_int32(int(max_fill_in)), ctypes.c_double(float(threshold)))


def IChol0Preconditioner(A_data, A_indices, A_indptr, is_csr):
    return _create(lib().ilupp_hip_ichol0_create, A_data, A_indices, A_indptr, is_csr)


def ICholTPreconditioner(A_data, A_indices, A_indptr, is_csr, add_fill_in, threshold):
    return _create(lib().ilupp_hip_icholt_create, A_data, A_indices, A_indptr, is_csr,
                   ctypes.c_int32(int(add_fill_in)), ctypes.c_double(float(threshold)))


# ---- stand-alone factor functions, binding.cpp:399-447 ----------------------------------------------
def ilu0(A_data, A_indices, A_indptr, is_csr):
    return tuple(ILU0Preconditioner(A_data, A_indices, A_indptr, is_csr).factors_info())


def ilut(A_data, A_indices, A_indptr, is_csr, fill_in, threshold):
    return tuple(ILUTPreconditioner(A_data, A_indices, A_indptr, is_csr, fill_in, threshold).factors_info())


def iluc(A_data, A_indices, A_indptr, is_csr, fill_in, threshold):
    """binding.cpp:449-460"""
    return tuple(ILUCPreconditioner(A_data, A_indices, A_indptr, is_csr, fill_in, threshold).factors_info())


def ichol0(A_data, A_indices, A_indptr, is_csr):
    return IChol0Preconditioner(A_data, A_indices, A_indptr, is_csr).factors_info()[0]


def icholt(A_data, A_indices, A_indptr, is_csr, add_fill_in, threshold):
    return ICholTPreconditioner(A_data, A_indices, A_indptr, is_csr, add_fill_in, threshold).factors_info()[0]


class PivotedPreconditioner:
    """ILUCPPreconditioner of binding.cpp:343-356: apply / apply_trans in place, total_nnz, factors_info(), permutations()"""
    memory = 0.0
    memory_used_calculations = 0.0
    memory_allocated_calculations = 0.0
    exists = True
    special_info = ""

    def __init__(self, handle, n, is_csr, rows=False):
        self._h, self._n, self._csr, self._rows = handle, n, bool(is_csr), bool(rows)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.ilupp_hip_ilucp_destroy(h)

    def _solve(self, x, transpose):
        mv = _check_real(x, "b")
        if mv.readonly:
            raise RuntimeError("b must be writable")
        a = np.frombuffer(mv, dtype=np.float64)
        if a.shape[0] != self._n:
            raise RuntimeError("vector has wrong size for preconditioner!")
        rc = _lib_holding_gil().ilupp_hip_ilucp_apply(self._h, a.ctypes.data, a.shape[0], transpose)
        if rc:
            _raise(rc)

    def apply(self, x):
        self._solve(x, 0)

    def apply_trans(self, x):
        self._solve(x, 1)

    @property
    def total_nnz(self):
        return int(lib().ilupp_hip_ilucp_total_nnz(self._h))

    @property
    def zero_pivots(self):
        return int(lib().ilupp_hip_ilucp_zero_pivots(self._h))

    @property
    def kernel_ms(self):
        ms = ctypes.c_float()
        lib().ilupp_hip_ilucp_info(self._h, None, None, None, ctypes.byref(ms))
        return ms.value

    def raw(self):
        """(L, U, perm) as ILUCP4 returns them for the major-order view of the input: L by columns, U by rows (pivot first, original column
        indices), each (data, indices, indptr)"""
        n, nl, nu = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        rc = lib().ilupp_hip_ilucp_info(self._h, ctypes.byref(n), ctypes.byref(nl), ctypes.byref(nu), None)
        if rc:
            _raise(rc)
        Ld, Li, Lp = np.empty(nl.value), np.empty(nl.value, dtype=np.int32), np.empty(n.value + 1, dtype=np.int32)
        Ud, Ui, Up = np.empty(nu.value), np.empty(nu.value, dtype=np.int32), np.empty(n.value + 1, dtype=np.int32)
        perm = np.empty(n.value, dtype=np.int32)
        rc = lib().ilupp_hip_ilucp_copy(self._h, Ld.ctypes.data, Li.ctypes.data, Lp.ctypes.data, Ud.ctypes.data, Ui.ctypes.data, Up.ctypes.data, perm.ctypes.data)
        if rc:
            _raise(rc)
        return (Ld, Li, Lp), (Ud, Ui, Up), perm

    def factors_info(self):
        """[left, right] as the class holds them (preconditioner_implementation.h:1117-1147): COLUMN input: L by columns, U by rows; ROW input:
        the factors of the transposed matrix change sides and labels (transpose_in_place): U's arrays as a column matrix, L's as a row matrix"""
        L, U, _ = self.raw()
        n = self._n
        if self._rows:
            # ILUTP (preconditioner_implementation.h:1050-1078): ROW input: L by rows, U by rows; COLUMN input: the factors of the transposed
            # matrix change sides and labels: U's arrays as a column matrix, L's as a column matrix
            if self._csr:
                return [(L[0], L[1], L[2], True, n, n), (U[0], U[1], U[2], True, n, n)]
            return [(U[0], U[1], U[2], False, n, n), (L[0], L[1], L[2], False, n, n)]
        if not self._csr:
            return [(L[0], L[1], L[2], False, n, n), (U[0], U[1], U[2], True, n, n)]
        return [(U[0], U[1], U[2], False, n, n), (L[0], L[1], L[2], True, n, n)]

    def permutations(self):
        """binding.cpp:178-196: (left, right) -- the permutation belongs to the factor U came from"""
        perm = self.raw()[2]
        if self._rows:
            return (None, perm) if self._csr else (perm, None)
        return (perm, None) if self._csr else (None, perm)

    def print_info(self):
        print("An incomplete LU factorisation with column pivoting: %d entries" % self.total_nnz)


def ILUCPPreconditioner(A_data, A_indices, A_indptr, is_csr, max_fill_in, threshold, piv_tol, row_pos, mem_factor):
    """binding.cpp:343-356"""
    args, keep = _matrix_args(A_data, A_indices, A_indptr, is_csr)
    h = _VP()
    rc = lib().ilupp_hip_ilucp_create(*args, int(max_fill_in), float(threshold), float(piv_tol), int(row_pos), float(mem_factor), ctypes.byref(h))
    if rc:
        _raise(rc)
    return PivotedPreconditioner(h, args[3], is_csr)


def ILUTPPreconditioner(A_data, A_indices, A_indptr, is_csr, max_fill_in, threshold, piv_tol, row_pos, mem_factor):
    """binding.cpp:313-326"""
    args, keep = _matrix_args(A_data, A_indices, A_indptr, is_csr)
    h = _VP()
    rc = lib().ilupp_hip_ilutp_create(*args, int(max_fill_in), float(threshold), float(piv_tol), int(row_pos), float(mem_factor), ctypes.byref(h))
    if rc:
        _raise(rc)
    return PivotedPreconditioner(h, args[3], is_csr, rows=True)
