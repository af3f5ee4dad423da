"""CPU tests: the C-ABI library loads and exports every symbol include/ilupp_hip.h declares; host-side
validation logic of the Python layer (no GPU compute)."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ilupp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ilupp_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ilupp_amd import _native
    lib = _native.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libilupp_hip.so lacks %s" % name
    assert sorted(_native.ABI_SYMBOLS) == declared


def test_index_size():
    from ilupp_amd import _native
    assert _native.index_size() == 4


def test_python_surface_matches_reference_names():
    import ilupp_amd as ilupp
    for name in ("ILU0Preconditioner", "ILUTPreconditioner", "IChol0Preconditioner", "ICholTPreconditioner",
                 "ilu0", "ilut", "ichol0", "icholt"):
        assert hasattr(ilupp, name)
    import inspect
    assert str(inspect.signature(ilupp.ILUTPreconditioner.__init__)) == "(self, A, fill_in=100, threshold=0.1)"
    assert str(inspect.signature(ilupp.ICholTPreconditioner.__init__)) == "(self, A, add_fill_in=0, threshold=0.0)"
    assert str(inspect.signature(ilupp.ilut)) == "(A, fill_in=100, threshold=0.1)"
    assert str(inspect.signature(ilupp.icholt)) == "(A, add_fill_in=0, threshold=0.0)"


def test_pivoting_classes_have_the_signatures_of_the_reference():
    """the reference's ILUTP / ILUCP classes (tests/test_gpu_ilucp.py, tests/test_gpu_ilutp.py)"""
    import inspect
    import ilupp_amd as ilupp
    A = sp.eye(4, format="csr")
    for cls in (ilupp.ILUTPPreconditioner, ilupp.ILUCPPreconditioner):
        assert str(inspect.signature(cls.__init__)) == "(self, A, fill_in=100, threshold=0.1, piv_tol=0.1, mem_factor=10.0)"
        assert callable(cls.permutations)
    del A


def test_input_validation_types():
    """exception types/messages of ilupp/__init__.py:55-71 and binding.cpp:33-98 (no GPU needed:
    all raised before the native call)"""
    import ilupp_amd as ilupp
    A = sp.eye(4, format="coo")
    with pytest.raises(TypeError, match="A must be a csr_matrix or a csc_matrix"):
        ilupp.ILU0Preconditioner(A)
    with pytest.raises(ValueError, match="A must be a square matrix!"):
        ilupp.ILU0Preconditioner(sp.csr_matrix(np.ones((2, 3))))
    B = sp.eye(4, format="csr")
    B.indices = B.indices.astype(np.int64)
    B.indptr = B.indptr.astype(np.int64)
    with pytest.raises(TypeError, match="8-byte indices"):
        ilupp.ILU0Preconditioner(B)
    C = sp.eye(4, format="csr", dtype=np.float32)
    with pytest.raises(RuntimeError, match=r"Expected d \(d\) array for A_data, got f!"):
        ilupp.ILU0Preconditioner(C)


def test_product_never_imports_oracle():
    """the product path must not route through the oracle (or any CPU fallback)"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ilupp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower().replace("no cpu fallback", ""), f


def test_pybind11_shim_surface_and_validation():
    """the compiled pybind11 shim over the C ABI (ilupp_amd/csrc/pybind_module.cpp): the names of the reference's `_ilupp`
    for this path, and the buffer checks of binding.cpp:33-98 with the reference's messages (raised before any GPU call)"""
    from ilupp_amd import _ilupp_hip as m
    for name in ("index_size", "ILU0Preconditioner", "IChol0Preconditioner", "ICholTPreconditioner", "ILUTPreconditioner",
                 "ILUCPreconditioner", "ILUTPPreconditioner", "ILUCPPreconditioner", "MultilevelILUCDPPreconditioner", "GenericLUPreconditioner", "GenericLLTPreconditioner", "ilu0", "ilut", "iluc", "ichol0", "icholt"):
        assert hasattr(m, name), name
    assert m.index_size() == 4
    for member in ("apply", "apply_trans", "total_nnz", "factors_info", "memory_used_calculations",
                   "memory_allocated_calculations", "memory", "exists", "special_info", "print_info"):
        assert hasattr(m.GenericLUPreconditioner, member) and hasattr(m.ILUTPreconditioner, member) and hasattr(m.ILUCPreconditioner, member), member
        assert hasattr(m.ILUTPPreconditioner, member) and hasattr(m.ILUCPPreconditioner, member) and hasattr(m.MultilevelILUCDPPreconditioner, member), member
    assert hasattr(m.ILUTPPreconditioner, "permutations") and hasattr(m.ILUCPPreconditioner, "permutations")      # binding.cpp:326, :356
    with pytest.raises(RuntimeError, match="indices and data should have the same size!"):
        m.ILUCPPreconditioner(np.ones(3), np.arange(2, dtype=np.int32), np.arange(4, dtype=np.int32), True, 5, 0.1, 0.1, -1, 10.0)
    d, i, p = np.ones(3), np.arange(3, dtype=np.int32), np.arange(4, dtype=np.int32)
    with pytest.raises(RuntimeError, match=r"Expected d \(d\) array for A_data, got f!"):
        m.ILU0Preconditioner(d.astype(np.float32), i, p, True)
    with pytest.raises(RuntimeError, match="Expected integer type with length 4 for A_indices"):
        m.ILU0Preconditioner(d, i.astype(np.int64), p, True)
    with pytest.raises(RuntimeError, match="matrix has size 0!"):
        m.ILU0Preconditioner(d[:0], i[:0], p[:1], True)
    with pytest.raises(RuntimeError, match="Expected 1D array"):
        m.ILU0Preconditioner(np.ones((2, 2)), i, p, True)
