"""Golden vectors from the REAL reference for the rows of SURVEY section 8(f) that lie on top of the factorisations:

* f2: the reference's BiCGstab (iterative_solvers_implementation.h:385-530) with an ILU(0) preconditioner applied from the left,
  iterates after 1..6 iterations from the zero start vector (pins ilupp_amd.device.bicgstab);
* f3: the multilevel ILU++ preconditioner (binding.cpp:284-298 -> preconditioner_implementation.h:1350-1665) for the presets of
  iluplusplus_precond_parameter::default_configuration (parameters_implementation.h:538-609): apply(b), apply_trans(b), number of
  levels and total_nnz on the reference's own test matrices (test/tests.py:9-36, 344-402) and config-shaped small cases.

Run in the build container only:   make -C oracle ref && python tests/golden/make_golden_ml.py    -> tests/golden/ml.npz
(the fixture holds inputs and expected outputs only; oracle/ref_shim.cpp is the code that calls the reference)."""
import ctypes
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import matgen  # noqa: E402
from make_golden import laplace2d_matrix, random_matrix, put_mat, rhs  # noqa: E402
from oracle import oracle as O  # noqa: E402

ref = O.ref()
lib = ref.lib
_I32P = ctypes.POINTER(ctypes.c_int32)
_F64P = ctypes.POINTER(ctypes.c_double)
lib.ref_bicgstab_ilu0_left.argtypes = [ctypes.c_int32, _I32P, _I32P, _F64P, ctypes.c_int, _F64P, ctypes.c_int32, _F64P]
lib.ref_bicgstab_ilu0_left.restype = ctypes.c_int
lib.ref_ilupp_apply.argtypes = [ctypes.c_int32, _I32P, _I32P, _F64P, ctypes.c_int, ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
                                ctypes.c_int, _F64P, _I32P, _I32P]
lib.ref_ilupp_apply.restype = ctypes.c_int


def _args(M):
    d = np.ascontiguousarray(M.data, dtype=np.float64)
    i = np.ascontiguousarray(M.indices, dtype=np.int32)
    p = np.ascontiguousarray(M.indptr, dtype=np.int32)
    return (d, i, p), (ctypes.c_int32(p.shape[0] - 1), p.ctypes.data_as(_I32P), i.ctypes.data_as(_I32P), d.ctypes.data_as(_F64P),
                       ctypes.c_int(1 if sp.isspmatrix_csr(M) else 0))


def bicgstab(M, b, iters):
    keep, a = _args(M)
    x = np.zeros(M.shape[0])
    bb = np.ascontiguousarray(b, dtype=np.float64)
    rc = lib.ref_bicgstab_ilu0_left(*a, bb.ctypes.data_as(_F64P), ctypes.c_int32(iters), x.ctypes.data_as(_F64P))
    assert rc == 0
    return x


def ilupp_apply(M, config, threshold, fill_in, b, use):
    keep, a = _args(M)
    x = np.ascontiguousarray(b, dtype=np.float64).copy()
    lev, nnz = ctypes.c_int32(0), ctypes.c_int32(0)
    rc = lib.ref_ilupp_apply(*a, ctypes.c_int32(config), ctypes.c_double(threshold), ctypes.c_int32(fill_in), ctypes.c_int(use),
                             x.ctypes.data_as(_F64P), ctypes.byref(lev), ctypes.byref(nnz))
    return rc, x, lev.value, nnz.value


def main():
    out = {}
    # ---- BiCGstab: a nonsymmetric, diagonally dominant matrix with a convection term
    d, i, p = matgen.poisson3d(12, 10, 9)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n)).tolil()
    A = (sp.csr_matrix(A) + 0.3 * sp.diags([np.ones(n - 1)], [1], shape=(n, n), format="csr")).tocsr()
    A.sort_indices()
    put_mat(out, "bicg/A", (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True))
    b = rhs(n)
    out["bicg/b"] = b
    for k in range(1, 7):
        out["bicg/x_%d" % k] = bicgstab(A, b, k)
    # ---- multilevel ILU++: the reference's test matrices and two config-shaped ones, CSR and CSC
    mats = [("laplace2d", laplace2d_matrix(900)), ("random", random_matrix(60)),
            ("rdd_400", sp.csr_matrix(matgen.random_dd(400, k=9), shape=(400, 400))),
            ("p3d_6_7_5", sp.csr_matrix(matgen.poisson3d(6, 7, 5), shape=(210, 210)))]
    for name, A in mats:
        for fmt in ("csr", "csc"):
            M = A.tocsr() if fmt == "csr" else A.tocsc()
            M.sort_indices()
            n = M.shape[0]
            b = rhs(n)
            key = "ml_%s_%s" % (name, fmt)
            put_mat(out, key + "/A", (M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32), fmt == "csr"))
            for config in (-1, 0, 1, 10, 11):                   # -1: the default-constructed parameters (ilupp/__init__.py:190)
                for (thr, fill) in ((1.0, -1), (0.0, -1), (2.0, 20)):
                    tag = "%s/c%d_t%g_f%d" % (key, config, thr, fill)
                    rc, x, lev, nnz = ilupp_apply(M, config, thr, fill, b, O.ID)
                    if rc != 0:
                        out[tag + "_error"] = np.array([rc])
                        continue
                    rc2, xt, _, _ = ilupp_apply(M, config, thr, fill, b, O.TRANSPOSE)
                    assert rc2 == 0
                    out[tag + "_apply"] = x
                    out[tag + "_apply_trans"] = xt
                    out[tag + "_info"] = np.array([lev, nnz])
    np.savez_compressed(os.path.join(HERE, "ml.npz"), **out)
    print("ml.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "ml.npz")), "bytes")
    for k in sorted(out):
        if k.endswith("_info"):
            print(k, out[k])


if __name__ == "__main__":
    main()
