"""Golden vectors of ILUTPPreconditioner (ILUTP2, ILUTP.hpp:13-140; binding.cpp:313-326) from the REAL reference, for the
oracle's restatement and the GPU kernel: the reference's own test matrices (test/tests.py:9-42) and config-shaped
small ones, CSR and CSC, several (fill_in, threshold, piv_tol) -- both factors, the permutation, apply(b), apply_trans(b).

Run in the build container only:   make -C oracle ref && python tests/golden/make_golden_ilutp.py    -> tests/golden/ilutp.npz"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import matgen  # noqa: E402
import ml_cases as C  # noqa: E402
from make_golden import laplace2d_matrix, random_matrix  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [(100, 0.1, 0.1), (100, 0.0, 0.0), (3, 1e-3, 1.0), (8, 1e-2, 0.5), (1, 0.1, 0.1)]      # (fill_in, threshold, piv_tol); the first: the class' defaults


def matrices():
    yield "laplace2d", laplace2d_matrix(400)
    yield "random", random_matrix(60)
    yield "rdd_300", sp.csr_matrix(matgen.random_dd(300, k=7, diag=3.0), shape=(300, 300))
    yield "weak_200", C.weak_random(200, 0.04, 0.05, 23)
    yield "offdiag_150", C.offdiag_random(150, 3)


def main():
    out = {}
    for name, A in matrices():
        for fmt in ("csr", "csc"):
            M = A.asformat(fmt).copy()
            M.sort_indices()
            key = "%s_%s" % (name, fmt)
            a = (M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32), fmt == "csr")
            out[key + "/data"], out[key + "/indices"], out[key + "/indptr"] = a[0], a[1], a[2]
            b = C.rhs(M.shape[0])
            for fill, thr, tol in CASES:
                tag = "%s/f%d_t%g_p%g" % (key, fill, thr, tol)
                R = O.ILUTP(O.ref(), a, fill_in=fill, threshold=thr, piv_tol=tol)
                for nm, arr in zip(("L_data", "L_indices", "L_indptr", "U_data", "U_indices", "U_indptr"), R.L + R.U):
                    out[tag + "/" + nm] = arr
                out[tag + "/perm"] = R.perm
                out[tag + "/apply"] = R.apply(b)
                out[tag + "/apply_trans"] = R.apply(b, O.TRANSPOSE)
                print(tag, len(R.L[0]), len(R.U[0]), int((R.perm != np.arange(len(R.perm))).sum()))
    path = os.path.join(HERE, "ilutp.npz")
    np.savez_compressed(path, **out)
    print("ilutp.npz:", len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
