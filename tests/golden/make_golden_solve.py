"""Golden vectors of `_ilupp.solve` (binding.cpp:200-230: multilevel preconditioner + BiCGstab with SPLIT preconditioning from the
zero vector) from the REAL reference: for each case the solution, whether it converged, the number of iterations and the two
residual measures the binding returns.  The matrices are stored as arrays.

Run in the build container only:   make -C oracle ref && python tests/golden/make_golden_solve.py    -> tests/golden/solve.npz
(oracle/ref_shim.cpp: ref_solve is the code that calls the reference)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import ml_cases as C  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out = {}
    ref = O.ref()
    for name, A, b in C.solve_matrices():
        for fmt in ("csr", "csc"):
            M = A.asformat(fmt).copy()
            M.sort_indices()
            key = "%s_%s" % (name, fmt)
            out[key + "/data"], out[key + "/indices"], out[key + "/indptr"] = M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32)
            out[key + "/b"] = b
            a = (M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32), fmt == "csr")
            for tag, thr, pre, knobs, rtol, atol, max_iter in C.SOLVE_PARAMS:
                k2 = "%s/%s" % (key, tag)
                x, ok, it, rel, res = ref.solve(a, b, C.oracle_params(O, thr, pre, knobs), rtol, atol, max_iter)
                out[k2 + "/x"] = x
                out[k2 + "/info"] = np.array([float(ok), float(it), rel, res])
                print(k2, ok, it, rel, res)
    path = os.path.join(HERE, "solve.npz")
    np.savez_compressed(path, **out)
    print("solve.npz:", len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
