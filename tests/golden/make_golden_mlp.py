"""Golden vectors of the multilevel ILU++ preconditioner WITH pivoting (partialILUCDP: the reference's default-constructed parameters and variations, tests/ml_cases.py PIVOT_PARAMS) from the REAL reference:
for every case of tests/ml_cases.py the number of levels, their sizes, total_nnz, a sha256 over every level's factors / middle
diagonal / permutations / scalings (extract_left_matrix(k) ... extract_right_scaling(k), preconditioner.h:288-296), and
apply(b) / apply_trans(b) in full.  The input matrices are stored as arrays.

Run in the build container only:   make -C oracle ref && python tests/golden/make_golden_mlp.py    -> tests/golden/mlp.npz
(oracle/ref_shim.cpp: ref_ml_* is the code that calls the reference, through its own setters)."""
import hashlib
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import ml_cases as C  # noqa: E402
from oracle import oracle as O  # noqa: E402


def digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def main():
    out = {}
    ref = O.ref()
    for name, A in C.matrices():
        for fmt in ("csr", "csc"):
            M = A.asformat(fmt).copy()
            M.sort_indices()
            key = "%s_%s" % (name, fmt)
            out[key + "/data"], out[key + "/indices"], out[key + "/indptr"] = M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32)
            n = M.shape[0]
            b = C.rhs(n)
            a = (M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32), fmt == "csr")
            for tag, thr, pre, knobs in C.PIVOT_PARAMS:
                k2 = "%s/%s" % (key, tag)
                try:
                    O.orc().ml(a, C.oracle_params(O, thr, pre, knobs))
                except O.OracleError:
                    raise
                R = ref.ml(a, C.oracle_params(O, thr, pre, knobs))
                nl = R.levels()
                out[k2 + "/info"] = np.array([nl, R.total_nnz()] + [R.level(k)["n"] for k in range(nl)], dtype=np.int64)
                out[k2 + "/levels_sha"] = np.stack([digest(C.level_arrays(R.level(k))) for k in range(nl)])
                out[k2 + "/apply"] = R.apply(b)
                out[k2 + "/apply_trans"] = R.apply(b, O.TRANSPOSE)
                print(k2, out[k2 + "/info"][:8])
    path = os.path.join(HERE, "mlp.npz")
    np.savez_compressed(path, **out)
    print("mlp.npz:", len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
