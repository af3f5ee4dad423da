"""Golden vectors of ILUC (Crout ILU, reference ILUC.hpp:112-207 through binding.cpp:449-460 / ref_shim.cpp) from the REAL
reference.  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden_iluc.py        -> tests/golden/iluc.npz

Inputs: the reference's own test matrices (test/tests.py:9-36; stored, scipy's random stream is version dependent), small
config-shaped cases from tests/matgen.py (regenerated at test time), and edge cases (1x1, a structurally missing pivot, the
reservation of 10 x nnz(A) exceeded, ties in the top-k cut)."""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import matgen  # noqa: E402
from make_golden import laplace_matrix, laplace2d_matrix, random_matrix, put_mat, rhs  # noqa: E402
from oracle import oracle as O  # noqa: E402

ref = O.ref()
PARAMS = ((5, 0.1), (100, 0.0), (3, 1e-3), (1, 0.0), (20, 1e-2))


def run(out, key, M, params=PARAMS):
    n = M[2].shape[0] - 1
    b = rhs(n)
    for (p, t) in params:
        tag = "%s/iluc_%d_%g" % (key, p, t)
        try:
            L, U = ref.iluc(M, p, t)
        except O.OracleError as e:
            out[tag + "_error"] = np.array([e.code, e.row])          # (ORC_ERR_ZERO_PIVOT, k) or (ORC_ERR_MEMORY, -1)
            continue
        put_mat(out, tag + "_L", L)
        put_mat(out, tag + "_U", U)
        out[tag + "_apply"] = ref.apply_lu(L, U, b, O.ID)
        out[tag + "_apply_trans"] = ref.apply_lu(L, U, b, O.TRANSPOSE)


def main():
    out = {}
    for name, A in (("laplace", laplace_matrix(50)), ("laplace2d", laplace2d_matrix(50)), ("random", random_matrix(50))):
        for fmt in ("csr", "csc"):
            M = A.tocsr() if fmt == "csr" else A.tocsc()
            M.sort_indices()
            key = "ref_%s_%s" % (name, fmt)
            put_mat(out, key + "/A", (M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32), fmt == "csr"))
            run(out, key, (M.data, M.indices, M.indptr, fmt == "csr"))
    for name, gen in (("p2d_12", lambda: matgen.poisson2d(12)), ("p3d_7", lambda: matgen.poisson3d(7)), ("p3d_5_9_4", lambda: matgen.poisson3d(5, 9, 4)),
                      ("rdd_300", lambda: matgen.random_dd(300, k=9)), ("rdd_600", lambda: matgen.random_dd(600, k=11, diag=3.0))):
        d, i, p = gen()
        n = p.shape[0] - 1
        A = sp.csr_matrix((d, i, p), shape=(n, n))
        for fmt in ("csr", "csc"):
            M = A if fmt == "csr" else A.tocsc()
            run(out, "cfg_%s_%s" % (name, fmt), (M.data, M.indices, M.indptr, fmt == "csr"))
    # edges
    one = (np.array([2.5]), np.array([0], dtype=np.int32), np.array([0, 1], dtype=np.int32), True)
    run(out, "edge_one", one, ((5, 0.1),))
    # row 2 has no entry at or right of the diagonal that survives: structurally missing pivot -> "zero pivot" with k
    Z = sp.csr_matrix(np.array([[4.0, 1.0, 0.0, 0.0], [1.0, 4.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0], [0.0, 0.0, 1.0, 4.0]]))
    Z.eliminate_zeros(); Z.sort_indices()
    put_mat(out, "edge_nopivot/A", (Z.data, Z.indices.astype(np.int32), Z.indptr.astype(np.int32), True))
    run(out, "edge_nopivot", (Z.data, Z.indices, Z.indptr, True), ((5, 0.1),))
    # ties: all off-diagonal magnitudes equal, cut in the middle of them
    d, i, p = matgen.poisson3d(6)
    run(out, "edge_ties", (d, i, p, True), ((2, 0.0), (3, 0.0), (4, 0.0)))
    np.savez_compressed(os.path.join(HERE, "iluc.npz"), **out)
    print("wrote iluc.npz:", len(out), "arrays")


main()
