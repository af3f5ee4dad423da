import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import numpy as np, scipy.sparse as sp
import matgen, ilupp_amd as ilupp
from oracle import oracle as O
rng = np.random.default_rng(3)
for shape in ((70, 16, 16), (72, 16, 16), (64, 16, 16), (100, 16, 16)):
    d, i, p = matgen.poisson3d(*shape)
    d = d * (1.0 + 0.3 * rng.random(d.shape[0]))
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    P = ilupp.ILU0Preconditioner(A)
    Lo, Uo = O.orc().ilu0((A.data, A.indices, A.indptr, True))
    try:
        L, U = P.factors()
    except Exception as e:
        print(shape, "factors failed", e); continue
    ok_p = np.array_equal(U.indptr, Uo[2]) and np.array_equal(L.indptr, Lo[2])
    print(shape, P.pr.path(), "ptr ok", ok_p, flush=True)
    if ok_p:
        nx = shape[0]
        badU = np.flatnonzero(U.data != Uo[0]); badL = np.flatnonzero(L.data != Lo[0])
        rowsU = np.unique(np.searchsorted(U.indptr, badU, side="right") - 1)
        rowsL = np.unique(np.searchsorted(L.indptr, badL, side="right") - 1)
        print("  U bad entries", badU.size, "rows", rowsU.size, "k of first rows", (rowsU[:12] % nx), "lines", (rowsU[:12] // nx))
        print("  L bad entries", badL.size, "rows", rowsL.size, "k of first rows", (rowsL[:12] % nx), "lines", (rowsL[:12] // nx))
        ks = np.bincount(rowsU % nx, minlength=nx)
        print("  U bad rows per k:", ks)
