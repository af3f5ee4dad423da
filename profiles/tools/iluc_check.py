#!/usr/bin/env python3
"""ILUC on the GPU against the oracle: usage iluc_check.py [n k diag fill tau]..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp
import matgen, golden_util as G
import ilupp_amd as ilupp
from oracle import oracle as O
def fac(F): return (F.data, F.indices, F.indptr, isinstance(F, sp.csr_matrix))
def run(n, k, diag, fill, tau, fmt="csr"):
    d, i, p = matgen.random_dd(n, k=k, diag=diag)
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    M = A if fmt == "csr" else A.tocsc()
    t0 = time.time()
    try:
        Lo, Uo = O.orc().iluc((M.data, M.indices, M.indptr, fmt == "csr"), fill, tau)
    except O.OracleError as e:
        Lo = Uo = None; print("oracle error", e)
    t1 = time.time()
    try:
        P = ilupp.ILUCPreconditioner(M, fill_in=fill, threshold=tau)
    except Exception as e:
        print(n, k, fill, tau, fmt, "GPU error:", e); return
    t2 = time.time()
    L, U = P.factors()
    ok = Lo is not None and G.mat_equal(fac(L), Lo) and G.mat_equal(fac(U), Uo)
    print(n, k, fill, tau, fmt, "OK" if ok else "MISMATCH", "oracle %.3f s gpu %.3f s kernel %.2f ms nnz %d+%d" % (t1 - t0, t2 - t1, P.pr.timings()["numeric_kernel_ms"], L.nnz, U.nnz), flush=True)
if len(sys.argv) > 1:
    a = sys.argv[1:]
    run(int(a[0]), int(a[1]), float(a[2]), int(a[3]), float(a[4]), a[5] if len(a) > 5 else "csr")
else:
    for (fill, tau) in ((5, 0.1), (100, 0.0), (3, 1e-3), (1, 0.0), (20, 1e-2)):
        run(600, 11, 3.0, fill, tau)
