"""ILUTPPreconditioner construct / apply times next to the reference (oracle/_ref where present, else the oracle): python profiles/tools/ilutp_time.py [n ...]"""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import matgen
import ilupp_amd as ilupp
from oracle import oracle as O

for n in [int(v) for v in sys.argv[1:]] or [100000]:
    d, i, p = matgen.random_dd(n, 8, 25.0, 12345)
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    for fill, thr, tol in ((100, 0.1, 0.1), (10, 1e-3, 1.0)):
        for rep in range(2):
            t0 = time.perf_counter(); P = ilupp.ILUTPPreconditioner(A, fill_in=fill, threshold=thr, piv_tol=tol); t1 = time.perf_counter()
        x = np.ones(n)
        P.apply(x)
        t2 = time.perf_counter(); P.apply(x); t3 = time.perf_counter()
        lib = O.ref() if O.ref_available() else O.orc()
        t4 = time.perf_counter(); Q = O.ILUTP(lib, (d, i, p, True), fill_in=fill, threshold=thr, piv_tol=tol); t5 = time.perf_counter()
        same = np.array_equal(P.pr.raw()[2], Q.perm) and all(np.array_equal(a, b) for a, b in zip(P.pr.raw()[0] + P.pr.raw()[1], Q.L + Q.U))
        print("n", n, "fill", fill, "threshold", thr, "piv_tol", tol, "nnz", P.total_nnz, "construct %.3f s (kernel %.1f ms)" % (t1 - t0, P.pr.kernel_ms),
              "apply %.2f ms (host vector)" % ((t3 - t2) * 1e3), "| %s %.3f s" % (lib.prefix, t5 - t4), "identical" if same else "DIFFERENT", flush=True)
