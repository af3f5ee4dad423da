"""BASELINE config 5 as named (default_configuration(10): maximum weighted matching + the factorisation WITH pivoting) -- construct times.
   python profiles/tools/ml_c5p.py [n ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import matgen
import ilupp_amd as ilupp
from ilupp_amd import _native

for n in [int(v) for v in sys.argv[1:]] or [100000]:
    d, i, p = matgen.random_dd(n, 8, 25.0, 12345)
    for thr in (1.0, 1e-3):
        prm = ilupp.iluplusplus_precond_parameter()
        prm.default_configuration(10)
        prm.threshold = thr
        for rep in range(2):
            t0 = time.perf_counter()
            P = _native.MultilevelILUCDPPreconditioner(d, i, p, True, prm)
            t1 = time.perf_counter()
        x = np.ones(n); P.apply(x)
        print("n", n, "threshold", thr, "levels", P.levels(), "total_nnz", P.total_nnz, "construct %.3f s" % (t1 - t0), P.timings(), flush=True)
