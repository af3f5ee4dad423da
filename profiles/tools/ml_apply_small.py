"""apply of a multilevel object with small levels, with and without the one-workgroup sweeps (ILUPP_NO_SMALL_SWEEPS=1) (development tool)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp
import torch  # noqa: F401
import ilupp_amd as ilupp
from ilupp_amd import _native
import ml_cases as C
for n, dens, dg, thr in ((700, 0.01, 0.3, 0.05), (3000, 0.002, 0.3, 0.1), (6000, 0.001, 0.5, 0.2)):
    A = C.weak_random(n, dens, dg, 7)
    A.indices = A.indices.astype(np.int32); A.indptr = A.indptr.astype(np.int32)
    p = ilupp.iluplusplus_precond_parameter(); p.default_configuration(1); p.threshold = thr
    G = _native.MultilevelILUCDPPreconditioner(A.data, A.indices, A.indptr, True, p)
    b = np.ones(n)
    ts = []
    for rep in range(4):
        x = b.copy(); G.apply(x); ts.append(G.timings()["last_apply_ms"])
    print("n %d: %d levels %s, nnz %d, construct %.1f ms, apply %s ms" % (n, G.levels(), [G.level_sizes(k)[0] for k in range(G.levels())][:6], G.total_nnz,
          G.timings()["construct_ms"], ["%.3f" % t for t in ts]), flush=True)
