"""the two slow multilevel cases with the attempts' log (development tool)"""
import sys, os, time, faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "500")), exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp
import torch  # noqa: F401
import ilupp_amd as ilupp
from ilupp_amd import _native
import matgen
which = sys.argv[1]
if which == "weak":
    A = (sp.random(2000, 2000, density=0.004, random_state=np.random.default_rng(7), format='csr') + sp.eye(2000) * 0.3).tocsr(); thr = 0.05
else:
    A = sp.csr_matrix(matgen.random_dd(20000, k=8, diag=2.0), shape=(20000, 20000)); thr = 1e-2
A.sort_indices(); A.indices = A.indices.astype(np.int32); A.indptr = A.indptr.astype(np.int32)
p = ilupp.iluplusplus_precond_parameter(); p.default_configuration(1); p.threshold = thr
t0 = time.time()
G = _native.MultilevelILUCDPPreconditioner(A.data, A.indices, A.indptr, True, p)
print("created %.2fs" % (time.time() - t0), G.levels(), G.total_nnz, G.timings(), flush=True)
b = np.ones(A.shape[0])
for rep in range(3):
    x = b.copy(); t0 = time.time(); G.apply(x); print("apply %.2f ms wall" % (1e3 * (time.time() - t0)), G.timings()["last_apply_ms"], flush=True)
x = b.copy(); G.apply_trans(x); print("apply_trans", G.timings()["last_apply_ms"], flush=True)
