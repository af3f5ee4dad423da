"""one multilevel case on the GPU with timing of the phases (development tool)"""
import sys, os, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp
import torch  # noqa: F401
import ilupp_amd as ilupp
from ilupp_amd import _native
import matgen
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "60")), exit=True)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 40
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
d, i, pp = matgen.poisson3d(g, g, g); A = sp.csr_matrix((d, i, pp))
A.indices = A.indices.astype(np.int32); A.indptr = A.indptr.astype(np.int32)
p = ilupp.iluplusplus_precond_parameter(); p.default_configuration(1); p.threshold = thr
t0 = time.time()
G = _native.MultilevelILUCDPPreconditioner(A.data, A.indices, A.indptr, True, p)
print("created %.2fs" % (time.time() - t0), G.levels(), G.total_nnz, G.timings(), flush=True)
b = np.ones(A.shape[0])
t0 = time.time(); x = b.copy(); G.apply(x); print("apply %.2fs" % (time.time() - t0), G.timings(), flush=True)
t0 = time.time(); x = b.copy(); G.apply(x); print("apply again %.3fs" % (time.time() - t0), G.timings(), flush=True)
t0 = time.time(); x = b.copy(); G.apply_trans(x); print("apply_trans %.2fs" % (time.time() - t0), flush=True)
