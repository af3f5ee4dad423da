import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, matgen, torch
from ilupp_amd import _native
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d, i, p = matgen.poisson3d(g, g, g)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    P = _native.ICholTPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, 0, 0.0)
    t1 = time.perf_counter()
    print("construct %.3f ms" % ((t1 - t0) * 1e3), P.path(), P.timings() if hasattr(P, "timings") else "", flush=True)
    del P
