"""BASELINE config C5's shape through the multilevel preconditioner, with the log of the level kernel's attempts (development tool)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import ilupp_amd as ilupp
from ilupp_amd import _native
import matgen
d, i, p = matgen.random_dd(1000000, 8, 25.0, 12345)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
prm = ilupp.iluplusplus_precond_parameter(); prm.default_configuration(1); prm.threshold = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    P = _native.MultilevelILUCDPPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, prm)
    print("construct %.1f ms" % (1e3 * (time.perf_counter() - t0)), P.timings(), P.total_nnz, flush=True)
    P = None
