#!/usr/bin/env python3
"""IChol0 construct / factor kernel / apply on 7-point meshes, device-resident input.  usage: ichol0_time.py GRID ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
dev = torch.device("cuda", 0)
for arg in sys.argv[1:]:
    g = int(arg)
    d, i, p = matgen.poisson3d(g)
    n = p.shape[0] - 1
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    x = torch.ones(n, dtype=torch.float64, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P = _native.IChol0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    tm = P.timings()
    aps = []
    for k in range(3):
        x.fill_(1.0); torch.cuda.synchronize(); t0 = time.perf_counter()
        P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
        torch.cuda.synchronize(); aps.append(time.perf_counter() - t0)
    print("IChol0 %d^3: path %s construct %.2f ms (analysis %.2f, numeric %.2f, kernel %.3f) | applies %s ms"
          % (g, P.path(), 1e3 * best, tm["analysis_ms"], tm["numeric_ms"], tm["numeric_kernel_ms"], " ".join("%.2f" % (1e3 * a) for a in aps)), flush=True)
