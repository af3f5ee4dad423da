#!/usr/bin/env python3
"""where the construction time of ILU(0) goes on a box stencil: analysis / numeric phase / factor kernel.  usage: construct_split.py NX,NY[,NZ] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
dev = torch.device("cuda", 0)
for arg in sys.argv[1:]:
    dims = tuple(int(v) for v in arg.split(","))
    d, i, p = matgen.box_stencil(dims)
    n = p.shape[0] - 1
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        tm = P.timings()
    print(dims, P.path(), "wall %.1f ms" % (1e3 * wall), {k: round(v, 2) for k, v in tm.items()}, flush=True)
