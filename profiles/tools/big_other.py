#!/usr/bin/env python3
"""IChol0 / ICholT(0,0) / ILUT on meshes with more lines than the chip has lanes: do they run, how long
usage: big_other.py NX,NY,NZ"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
dims = [int(v) for v in sys.argv[1].split(",")]
d, i, p = matgen.poisson3d(*dims)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
x = torch.ones(n, dtype=torch.float64, device=dev)
for name, make in (("ichol0", lambda: _native.IChol0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)),
                   ("icholt(0,0)", lambda: _native.ICholTPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, 0, 0.0)),
                   ("ilut(10,1e-3)", lambda: _native.ILUTPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, 10, 1e-3))):
    try:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P = make()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for tr in (False, True):
            x.fill_(1.0); torch.cuda.synchronize(); t2 = time.perf_counter()
            P.apply_device(x.data_ptr(), n, transpose=tr, sync=True)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            print("%-14s %s n=%d construct %.1f ms  apply(trans=%s) %.2f ms  sum %.6f" % (name, dims, n, 1e3 * (t1 - t0), tr, 1e3 * (t3 - t2), float(x.sum().item())), flush=True)
        del P
    except Exception as e:
        print("%-14s %s FAILED: %s" % (name, dims, e), flush=True)
