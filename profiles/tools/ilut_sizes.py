#!/usr/bin/env python3
"""ILUT construct times over a few shapes (device-resident input): C3, 3-D meshes, 2-D mesh, defaults.  usage: ilut_sizes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
dev = torch.device("cuda", 0)
cases = (("C3 random n=1e6 ILUT(10,1e-4)", matgen.random_dd(1000000, 19, 25.0, 12345), 10, 1e-4),
         ("poisson 64^3 ILUT(10,1e-4)", matgen.poisson3d(64), 10, 1e-4),
         ("poisson 96^3 ILUT(10,1e-4)", matgen.poisson3d(96), 10, 1e-4),
         ("poisson2d 1000^2 ILUT(10,1e-4)", matgen.poisson2d(1000), 10, 1e-4),
         ("random 2e5 k=9 ILUT(100,0.1)", matgen.random_dd(200000, k=9), 100, 0.1),
         ("random 2e5 k=9 ILUT(100,1e-3)", matgen.random_dd(200000, k=9), 100, 1e-3),
         ("poisson 48^3 ILUT(100,1e-3)", matgen.poisson3d(48), 100, 1e-3))
for name, (d, i, p), fill, tau in cases:
    n = p.shape[0] - 1
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P = _native.ILUTPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, fill, tau)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-36s n=%8d construct %9.1f ms  kernel %9.1f ms  nnz %d" % (name, n, 1e3 * best, P.timings()["numeric_kernel_ms"], P.total_nnz), flush=True)
