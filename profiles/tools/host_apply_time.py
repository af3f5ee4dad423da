#!/usr/bin/env python3
"""What the HOST entry point of apply costs next to the sweeps: ilupp_hip_apply on a numpy vector (pageable memory in, result back)
against ilupp_hip_apply_device on a vector resident in HBM; 3-D 7-point Poisson, ILU(0).   host_apply_time.py [GRID ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scipy.sparse as sp
import torch
import matgen
import ilupp_amd as ilupp

for g in [int(a) for a in sys.argv[1:]] or [128, 256]:
    d, i, p = matgen.poisson3d(g)
    n = p.shape[0] - 1
    P = ilupp.ILU0Preconditioner(sp.csr_matrix((d, i, p), shape=(n, n)))
    x = np.ones(n)
    ts = []
    for _ in range(7):
        x[:] = 1.0
        t0 = time.perf_counter(); P.apply(x); ts.append(time.perf_counter() - t0)
    xd = torch.ones(n, dtype=torch.float64, device="cuda")
    td = []
    for _ in range(7):
        xd.fill_(1.0); torch.cuda.synchronize()
        t0 = time.perf_counter(); P.pr.apply_device(xd.data_ptr(), n, transpose=False, sync=True); td.append(time.perf_counter() - t0)
    same = bool(np.array_equal(x, xd.cpu().numpy()))
    print("grid %d n %d: host apply %.2f ms (min %.2f), device apply %.3f ms, vector %.1f MB, same bits %s"
          % (g, n, 1e3 * np.median(ts), 1e3 * min(ts), 1e3 * np.median(td), 8e-6 * n, same), flush=True)
