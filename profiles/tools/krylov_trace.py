#!/usr/bin/env python3
"""Device-resident CG (ilupp_amd.device) on the 128^3 Laplacian with ICholT(0,0): run under
    rocprofv3 --memory-copy-trace --kernel-trace --stats -d OUT -- python3 profiles/tools/krylov_trace.py ITERS
twice with different ITERS: the number of memory copies must not depend on ITERS (nothing crosses PCIe per iteration)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp, torch
import matgen
import ilupp_amd.device as ild
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d, i, p = matgen.poisson3d(g)
n = p.shape[0] - 1
A = ild.DeviceCSR(torch.from_numpy(d).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda())
b = torch.ones(n, dtype=torch.float64, device="cuda")
M = ild.DevicePreconditioner("ICholT", A, add_fill_in=0, threshold=0.0)
ild.cg(A, b, M, maxiter=2); torch.cuda.synchronize()
t0 = time.perf_counter()
x = ild.cg(A, b, M, maxiter=iters)
torch.cuda.synchronize()
t1 = time.perf_counter()
r = b - A.matvec(x)
print("cg %d^3: %d iterations in %.3f ms (%.3f ms / iteration), |r|/|b| = %.3e" % (g, iters, 1e3 * (t1 - t0), 1e3 * (t1 - t0) / iters,
      float(torch.linalg.vector_norm(r) / torch.linalg.vector_norm(b))))
