#!/usr/bin/env python3
"""ILUC: device-resident construction + apply on the GPU next to the reference's C++ on one host core (and array compare).
usage: iluc_time.py [mesh G | random N K] FILL TAU"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen, golden_util as G
from ilupp_amd import _native
from oracle import oracle as O
a = sys.argv[1:]
if a[0] == "mesh":
    d, i, p = matgen.poisson3d(int(a[1])); fill, tau = int(a[2]), float(a[3]); name = "mesh %s^3" % a[1]
else:
    d, i, p = matgen.random_dd(int(a[1]), k=int(a[2])); fill, tau = int(a[3]), float(a[4]); name = "random n=%s k=%s" % (a[1], a[2])
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(x).to(dev) for x in (d, i, p))
best = None
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    P = _native.ILUCPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, fill, tau)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    best = min(best or 1e9, t1 - t0)
x = torch.ones(n, dtype=torch.float64, device=dev)
ap = []
for rep in range(3):
    x.fill_(1.0); torch.cuda.synchronize(); t0 = time.perf_counter()
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    torch.cuda.synchronize(); ap.append(time.perf_counter() - t0)
F = P.factors_info()
tm = P.timings()
ref = O.ref() if O.ref_available() else O.orc()
t0 = time.perf_counter()
Lo, Uo = ref.iluc((d, i, p, True), fill, tau)
t1 = time.perf_counter()
ok = G.mat_equal(tuple(F[0][:4]), Lo) and G.mat_equal(tuple(F[1][:4]), Uo)
xo = O.orc().apply_lu(Lo, Uo, np.ones(n), O.ID)
ok_x = np.array_equal(x.cpu().numpy(), xo)
print("%s fill=%d tau=%g: n=%d nnz(A)=%d nnz(L)+nnz(U)=%d | GPU construct %.1f ms (kernel %.1f ms), apply %.2f ms | reference on one core %.1f ms | arrays equal: %s, apply equal: %s"
      % (name, fill, tau, n, p[-1], F[0][0].shape[0] + F[1][0].shape[0], 1e3 * best, tm["numeric_kernel_ms"], 1e3 * min(ap), 1e3 * (t1 - t0), ok, ok_x), flush=True)
