#!/usr/bin/env python3
"""Timeline of the forward sweep's workgroups (diagnostics build lib_stamp.so): when each 16x16 patch of lines got its
first row and its last row.  usage: ILUPP_HIP_LIBRARY=profiles/tools/lib_stamp.so st_timeline.py GRID"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "128").split(",")]
dims = dims * 3 if len(dims) == 1 else dims
g = dims[0]
d, i, p = matgen.poisson3d(*dims)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
x = torch.ones(n, dtype=torch.float64, device=dev)
for rep in range(3):
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    x.fill_(1.0); torch.cuda.synchronize()
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
t = P.timings()
buf = (ctypes.c_ulonglong * (4096 * 4))()
assert _native.lib().ilupp_hip_debug_timeline(buf) == 0
Ty, Tz = dims[1] // 16, dims[2] // 16
T = Ty
a = np.array(buf[:Ty * Tz * 4], dtype=np.float64).reshape(Ty * Tz, 4)
t0 = a[:, 0].min()
a = (a - t0) / 100.0      # us
print("lsolve sweep %.1f us; tiles %d x %d; per tile: entry / first row / last row / exit (us)" % (1e3 * t["lsolve_kernel_ms"], T, T))
for z in range(Tz):
    print("  ".join("%6.1f %6.1f" % (a[z * Ty + y, 1], a[z * Ty + y, 2]) for y in range(Ty)))
if Ty != Tz:
    sys.exit(0)
diag = [a[k * T + k] for k in range(T)]
print("diagonal tiles (k,k): first-row times", " ".join("%.1f" % v[1] for v in diag))
print("  first-row deltas along the diagonal:", " ".join("%.1f" % (diag[k + 1][1] - diag[k][1]) for k in range(T - 1)))
print("  duration first->last row per diagonal tile:", " ".join("%.1f" % (v[2] - v[1]) for v in diag))
