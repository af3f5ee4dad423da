#!/usr/bin/env python3
"""Timeline of the wave-exchange factor kernel's workgroups (diagnostics build: profiles/tools/mkwx.sh stamp -DWX_STAMP): when each
16x16 patch of lines got its first and its last row.  usage: ILUPP_HIP_LIBRARY=profiles/tools/lib_stamp.so wf_timeline.py GRID"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256").split(",")]
dims = dims * 3 if len(dims) == 1 else dims
d, i, p = matgen.poisson3d(*dims)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
for rep in range(3):
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
t = P.timings()
buf = (ctypes.c_ulonglong * (4096 * 4))()
assert _native.lib().ilupp_hip_debug_wf_timeline(buf) == 0
Ty, Tz = dims[1] // 16, dims[2] // 16
a = np.array(buf[:Ty * Tz * 4], dtype=np.float64).reshape(Ty * Tz, 4)
t0 = a[:, 0].min()
a = (a - t0) / 100.0      # us
print("factor kernel %.1f us; tiles %d x %d; per tile: first row / last row of lane 0 (us)" % (1e3 * t["numeric_kernel_ms"], Ty, Tz))
for z in range(Tz):
    print("  ".join("%6.1f %6.1f" % (a[z * Ty + y, 1], a[z * Ty + y, 2]) for y in range(Ty)))
if Ty == Tz:
    diag = [a[k * Ty + k] for k in range(Ty)]
    print("diagonal tiles (k,k): first-row times", " ".join("%.1f" % v[1] for v in diag))
    print("  first-row deltas along the diagonal (two hops each):", " ".join("%.1f" % (diag[k + 1][1] - diag[k][1]) for k in range(Ty - 1)))
    print("  duration first->last row per diagonal tile:", " ".join("%.1f" % (v[2] - v[1]) for v in diag))
    print("  us per row:", " ".join("%.2f" % ((v[2] - v[1]) / (dims[0] - 1)) for v in diag))
