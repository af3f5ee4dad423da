#!/usr/bin/env python3
"""Where the waves of the wave-exchange factor kernel wait (diagnostics build: profiles/tools/mkwx.sh stamp -DWX_STAMP): per role, the
share of the main loop spent at the step barriers -- the role that waits least is the one the others wait for.
usage: ILUPP_HIP_LIBRARY=profiles/tools/lib_stamp.so wf_wait.py [GRID]"""
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
buf = (ctypes.c_ulonglong * (4096 * 16))()
assert _native.lib().ilupp_hip_debug_wf_wait(buf) == 0
Ty, Tz = (dims[1] + 15) // 16, (dims[2] + 15) // 16
a = np.array(buf[:Ty * Tz * 16], dtype=np.float64).reshape(Ty * Tz, 16)
print("factor kernel %.1f us (stamped build), %d tiles" % (1e3 * t["numeric_kernel_ms"], Ty * Tz))
loop = a[:, 12]; steps = a[:, 15]
print("cycles per step (wave 0 of each tile): median %.0f, min %.0f, max %.0f" % (np.median(loop / steps), (loop / steps).min(), (loop / steps).max()))
def share(cols):
    return np.median(a[:, cols].mean(axis=1) / loop)
print("share of the loop spent waiting at barriers: consumers %.2f  courier %.2f  producers %.2f" % (share([0, 1, 2, 3]), share([4]), share([5, 6, 7, 8, 9, 10])))
print("per consumer wave: " + " ".join("%.2f" % np.median(a[:, w] / loop) for w in range(4)))
print("per producer wave: " + " ".join("%.2f" % np.median(a[:, 5 + w] / loop) for w in range(6)))
print("courier: share of the loop inside the deliver (poll) section %.2f; producer wave 0: share inside write + load issue %.2f" % (np.median(a[:, 13] / loop), np.median(a[:, 14] / loop)))
for name, sel in (("first tile", [0]), ("middle tiles", [Ty * (Tz // 2) + Ty // 2, Ty * (Tz // 2) + Ty // 2 + 1]), ("last tile", [Ty * Tz - 1])):
    b = a[sel]
    print("%-12s cycles/step %.0f  wait share: consumers %.2f courier %.2f producers %.2f  courier poll %.2f  producer work %.2f" % (
        name, (b[:, 12] / b[:, 15]).mean(), (b[:, 0:4].mean(axis=1) / b[:, 12]).mean(), (b[:, 4] / b[:, 12]).mean(),
        (b[:, 5:11].mean(axis=1) / b[:, 12]).mean(), (b[:, 13] / b[:, 12]).mean(), (b[:, 14] / b[:, 12]).mean()))
