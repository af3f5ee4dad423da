#!/usr/bin/env python3
"""Per-segment cycle sums of one wave (diagnostics build lib_stamp.so: profiles/tools/mkst.sh stamp -DST_STAMP).
usage: ILUPP_HIP_LIBRARY=profiles/tools/lib_stamp.so st_stamp.py [2d|GRID]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
arg = sys.argv[1] if len(sys.argv) > 1 else "2d"
d, i, p = matgen.poisson2d(200) if arg == "2d" else matgen.poisson3d(int(arg))
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
x = torch.ones(n, dtype=torch.float64, device=dev)
for rep in range(3):
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
t = P.timings()
buf = (ctypes.c_ulonglong * 32)()
lib = _native.lib()
assert lib.ilupp_hip_debug_stamps(buf) == 0
names = ["loop", "barrier", "lds", "ghost", "compute", "ldswr+stores", "loads"]
for nm, off in (("factor", 0), ("lsolve", 10), ("usolve", 20)):
    steps = max(1, buf[off + 8])
    print("%-7s steps %5d  kernel %.3f ms | " % (nm, steps, t["numeric_kernel_ms" if off == 0 else ("lsolve_kernel_ms" if off == 10 else "usolve_kernel_ms")]) +
          "  ".join("%s %.0f" % (names[j], buf[off + j] / steps) for j in range(7)) + "  | sum %.0f ticks/step (100 MHz ticks)" % (sum(buf[off + j] for j in range(7)) / steps))
