#!/usr/bin/env python3
"""Per-segment tick sums (100 MHz) of one consumer wave and one producer wave of the direct-feed factor kernel
(diagnostics build: profiles/tools/mksd.sh stamp "-DSD_STAMP [-DSD_STAMP_WG=n]").
usage: ILUPP_HIP_LIBRARY=profiles/tools/lib_stamp.so sd_stamp.py [GRID]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d, i, p = matgen.poisson3d(g)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
for rep in range(3):
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
t = P.timings()
buf = (ctypes.c_ulonglong * 32)()
lib = _native.lib()
assert lib.ilupp_hip_debug_sd_stamps(buf) == 0
for nm, off, names in (("consumer", 0, ["loop", "barrier", "lds-read", "compute", "handoff+stores"]),
                       ("producer", 16, ["loop", "barrier-1", "wait-loads", "write+load", "barrier-2"])):
    steps = max(1, buf[off + 6])
    print("%-8s steps %5d kernel %.3f ms | " % (nm, steps, t["numeric_kernel_ms"]) +
          "  ".join("%s %.1f" % (names[j], 10.0 * buf[off + j] / steps) for j in range(5)) +
          " | sum %.1f ns/step" % (10.0 * sum(buf[off + j] for j in range(5)) / steps))
print("kernel start->end: %d memtime ticks, %d realtime ticks (100 MHz => %.3f ms); WG0 alive %d ticks" % (buf[10] - buf[8], buf[11] - buf[9], (buf[11] - buf[9]) / 1e5, buf[12] - buf[8]))
