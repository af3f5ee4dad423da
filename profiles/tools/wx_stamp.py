#!/usr/bin/env python3
"""Cycles and clock of one wave of the wave-exchange sweeps (diagnostics build: profiles/tools/mkwx.sh stamp -DWX_STAMP).
usage: ILUPP_HIP_LIBRARY=profiles/tools/lib_stamp.so wx_stamp.py NX,NY,NZ"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
for arg in sys.argv[1:]:
    dims = [int(v) for v in arg.split(",")]
    d, i, p = matgen.poisson3d(*dims)
    n = p.shape[0] - 1
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    x = torch.ones(n, dtype=torch.float64, device=dev)
    for rep in range(3):
        P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
        P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    t = P.timings()
    buf = (ctypes.c_ulonglong * 16)()
    assert _native.lib().ilupp_hip_debug_wx_stamps(buf) == 0
    for nm, o, key in (("L", 0, "lsolve_kernel_ms"), ("U", 4, "usolve_kernel_ms")):
        cyc, rt, steps = buf[o], buf[o + 1], max(1, buf[o + 2])
        print("%s %s: kernel %.3f ms; wave 0 of workgroup 0: %d steps, %.1f cycles/step, %.1f ns/step, clock %.2f GHz" % (
            arg, nm, t[key], steps, cyc / steps, 10.0 * rt / steps, cyc / (10.0 * rt) if rt else 0))
    print("   (last sweep) slow-path entries %d, spins %d, lanes that were not ready (bit i = count i): %s" % (buf[8], buf[9], bin(buf[10])))
