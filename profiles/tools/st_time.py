#!/usr/bin/env python3
"""Timings of the ILU(0) construction + apply on a 7-point mesh, device-resident, for one or more builds of the library.
usage: st_time.py GRID|NX,NY,NZ [lib.so ...]      (no lib: the in-tree one)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] != "--child":
    for lib in sys.argv[2:]:
        env = dict(os.environ, ILUPP_HIP_LIBRARY=os.path.abspath(lib))
        subprocess.call([sys.executable, os.path.abspath(__file__), sys.argv[1], "--child", os.path.basename(lib)], env=env)
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, time
import matgen
from ilupp_amd import _native
dims = [int(v) for v in sys.argv[1].split(",")]; g = dims[0]; tag = sys.argv[3] if len(sys.argv) > 3 else "in-tree"
d, i, p = matgen.poisson3d(*dims)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
x = torch.ones(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
rows = []
for rep in range(7):
    x.fill_(1.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    t1 = time.perf_counter()
    t = P.timings(); t["wall_ms"] = 1e3 * (t1 - t0)
    rows.append(t)
    cs = float(x.sum().item())
    P = None
med = {k: float(np.median([r[k] for r in rows[2:]])) for k in rows[0]}
print("%-14s g=%d " % (tag, g) + " ".join("%s=%.3f" % (k.replace("_ms", "").replace("_kernel", "K"), v) for k, v in med.items()) + " checksum=%.9f" % cs, flush=True)
