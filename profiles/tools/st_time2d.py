#!/usr/bin/env python3
"""single-tile timing (2-D 200x200: one workgroup, no border lanes) of the three static sweeps, for several builds.
usage: st_time2d.py [lib.so ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, ILUPP_HIP_LIBRARY=os.path.abspath(lib))
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", os.path.basename(lib)], env=env)
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
d, i, p = matgen.poisson2d(200)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
x = torch.ones(n, dtype=torch.float64, device=dev)
rows = []
for rep in range(8):
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    x.fill_(1.0); torch.cuda.synchronize()
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    rows.append(P.timings())
med = {k: float(np.median([r[k] for r in rows[2:]])) for k in rows[0]}
print("%-16s factorK %.1f us  lsolveK %.1f us  usolveK %.1f us   (400 steps each)" % (sys.argv[2] if len(sys.argv) > 2 else "in-tree",
      1e3 * med["numeric_kernel_ms"], 1e3 * med["lsolve_kernel_ms"], 1e3 * med["usolve_kernel_ms"]))
