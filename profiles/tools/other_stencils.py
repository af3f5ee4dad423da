#!/usr/bin/env python3
"""ILU(0) construct + apply on stencils the static form does not take (9-point 2-D, 27-point 3-D): which generation runs, how long,
array compare with the reference on one host core.  usage: other_stencils.py NX,NY[,NZ] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen, golden_util as G
from ilupp_amd import _native
from oracle import oracle as O
ref = O.ref() if O.ref_available() else O.orc()
dev = torch.device("cuda", 0)
for arg in sys.argv[1:]:
    dims = tuple(int(v) for v in arg.split(","))
    d, i, p = matgen.box_stencil(dims)
    n = p.shape[0] - 1
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    x = torch.ones(n, dtype=torch.float64, device=dev)
    best, ap, ap1 = 1e9, 1e9, 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        for k in range(2):                     # (the first apply builds what the sweeps need; both are timed)
            x.fill_(1.0); torch.cuda.synchronize(); t0 = time.perf_counter()
            P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
            torch.cuda.synchronize()
            if k == 0: ap1 = min(ap1, time.perf_counter() - t0)
            else: ap = min(ap, time.perf_counter() - t0)
    t0 = time.perf_counter(); Lo, Uo = ref.ilu0((d, i, p, True)); t1 = time.perf_counter()
    xo = O.orc().apply_lu(Lo, Uo, np.ones(n), O.ID); t2 = time.perf_counter()
    F = P.factors_info()
    ok = G.mat_equal(tuple(F[0][:4]), Lo) and G.mat_equal(tuple(F[1][:4]), Uo) and np.array_equal(x.cpu().numpy(), xo)
    print("%d-point %s: n=%d nnz=%d path=%s | GPU construct %.2f ms, first apply %.2f ms, apply %.2f ms | reference on one core %.0f + %.0f ms | arrays equal: %s"
          % (3 ** len(dims), "x".join(map(str, dims)), n, p[-1], P.path(), 1e3 * best, 1e3 * ap1, 1e3 * ap, 1e3 * (t1 - t0), 1e3 * (t2 - t1), ok), flush=True)
