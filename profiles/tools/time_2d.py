import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import matgen
from ilupp_amd import _native
for nx in (1024, 4096):
    d, i, p = matgen.poisson2d(nx)
    n = p.shape[0] - 1
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    x = torch.ones(n, dtype=torch.float64, device=dev)
    best = 1e9
    for rep in range(4):
        x.fill_(1.0); torch.cuda.synchronize(); t0 = time.perf_counter()
        P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
        P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("5-point %d^2 n=%d: path %s, step %.2f ms, %s" % (nx, n, P.path(), 1e3 * best, {k: round(v, 3) for k, v in P.timings().items()}), flush=True)
