import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import matgen
from ilupp_amd import _native
dims = [int(v) for v in sys.argv[1].split(",")]
d, i, p = matgen.poisson3d(*dims)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
x = torch.ones(n, dtype=torch.float64, device=dev)
P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
for rep in range(3):
    x.fill_(1.0); torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        P.apply_device(x.data_ptr(), n, transpose=True, sync=True)
    except Exception as e:
        print(dims, "apply_trans FAILED:", e); break
    torch.cuda.synchronize(); print(dims, "apply_trans %.2f ms" % (1e3 * (time.perf_counter() - t0)), P.timings()["last_apply_ms"])
