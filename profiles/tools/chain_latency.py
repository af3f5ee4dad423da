import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import matgen
from ilupp_amd import _native
for n in (20000,):
    d, i, p = matgen.laplace1d(n)
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    for rep in range(2):
        P = _native.ILUCPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, 3, 0.0)
    print("iluc chain n=%d: kernel %.2f ms -> %.2f us per step" % (n, P.timings()["numeric_kernel_ms"], 1e3 * P.timings()["numeric_kernel_ms"] / n))
    for rep in range(2):
        P = _native.ICholTPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, 0, 0.0)
    print("icholt chain n=%d: kernel %.2f ms -> %.2f us per step" % (n, P.timings()["numeric_kernel_ms"], 1e3 * P.timings()["numeric_kernel_ms"] / n))
