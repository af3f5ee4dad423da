import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import matgen, golden_util as G
from ilupp_amd import _native
from oracle import oracle as O
ref = O.ref()
for name, (d, i, p) in (("mesh 48^3", matgen.poisson3d(48)), ("random 2e5 k=9", matgen.random_dd(200000, k=9))):
    n = p.shape[0] - 1
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    for (fill, tau) in ((100, 0.1), (60, 0.1), (100, 1e-3)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P = _native.ILUTPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, fill, tau)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        Lo, Uo = ref.ilut((d, i, p, True), fill, tau); t2 = time.perf_counter()
        F = P.factors_info()
        ok = G.mat_equal(tuple(F[0][:4]), Lo) and G.mat_equal(tuple(F[1][:4]), Uo)
        print("%s ILUT(%d, %g): GPU %.1f ms  reference %.1f ms  equal %s  nnz %d" % (name, fill, tau, 1e3*(t1-t0), 1e3*(t2-t1), ok, F[0][0].shape[0]+F[1][0].shape[0]), flush=True)
