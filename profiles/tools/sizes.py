# ILU(0) construction + apply over sizes (GPU wall time from device-resident arrays, median of 5), incl. BASELINE config C1 (2-D 200x200)
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, matgen
from ilupp_amd import _native
dev = torch.device('cuda', 0)
def run(name, d, i, p):
    n, nnz = p.shape[0] - 1, int(p[-1])
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    ts = []
    for rep in range(7):
        tx = torch.ones(n, dtype=torch.float64, device=dev); torch.cuda.synchronize()
        t0 = time.perf_counter()
        P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
        P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
        ts.append(time.perf_counter() - t0)
        tm = P.timings(); P = None
    t = float(np.median(ts[2:]))
    print('%-22s n=%9d nnz=%10d  %.3f ms  %.3e nnz/s   (analysis %.2f, numeric %.2f, apply %.2f ms)' % (name, n, nnz, 1e3 * t, nnz / t, tm['analysis_ms'], tm['numeric_ms'], tm['last_apply_ms']), flush=True)
run('C1: 2-D 200x200', *matgen.poisson2d(200))
run('2-D 2048x2048', *matgen.poisson2d(2048))
for g in (64, 128, 192, 256):
    run('3-D %d^3' % g, *matgen.poisson3d(g))
