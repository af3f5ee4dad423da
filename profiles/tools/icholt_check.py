# ICholT dataflow kernel: bit-exact comparison with the reference (oracle/_ref when present, else the C restatement)
# and timings.  usage: python profiles/tools/icholt_check.py [grid sizes...]
import os, sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import matgen
import ilupp_amd as ilupp
from oracle import oracle as O
ref = O.ref() if O.ref_available() else O.orc()
kind = 'reference' if O.ref_available() else 'C restatement'

def t(f):
    t0 = time.perf_counter(); r = f(); return time.perf_counter() - t0, r

def check(name, d, i, p, a, tau, compare=True):
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    tg, P = t(lambda: ilupp.ICholTPreconditioner(A, add_fill_in=a, threshold=tau))
    tg2, P = t(lambda: ilupp.ICholTPreconditioner(A, add_fill_in=a, threshold=tau))
    L, = P.factors()
    k2 = P.pr.timings()['numeric_kernel_ms']
    tg3, P = t(lambda: ilupp.ICholTPreconditioner(A, add_fill_in=a, threshold=tau))
    k3 = P.pr.timings()['numeric_kernel_ms']
    msg = '%s ICholT(%d,%g) n=%d nnz(L)=%d: GPU %.4f s (2nd %.4f s, kernel %.1f ms; 3rd %.4f s, kernel %.1f ms)' % (name, a, tau, n, L.nnz, tg, tg2, k2, tg3, k3)
    if compare:
        tc, Lo = t(lambda: ref.icholt((d, i, p, True), a, tau))
        ok = (np.array_equal(L.indptr, Lo[2]) and np.array_equal(L.indices, Lo[1])
              and np.array_equal(L.data.view(np.int64), Lo[0].view(np.int64)))
        msg += '   %s (1 core) %.3f s   bit-exact=%s' % (kind, tc, ok)
    print(msg, flush=True)

sizes = [int(s) for s in sys.argv[1:]] or [8, 16, 32]
for g in sizes:
    d, i, p = matgen.poisson3d(g)
    for a, tau in ((0, 0.0), (5, 1e-3), (2, 0.05)):
        check('poisson %d^3' % g, d, i, p, a, tau, compare=g <= 128)
