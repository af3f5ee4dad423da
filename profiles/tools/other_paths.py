# timings of the other section-8(a) paths (ILUT, IChol0, ICholT) next to the reference's C++ on one host core:
# wall time of the whole constructor from host arrays (H2D included) and the dominant kernel alone.
#   python profiles/tools/other_paths.py [--full]      (--full: BASELINE configs C3 and C4 at their full sizes)
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import matgen
import ilupp_amd as ilupp
from oracle import oracle as O
ref = O.ref() if O.ref_available() else O.orc()
kind = 'reference' if O.ref_available() else 'C restatement'
full = '--full' in sys.argv

def t(f):
    t0 = time.perf_counter(); r = f(); return time.perf_counter() - t0, r

def kms(P):
    return P.pr.timings()['numeric_kernel_ms']

def line(name, tg, P, tc=None):
    x = np.ones(P.shape[0]); P.apply(x); x[:] = 1.0; P.apply(x)          # second apply: transposed storages / records exist
    tm = P.pr.timings()
    s = '%-58s GPU %.3f s (kernel %.1f ms; apply %.2f + %.2f ms)' % (name, tg, kms(P), tm['lsolve_kernel_ms'], tm['usolve_kernel_ms'])
    if tc is not None:
        s += '   %s, 1 core: %.3f s  -> x%.1f' % (kind, tc, tc / tg)
    print(s, flush=True)

n = 1000000 if full else 200000
d, i, p = matgen.random_dd(n, 19, 25.0, 12345)
A = sp.csr_matrix((d, i, p), shape=(n, n))
ilupp.ILUTPreconditioner(sp.identity(8, format='csr') * 2.0, fill_in=2, threshold=0.1)      # warm-up (library load)
tg, P = t(lambda: ilupp.ILUTPreconditioner(A, fill_in=10, threshold=1e-4))
tc, _ = t(lambda: ref.ilut((d, i, p, True), 10, 1e-4))
line('ILUT(10, 1e-4) random diag-dominant n=%d nnz=%d (C3)' % (n, p[-1]), tg, P, tc)
for g in ((128, 256) if full else (64, 96)):
    d, i, p = matgen.poisson3d(g)
    n3 = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n3, n3))
    cmp_ref = g <= 128
    tg, P = t(lambda: ilupp.IChol0Preconditioner(A))
    tc = t(lambda: ref.ichol0((d, i, p, True)))[0] if cmp_ref else None
    line('IChol0 poisson %d^3' % g, tg, P, tc)
    for a, tau in ((0, 0.0), (5, 1e-3)):
        tg, P = t(lambda: ilupp.ICholTPreconditioner(A, add_fill_in=a, threshold=tau))
        tc = t(lambda: ref.icholt((d, i, p, True), a, tau))[0] if (cmp_ref or a == 0) else None
        line('ICholT(%d, %g) poisson %d^3%s' % (a, tau, g, ' (C4)' if g == 256 and a == 0 else ''), tg, P, tc)
    if g <= 128:
        tg, P = t(lambda: ilupp.ILUTPreconditioner(A, fill_in=10, threshold=1e-4))
        tc, _ = t(lambda: ref.ilut((d, i, p, True), 10, 1e-4))
        line('ILUT(10, 1e-4) poisson %d^3' % g, tg, P, tc)
