# timings of the correctness-first paths (ILUT, IChol0, ICholT) next to the reference's C++ on one host core
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import matgen
import ilupp_amd as ilupp
from oracle import oracle as O
ref = O.ref() if O.ref_available() else O.orc()
kind = 'reference' if O.ref_available() else 'C restatement'

def t(f):
    t0 = time.perf_counter(); r = f(); return time.perf_counter() - t0, r

cases = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d, i, p = matgen.random_dd(n, 19, 25.0, 12345)
A = sp.csr_matrix((d, i, p), shape=(n, n))
tg, P = t(lambda: ilupp.ILUTPreconditioner(A, fill_in=10, threshold=1e-4))
tc, _ = t(lambda: ref.ilut((d, i, p, True), 10, 1e-4))
print('ILUT(fill_in=10, threshold=1e-4) random_dd n=%d nnz=%d: GPU %.3f s   %s (1 core) %.3f s' % (n, p[-1], tg, kind, tc))
for g in (64, 96):
    d, i, p = matgen.poisson3d(g)
    n3 = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n3, n3))
    tg, P = t(lambda: ilupp.IChol0Preconditioner(A))
    print('IChol0 poisson %d^3: GPU %.3f s' % (g, tg))
    tg, P = t(lambda: ilupp.ICholTPreconditioner(A, add_fill_in=2, threshold=1e-3))
    print('ICholT(add_fill_in=2, threshold=1e-3) poisson %d^3: GPU %.3f s' % (g, tg))
    tg, P = t(lambda: ilupp.ILUTPreconditioner(A, fill_in=10, threshold=1e-4))
    tc, _ = t(lambda: ref.ilut((d, i, p, True), 10, 1e-4))
    print('ILUT(fill_in=10, threshold=1e-4) poisson %d^3: GPU %.3f s   %s %.3f s' % (g, tg, kind, tc))
