# ILUT wave-parallel kernel: bit-exact comparison with the reference (oracle/_ref when present, else the C restatement)
# and timings.  usage: python profiles/tools/ilut_check.py [--grids g1,g2] [n_random ...]
import os, sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import matgen
import ilupp_amd as ilupp
from oracle import oracle as O
ref = O.ref() if O.ref_available() else O.orc()
kind = 'reference' if O.ref_available() else 'C restatement'

def t(f):
    t0 = time.perf_counter(); r = f(); return time.perf_counter() - t0, r

def eq(M, Mo):
    return (np.array_equal(M.indptr, Mo[2]) and np.array_equal(M.indices, Mo[1])
            and np.array_equal(M.data.view(np.int64), Mo[0].view(np.int64)))

def check(name, d, i, p, fill, tau, compare=True):
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    tg, P = t(lambda: ilupp.ILUTPreconditioner(A, fill_in=fill, threshold=tau))
    tg2, P = t(lambda: ilupp.ILUTPreconditioner(A, fill_in=fill, threshold=tau))
    L, U = P.factors()
    msg = '%s ILUT(%d,%g) n=%d nnz(L)=%d nnz(U)=%d: GPU %.4f s (2nd %.4f s)' % (name, fill, tau, n, L.nnz, U.nnz, tg, tg2)
    if compare:
        tc, (Lo, Uo) = t(lambda: ref.ilut((d, i, p, True), fill, tau))
        msg += '   %s (1 core) %.3f s   bit-exact=%s' % (kind, tc, eq(L, Lo) and eq(U, Uo))
    print(msg, flush=True)

grids = (8, 16, 32)
argv = sys.argv[1:]
if argv and argv[0] == '--grids':
    grids = tuple(int(x) for x in argv[1].split(',') if x); argv = argv[2:]
args = [int(s) for s in argv] or [2000, 50000]
for g in grids:
    d, i, p = matgen.poisson3d(g)
    for fill, tau in ((10, 1e-4), (5, 0.1), (100, 0.0)):
        if fill == 100 and g > 16: continue
        check('poisson %d^3' % g, d, i, p, fill, tau)
for n in args:
    d, i, p = matgen.random_dd(n, 19, 25.0, 12345)
    check('random_dd', d, i, p, 10, 1e-4, compare=n <= 1000000)
    if n <= 50000: check('random_dd', d, i, p, 5, 0.1)
