# apply kernel times (L-type + U-type sweep) of factors with long rows:  python profiles/tools/apply_times.py
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import matgen
import ilupp_amd as ilupp

def show(name, P):
    x = np.ones(P.shape[0]); P.apply(x); x[:] = 1.0; P.apply(x)
    tm = P.pr.timings()
    print('%-46s apply %.2f + %.2f ms' % (name, tm['lsolve_kernel_ms'], tm['usolve_kernel_ms']), flush=True)

d, i, p = matgen.random_dd(1000000, 19, 25.0, 12345)
A = sp.csr_matrix((d, i, p), shape=(1000000, 1000000))
show('ILUT(10,1e-4) C3', ilupp.ILUTPreconditioner(A, fill_in=10, threshold=1e-4))
for g in (128, 256):
    d, i, p = matgen.poisson3d(g)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    show('ICholT(5,1e-3) poisson %d^3' % g, ilupp.ICholTPreconditioner(A, add_fill_in=5, threshold=1e-3))
