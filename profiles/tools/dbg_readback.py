import sys, time, faulthandler
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.sparse as sp
import matgen, ilupp_amd as ilupp
d, i, p = matgen.poisson3d(24)
n = p.shape[0] - 1
A = sp.csr_matrix((d, i, p), shape=(n, n))
t0 = time.time()
P = ilupp.ILU0Preconditioner(A)
print('create ok', time.time() - t0, flush=True)
x = np.ones(n); P.apply(x)
print('apply ok', x[:3], flush=True)
