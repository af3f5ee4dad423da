import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import matgen
import ilupp_amd as ilupp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d, i, p = matgen.random_dd(n, 19, 25.0, 12345)
A = sp.csr_matrix((d, i, p), shape=(n, n))
t0 = time.perf_counter()
P = ilupp.ILUTPreconditioner(A, fill_in=10, threshold=1e-4)
print('ILUT C3 n=%d: %.3f s, nnz %d' % (n, time.perf_counter() - t0, P.total_nnz), flush=True)
