# ILU(0) on a random (non-mesh) matrix: factor and apply times next to the reference
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import matgen
import ilupp_amd as ilupp
from oracle import oracle as O
ref = O.ref() if O.ref_available() else O.orc()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d, i, p = matgen.random_dd(n, 19, 25.0, 12345)
A = sp.csr_matrix((d, i, p), shape=(n, n))
for rep in range(2):
    t0 = time.perf_counter(); P = ilupp.ILU0Preconditioner(A); tg = time.perf_counter() - t0
x = np.ones(n); P.apply(x); x[:] = 1.0; P.apply(x)
tm = P.pr.timings()
t0 = time.perf_counter(); Lo, Uo = ref.ilu0((d, i, p, True)); tc = time.perf_counter() - t0
L, U = P.factors()
ok = np.array_equal(L.data, Lo[0]) and np.array_equal(U.data, Uo[0]) and np.array_equal(L.indices, Lo[1])
print('ILU0 random n=%d nnz=%d: GPU %.3f s (analysis %.1f ms, numeric %.1f ms, kernel %.1f ms; apply %.2f + %.2f ms)   reference %.3f s   bit-exact=%s'
      % (n, p[-1], tg, tm['analysis_ms'], tm['numeric_ms'], tm['numeric_kernel_ms'], tm['lsolve_kernel_ms'], tm['usolve_kernel_ms'], tc, ok))
