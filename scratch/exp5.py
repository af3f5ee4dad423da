import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen, ilupp_amd as ilupp
def run(name, ctor, A, **kw):
    t0=time.perf_counter(); P=ctor(A, **kw); t1=time.perf_counter()
    x=np.ones(A.shape[0]); P.apply(x); t2=time.perf_counter()
    t=P.pr.timings()
    print('%-34s n=%8d nnz=%9d  ctor %.3f s (numeric kernel %.1f ms)  apply %.3f s (kernels %.2f+%.2f ms) total_nnz=%d'%(name,A.shape[0],A.nnz,t1-t0,t['numeric_kernel_ms'],t2-t1,t['lsolve_kernel_ms'],t['usolve_kernel_ms'],P.total_nnz), flush=True)
for n in (50000, 200000, 1000000):
    d,i,p=matgen.random_dd(n,19,25.0,12345); A=sp.csr_matrix((d,i,p),shape=(n,n))
    run('ILUT(10,1e-4) random_dd', ilupp.ILUTPreconditioner, A, fill_in=10, threshold=1e-4)
for g in (64, 128):
    d,i,p=matgen.poisson3d(g); n=g**3; A=sp.csr_matrix((d,i,p),shape=(n,n))
    run('IChol0 poisson3d %d'%g, ilupp.IChol0Preconditioner, A)
    run('ICholT(0,0) poisson3d %d'%g, ilupp.ICholTPreconditioner, A)
    if g==64: run('ILUT(10,1e-4) poisson3d %d'%g, ilupp.ILUTPreconditioner, A, fill_in=10, threshold=1e-4)
