import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import matgen, ilupp_amd as ilupp
for (nx,ny) in ((256,256),(256,64),(64,256),(256,1024)):
    d,i,p = matgen.poisson2d(nx,ny)
    n=p.shape[0]-1
    A=sp.csr_matrix((d,i,p),shape=(n,n))
    P=ilupp.ILU0Preconditioner(A)
    x=np.ones(n)
    for _ in range(3): P.apply(x)
    t=P.pr.timings()
    steps=nx+ny-1
    print(nx,ny,'n',n,t, 'per-step us: numeric %.2f lsolve %.2f usolve %.2f'%(1e3*t['numeric_kernel_ms']/steps,1e3*t['lsolve_kernel_ms']/steps,1e3*t['usolve_kernel_ms']/steps))
