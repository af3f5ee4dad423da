import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen, ilupp_amd as ilupp
for (gx,gy,gz) in ((256,16,16),(256,32,32),(256,64,64),(256,128,128),(256,256,64),(256,16,256),(256,256,16)):
    d,i,p = matgen.poisson3d(gx,gy,gz)
    n=p.shape[0]-1
    A=sp.csr_matrix((d,i,p),shape=(n,n))
    P=ilupp.ILU0Preconditioner(A)
    x=np.ones(n)
    for _ in range(3): P.apply(x)
    t=P.pr.timings()
    steps=gx+gy+gz-2
    print(gx,gy,gz,'tiles',(gy//16)*(gz//16),'steps',steps,'ms: num %.3f L %.3f U %.3f | per-step us: num %.2f L %.2f U %.2f'%(t['numeric_kernel_ms'],t['lsolve_kernel_ms'],t['usolve_kernel_ms'],1e3*t['numeric_kernel_ms']/steps,1e3*t['lsolve_kernel_ms']/steps,1e3*t['usolve_kernel_ms']/steps))
