import sys, ctypes, numpy as np, scipy.sparse as sp
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen, ilupp_amd as ilupp
from ilupp_amd import _native
L=_native.lib()
L.ilupp_hip_debug_ctrl.argtypes=[ctypes.c_void_p, ctypes.c_void_p]
for shape in ((256,16,16),(256,256,1)):
    gx,gy,gz=shape
    d,i,p = matgen.poisson3d(gx,gy,gz) if gz>1 else matgen.poisson2d(gx,gy)
    n=p.shape[0]-1
    A=sp.csr_matrix((d,i,p),shape=(n,n))
    P=ilupp.ILU0Preconditioner(A)
    x=np.ones(n)
    for _ in range(2): P.apply(x)
    t=P.pr.timings()
    out=np.zeros(16,dtype=np.int32)
    L.ilupp_hip_debug_ctrl(P.pr._h, out.ctypes.data)
    print(shape,'L ms %.3f U ms %.3f'%(t['lsolve_kernel_ms'],t['usolve_kernel_ms']),'wdata',out[4:7],'iters',out[8:11],'wdep',out[11:14])
