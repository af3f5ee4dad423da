import sys, ctypes, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen
from ilupp_amd import _native
L=_native.lib()
L.ilupp_hip_debug_ctrl.argtypes=[ctypes.c_void_p, ctypes.c_void_p]
g=256
d,i,p = matgen.poisson3d(g)
n=p.shape[0]-1
dev=torch.device('cuda',0)
td=torch.from_numpy(d).to(dev); ti=torch.from_numpy(i).to(dev); tp=torch.from_numpy(p).to(dev)
tx=torch.ones(n,dtype=torch.float64,device=dev)
torch.cuda.synchronize()
P=_native.ILU0Preconditioner_device(td.data_ptr(),ti.data_ptr(),tp.data_ptr(),n,True)
for _ in range(2):
    tx.fill_(1.0); torch.cuda.synchronize()
    P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
t=P.timings()
out=np.zeros(16,dtype=np.int32)
L.ilupp_hip_debug_ctrl(P._h, out.ctypes.data)
print('L ms %.3f U ms %.3f'%(t['lsolve_kernel_ms'],t['usolve_kernel_ms']),'iters(t0,t17,t255)',out[8:11],'wdata',out[11:14],'wdep(t17,t255)',out[14:16])
