# timeline of the level-major factor kernel (needs a -DILUPP_TIMELINE build: ILUPP_HIP_LIBRARY=...)
import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen
from ilupp_amd import _native
dev=torch.device('cuda',0)
gx,gy,gz=[int(v) for v in (sys.argv[1] if len(sys.argv)>1 else '256x256x256').split('x')]
d,i,p = matgen.poisson3d(gx,gy,gz)
n=p.shape[0]-1
td=torch.from_numpy(d).to(dev); ti=torch.from_numpy(i).to(dev); tp=torch.from_numpy(p).to(dev)
for _ in range(2):
    P=_native.ILU0Preconditioner_device(td.data_ptr(),ti.data_ptr(),tp.data_ptr(),n,True)
t=P.timings(); print('factor kernel %.3f ms'%t['numeric_kernel_ms'])
a=np.fromfile('/tmp/timeline_factor.bin',dtype=np.uint64).reshape(-1,8).astype(np.float64)
t0=a[:,0].min(); a=(a-t0)/100.0
NY=gy//16; NZ=gz//16
np.set_printoptions(linewidth=250,precision=2,suppress=True)
w0f=a[:,1].reshape(NZ,NY); w0m=a[:,2].reshape(NZ,NY); w0l=a[:,3].reshape(NZ,NY)
print('wave0 first step time [us]'); print(np.round(w0f))
nch=gx+18
print('wave0 us/step first half, second half'); print((w0m-w0f)/(nch/2)); print((w0l-w0m)/(nch/2))
print('last end %.1f us'%a[:,6].max())
