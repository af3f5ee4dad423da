# timeline summary for many tiles (needs -DILUPP_TIMELINE build via EXPLIB)
import sys, os, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen
from ilupp_amd import _native
if os.environ.get('EXPLIB'): _native._LIB_PATH=os.path.abspath(os.environ['EXPLIB'])
dev=torch.device('cuda',0)
gx,gy,gz=[int(v) for v in (sys.argv[1] if len(sys.argv)>1 else '256x256x256').split('x')]
d,i,p = matgen.poisson3d(gx,gy,gz)
n=p.shape[0]-1
td=torch.from_numpy(d).to(dev); ti=torch.from_numpy(i).to(dev); tp=torch.from_numpy(p).to(dev)
tx=torch.ones(n,dtype=torch.float64,device=dev)
P=_native.ILU0Preconditioner_device(td.data_ptr(),ti.data_ptr(),tp.data_ptr(),n,True)
for _ in range(3):
    tx.fill_(1.0); torch.cuda.synchronize()
    P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
t=P.timings(); print('L %.3f U %.3f ms'%(t['lsolve_kernel_ms'],t['usolve_kernel_ms']))
a=np.fromfile('/tmp/timeline_0.bin',dtype=np.uint64).reshape(-1,8).astype(np.float64)
t0=a[:,0].min()
a=(a-t0)/100.0
NY=gy//16; NZ=gz//16
np.set_printoptions(linewidth=250,precision=0,suppress=True)
first=a[:,1].reshape(NZ,NY); last=a[:,2].reshape(NZ,NY); mid=a[:,7].reshape(NZ,NY)
l255f=a[:,3].reshape(NZ,NY); l255l=a[:,4].reshape(NZ,NY)
print('entry time max %.1f'%a[:,0].max())
print('lane0 first-row time [us] (rows tz, cols ty):'); print(first)
print('lane0 rate first half / second half [us/row]:')
np.set_printoptions(linewidth=250,precision=2,suppress=True)
print((mid-first)/(gx//2)); print((last-mid)/(gx-1-gx//2))
print('lane255 last: max %.1f'%l255l.max())
print('startup skew lane0->lane255 first [us]:'); print(l255f-first)
