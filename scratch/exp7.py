# timeline of the L-solve tiles (needs a -DILUPP_TIMELINE build)
import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen
from ilupp_amd import _native
import os
if os.environ.get('EXPLIB'): _native._LIB_PATH=os.path.abspath(os.environ['EXPLIB'])
dev=torch.device('cuda',0)
gx,gy,gz=[int(v) for v in (sys.argv[1] if len(sys.argv)>1 else '256x64x64').split('x')]
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
a=(a-t0)/100.0   # us (100 MHz)
NY=gy//16
np.set_printoptions(linewidth=250,precision=1,suppress=True)
print('tiles %d (NY=%d); columns: entry, lane0 first, lane0 mid, lane0 last, lane255 first, lane255 last, lane15 first, lane240 first [us]'%(a.shape[0],NY))
for w in range(min(a.shape[0], 40)):
    print('wg %3d (ty %2d tz %2d)'%(w, w%NY, w//NY), a[w,[0,1,7,2,3,4,5,6]])
if a.shape[0]>40:
    for w in [a.shape[0]-NY-1, a.shape[0]-2, a.shape[0]-1]:
        print('wg %3d (ty %2d tz %2d)'%(w, w%NY, w//NY), a[w,[0,1,7,2,3,4,5,6]])
