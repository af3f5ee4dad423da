# step time (us per dependency level) of the sweeps versus the number of concurrently active tiles
import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen
from ilupp_amd import _native
import os
if os.environ.get('EXPLIB'): _native._LIB_PATH=os.path.abspath(os.environ['EXPLIB'])
dev=torch.device('cuda',0)
shapes=[(256,16,16),(256,32,32),(256,64,64),(256,128,128),(256,256,256),(512,128,128),(128,256,256)]
if len(sys.argv)>1: shapes=[tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (gx,gy,gz) in shapes:
    d,i,p = matgen.poisson3d(gx,gy,gz)
    n=p.shape[0]-1
    td=torch.from_numpy(d).to(dev); ti=torch.from_numpy(i).to(dev); tp=torch.from_numpy(p).to(dev)
    tx=torch.ones(n,dtype=torch.float64,device=dev)
    torch.cuda.synchronize()
    res=[]
    for _ in range(4):
        P=_native.ILU0Preconditioner_device(td.data_ptr(),ti.data_ptr(),tp.data_ptr(),n,True)
        tx.fill_(1.0); torch.cuda.synchronize()
        P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
        t=P.timings(); res.append((t['analysis_ms'],t['numeric_kernel_ms'],t['lsolve_kernel_ms'],t['usolve_kernel_ms']))
        P=None
    a=np.median(np.array(res[1:]),axis=0)
    steps=gx+gy+gz-2
    print('%4dx%4dx%4d n=%9d steps=%4d  analysis %.3f  numeric %.3f ms (%.2f us/step)  L %.3f ms (%.2f us/step)  U %.3f ms (%.2f us/step)  chk %.6f'
          %(gx,gy,gz,n,steps,a[0],a[1],1e3*a[1]/steps,a[2],1e3*a[2]/steps,a[3],1e3*a[3]/steps,float(tx.sum().item())))
