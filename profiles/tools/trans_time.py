# apply_trans (transposed storages) and IChol(0) apply timings
import sys, time, numpy as np, torch, scipy.sparse as sp
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen
from ilupp_amd import _native
import ilupp_amd as ilupp
dev=torch.device('cuda',0)
g=int(sys.argv[1]) if len(sys.argv)>1 else 256
d,i,p = matgen.poisson3d(g)
n=p.shape[0]-1
td=torch.from_numpy(d).to(dev); ti=torch.from_numpy(i).to(dev); tp=torch.from_numpy(p).to(dev)
tx=torch.ones(n,dtype=torch.float64,device=dev)
P=_native.ILU0Preconditioner_device(td.data_ptr(),ti.data_ptr(),tp.data_ptr(),n,True)
for k in range(3):
    tx.fill_(1.0); torch.cuda.synchronize()
    t0=time.perf_counter(); P.apply_device(tx.data_ptr(), n, transpose=True, sync=True); t1=time.perf_counter()
    t=P.timings(); print('apply_trans %d^3: first sweep %.3f ms, second %.3f ms, wall %.1f ms, chk %.6f'%(g,t['lsolve_kernel_ms'],t['usolve_kernel_ms'],1e3*(t1-t0),float(tx.sum().item())))
if g <= 128:
    A = sp.csr_matrix((d,i,p),shape=(n,n))
    Pc = ilupp.IChol0Preconditioner(A)
    x = np.ones(n)
    for k in range(3):
        t0=time.perf_counter(); Pc.apply(x); t1=time.perf_counter()
    print('IChol0 apply (host vector) wall %.1f ms'%(1e3*(t1-t0)))
