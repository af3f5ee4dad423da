import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import matgen, ilupp_amd as ilupp
from oracle import oracle as O
g=16
d,i,p = matgen.poisson3d(g)
n=p.shape[0]-1
A=sp.csr_matrix((d,i,p),shape=(n,n))
Lo,Uo = O.orc().ilu0((d,i,p,True))
for t in range(6):
    P=ilupp.ILU0Preconditioner(A)
    L,U=P.factors()
    bu = np.nonzero(U.data!=Uo[0])[0]; bl=np.nonzero(L.data!=Lo[0])[0]
    rows_u = np.searchsorted(U.indptr, bu, side='right')-1
    print('trial',t,'badU',len(bu),'badL',len(bl), 'first bad U rows', rows_u[:8], 'pos in row', (bu-U.indptr[rows_u])[:8])
    if len(bu):
        r=rows_u[0]; print('  row',r,'xyz',r%g,(r//g)%g,r//(g*g),'got',U.data[U.indptr[r]:U.indptr[r+1]],'want',Uo[0][U.indptr[r]:U.indptr[r+1]])
