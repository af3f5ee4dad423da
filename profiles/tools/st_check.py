#!/usr/bin/env python3
"""Static level-major path (st.hip) against the CPU oracle on 7-point / 5-point meshes: was it taken, is it bit-exact,
how long do the three sweeps take.  Usage: st_check.py [grid ...]   (ILUPP_DEBUG=1 shows the analysis verdicts)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scipy.sparse as sp
import matgen
import ilupp_amd as ilupp
from oracle import oracle as O

def check(name, d, i, p, csc=False):
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    if csc:
        A = A.tocsc()
    t0 = time.time()
    P = ilupp.ILU0Preconditioner(A)
    t1 = time.time()
    b = np.linspace(1.0, 2.0, n)
    x = b.copy(); P.apply(x)
    tm = P.pr.timings() if hasattr(P.pr, "timings") else {}
    L, U = P.factors()
    Ao = A if not csc else A
    Lo, Uo = O.orc().ilu0((Ao.data, Ao.indices, Ao.indptr, not csc))
    ok_i = np.array_equal(L.indices, Lo[1]) and np.array_equal(L.indptr, Lo[2]) and np.array_equal(U.indices, Uo[1]) and np.array_equal(U.indptr, Uo[2])
    ok_v = np.array_equal(L.data, Lo[0]) and np.array_equal(U.data, Uo[0])
    xo = O.orc().apply_lu(Lo, Uo, b, O.ID)
    ok_x = np.array_equal(x, xo)
    xt = b.copy(); P.apply_trans(xt)
    xto = O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE)
    ok_t = np.array_equal(xt, xto)
    print("%-22s n=%9d idx %s val %s apply %s trans %s  ctor %.1f ms  %s" % (name, n, ok_i, ok_v, ok_x, ok_t, 1e3 * (t1 - t0),
          {k: round(v, 3) for k, v in tm.items()}), flush=True)
    return ok_i and ok_v and ok_x and ok_t

def main():
    grids = [int(a) for a in sys.argv[1:]] or [8, 24, 40, 64]
    ok = True
    for g in grids:
        ok &= check("poisson3d_%d" % g, *matgen.poisson3d(g))
    ok &= check("poisson2d_200", *matgen.poisson2d(200))
    ok &= check("poisson3d_24_csc", *matgen.poisson3d(24), csc=True)
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)

main()
