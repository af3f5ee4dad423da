#!/usr/bin/env python3
"""Direct-feed static factor kernel (st_direct.hip) against the CPU oracle: which path ran, is it bit-exact (factors, apply,
apply_trans, a re-factorisation with new values), and the phase times of the 256^3 headline config.
Usage: sd_check.py [--big]   (ILUPP_DEBUG=1 shows the analysis verdicts; ILUPP_NO_DIRECT=1 the records path)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
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
    L, U = P.factors()
    Lo, Uo = O.orc().ilu0((A.data, A.indices, A.indptr, not csc))
    ok_i = np.array_equal(L.indices, Lo[1]) and np.array_equal(L.indptr, Lo[2]) and np.array_equal(U.indices, Uo[1]) and np.array_equal(U.indptr, Uo[2])
    ok_v = np.array_equal(L.data, Lo[0], equal_nan=True) and np.array_equal(U.data, Uo[0], equal_nan=True)
    ok_x = np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, O.ID), equal_nan=True)
    xt = b.copy(); P.apply_trans(xt)
    ok_t = np.array_equal(xt, O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE), equal_nan=True)
    print("%-26s n=%9d %-24s idx %s val %s apply %s trans %s  ctor %.1f ms" % (name, n, P.pr.path(), ok_i, ok_v, ok_x, ok_t, 1e3 * (t1 - t0)), flush=True)
    if not ok_v:
        bad = np.flatnonzero(~((U.data == Uo[0]) | (np.isnan(U.data) & np.isnan(Uo[0]))))
        badl = np.flatnonzero(~((L.data == Lo[0]) | (np.isnan(L.data) & np.isnan(Lo[0]))))
        print("   U mismatches %d (first at %s), L mismatches %d (first at %s)" % (bad.size, bad[:5], badl.size, badl[:5]))
    return ok_i and ok_v and ok_x and ok_t


def main():
    ok = True
    rng = np.random.default_rng(3)
    for shape in ((8, 8, 8), (24, 24, 24), (70, 45, 37), (64, 24, 16), (17, 33, 65), (40, 40, 40), (100, 20, 300)):
        d, i, p = matgen.poisson3d(*shape)
        d = d * (1.0 + 0.3 * rng.random(d.shape[0]))          # nonsymmetric values
        ok &= check("7pt %s" % (shape,), d, i, p)
    d, i, p = matgen.poisson3d(33, 20, 50)
    ok &= check("7pt csc", d * (1.0 + 0.3 * rng.random(d.shape[0])), i, p, csc=True)
    for shape in ((300, 300), (64, 1000), (1000, 64)):
        d, i, p = matgen.poisson2d(*shape)
        ok &= check("5pt %s" % (shape,), d * (1.0 + 0.3 * rng.random(d.shape[0])), i, p)
    # marker payloads and NaNs in the input
    d, i, p = matgen.poisson3d(24, 24, 24)
    d = d * (1.0 + 0.3 * rng.random(d.shape[0]))
    dv = d.view(np.uint64).copy()
    dv[1000] = 0x7FF85EEDC0DE0002; dv[5001] = 0x7FF85EEDC0DE0001; dv[20000] = 0x7FF8000000000000
    ok &= check("7pt markers", dv.view(np.float64), i, p)
    print("ALL OK" if ok else "FAILURES")
    if "--big" in sys.argv:
        import torch
        from ilupp_amd import _native
        d, i, p = matgen.poisson3d(256)
        n = p.shape[0] - 1
        dev = torch.device("cuda:0")
        td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
            x = torch.ones(n, dtype=torch.float64, device=dev)
            P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
            t1 = time.time()
            print("256^3 %s step %.3f ms %s" % (P.path(), 1e3 * (t1 - t0), {k: round(v, 3) for k, v in P.timings().items()}), flush=True)
        print("checksum", float(x.sum().item()))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
