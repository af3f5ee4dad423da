"""engine vs oracle for the multilevel factorisation WITH pivoting (partialILUCDP): python profiles/tools/dp_check.py [--big]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import ilupp_amd as ilupp
from ilupp_amd import _native
from oracle import oracle as O
import ml_cases as C

bad = 0
mats = dict(C.matrices())
big = "--big" in sys.argv
if big:
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    mats = {"rand20k": (sp.random(20000, 20000, 5.0 / 20000, random_state=rng) + sp.eye(20000) * 2).tocsr(), "lap100": C.laplace2d_matrix(10000)}
for mname, A in mats.items():
    A = A.tocsr(); A.sort_indices()
    for pname, thr, pre, knobs in C.PIVOT_PARAMS:
        t0 = time.time()
        try:
            o = O.orc().ml((A.data, A.indices, A.indptr, True), C.oracle_params(O, thr, pre, knobs))
        except O.OracleError as e:
            o = None
        t1 = time.time()
        try:
            P = _native.MultilevelILUCDPPreconditioner(A.data, A.indices, A.indptr, True, C.engine_params(ilupp, thr, pre, knobs))
        except Exception as e:
            if o is None:
                print(mname, pname, "both refuse"); continue
            bad += 1; print(mname, pname, "ENGINE FAILED", repr(e)[:300]); continue
        t2 = time.time()
        if o is None:
            bad += 1; print(mname, pname, "oracle refused, engine did not"); continue
        ok = o.levels() == P.levels()
        why = "" if ok else "levels %d vs %d" % (o.levels(), P.levels())
        if ok:
            for k in range(o.levels()):
                a, b = C.level_arrays(o.level(k)), C.level_arrays(P.level(k))
                for q, (x, y) in enumerate(zip(a, b)):
                    if not np.array_equal(x, y, equal_nan=(np.asarray(x).dtype.kind == "f")):
                        ok = False; why = "level %d array %d" % (k, q); break
                if not ok: break
        if ok:
            x = C.rhs(A.shape[0])
            y = x.copy(); P.apply(y)
            if not np.array_equal(o.apply(x), y, equal_nan=True):
                ok = False; why = "apply"
        if not ok: bad += 1
        if not ok or big or "-v" in sys.argv:
            print(mname, pname, "ok" if ok else "MISMATCH " + why, "levels", P.levels(), "oracle %.2fs engine %.2fs" % (t1 - t0, t2 - t1), flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
