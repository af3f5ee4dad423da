"""GPU check of the multilevel ILU++ preconditioner against the oracle, level by level (development tool; the tests proper are
tests/test_gpu_ml.py).   python profiles/tools/ml_check.py [--big]"""
import sys, os, time, faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "600")), exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp
import torch  # noqa: F401
import ilupp_amd as ilupp
from ilupp_amd import _native
from oracle import oracle as O
import matgen

def params(thr, pre=("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), **kw):
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(1)
    p.threshold = thr
    p.PREPROCESSING = ilupp.preprocessing_sequence(pre)
    for k, v in kw.items():
        setattr(p, k, v)
    return p

def oparams(p):
    names = {"NORMALIZE_COLUMNS": 1, "NORMALIZE_ROWS": 2, "PQ_ORDERING": 3}
    return O.ml_params(p.threshold, preprocessing=tuple(names[s] for s in p.PREPROCESSING), max_levels=p.MAX_LEVELS, min_pivot=p.MIN_PIVOT,
                       threshold_shift_schur=p.THRESHOLD_SHIFT_SCHUR, min_elim_factor=p.MIN_ELIM_FACTOR, pq_threshold=p.PQ_THRESHOLD)

def check(A, p, name, verbose=True):
    A = A.copy(); A.sort_indices()
    A.indices = A.indices.astype(np.int32); A.indptr = A.indptr.astype(np.int32)
    a = (A.data, A.indices, A.indptr, sp.isspmatrix_csr(A))
    t0 = time.time(); C = O.orc().ml(a, oparams(p)); t1 = time.time()
    try:
        G = _native.MultilevelILUCDPPreconditioner(A.data, A.indices, A.indptr, sp.isspmatrix_csr(A), p)
    except Exception as e:
        print(name, "GPU FAILED:", e); return False
    t2 = time.time()
    ok = G.levels() == C.levels() and G.total_nnz == C.total_nnz()
    bad = []
    for k in range(min(G.levels(), C.levels())):
        lg, lc = G.level(k), C.level(k)
        for key in ("n", "D", "perm_rows", "perm_cols", "inv_perm_rows", "inv_perm_cols", "D_l", "D_r"):
            same = np.array_equal(lg[key], lc[key], equal_nan=True) if isinstance(lc[key], np.ndarray) else lg[key] == lc[key]
            if not same: bad.append((k, key))
        for key in ("L", "U"):
            for q in range(3):
                if not (lg[key][q].shape == lc[key][q].shape and np.array_equal(lg[key][q], lc[key][q], equal_nan=True)): bad.append((k, key, q))
    b = np.cos(np.arange(A.shape[0]) * 0.7) + 1.5
    x = b.copy(); G.apply(x); xt = b.copy(); G.apply_trans(xt)
    xa = np.array_equal(x, C.apply(b), equal_nan=True); xb = np.array_equal(xt, C.apply(b, O.TRANSPOSE), equal_nan=True)
    good = ok and not bad and xa and xb
    print(name, "levels", C.levels(), [C.level(k)["n"] for k in range(C.levels())][:8], "nnz", C.total_nnz(), "cpu %.2fs gpu %.2fs" % (t1 - t0, t2 - t1), G.timings(),
          "OK" if good else ("FAIL", G.levels(), G.total_nnz, bad[:6], xa, xb), flush=True)
    return good

allok = True
d, i, pp = matgen.poisson3d(6, 7, 5); A = sp.csr_matrix((d, i, pp))
for thr in (0.0, 0.01, 0.1, 1.0):
    allok &= check(A, params(thr), "p3d thr %g" % thr)
    allok &= check(A.tocsc(), params(thr), "p3d csc thr %g" % thr)
R = (sp.random(300, 300, density=0.03, random_state=np.random.default_rng(5), format='csr') + sp.eye(300) * 0.5).tocsr()
for thr in (0.0, 0.01, 0.1):
    allok &= check(R, params(thr), "rand thr %g" % thr)
    allok &= check(R.tocsc(), params(thr), "rand csc thr %g" % thr)
    allok &= check(R, params(thr, pre=()), "rand nopre thr %g" % thr)
    allok &= check(R, params(thr, pre=("NORMALIZE_COLUMNS", "NORMALIZE_ROWS")), "rand norm thr %g" % thr)
    allok &= check(R, params(thr, MAX_LEVELS=3), "rand maxlev3 thr %g" % thr)
    allok &= check(R, params(thr, THRESHOLD_SHIFT_SCHUR=1e-3, MIN_PIVOT=0.1), "rand shift thr %g" % thr)
if "--big" in sys.argv:
    for n, k in ((20000, 6), (100000, 8)):
        M = sp.csr_matrix(matgen.random_dd(n, k=k), shape=(n, n))
        for thr in (1e-3, 1e-4):
            allok &= check(M, params(thr), "rdd n=%d thr %g" % (n, thr))
    d, i, pp = matgen.poisson3d(40, 40, 40); A = sp.csr_matrix((d, i, pp))
    for thr in (0.05, 0.01):
        allok &= check(A, params(thr), "p3d 40^3 thr %g" % thr)
    d, i, pp = matgen.poisson3d(30, 30, 30); n = pp.shape[0] - 1
    A = (sp.csr_matrix((d, i, pp), shape=(n, n)) + 0.8 * sp.diags([np.ones(n - 1)], [1], shape=(n, n))).tocsr()
    allok &= check(A, params(0.005), "convdiff 30^3 thr 0.005")
    allok &= check(A.tocsc(), params(0.02), "convdiff 30^3 csc thr 0.02")
    R = (sp.random(2000, 2000, density=0.004, random_state=np.random.default_rng(7), format='csr') + sp.eye(2000) * 0.3).tocsr()
    allok &= check(R, params(0.05), "weak diagonal n=2000 thr 0.05 (26 levels)")
    M = sp.csr_matrix(matgen.random_dd(20000, k=8, diag=2.0), shape=(20000, 20000))
    allok &= check(M, params(1e-2), "rdd diag 2 n=20000 thr 0.01")
if "--huge" in sys.argv:
    M = sp.csr_matrix(matgen.random_dd(1000000, k=8), shape=(1000000, 1000000))
    allok &= check(M, params(1e-3), "rdd n=1e6 thr 1e-3")
print("ALL OK" if allok else "SOME FAILED")
