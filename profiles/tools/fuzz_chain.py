#!/usr/bin/env python3
"""More seeds for the chain form of partialILUC (k_piluc_chain) and the rules that only run there (inverse-based, weighted dropping):
tests/fuzz_ml.py's cases FROM..TO, each built three times -- as dispatched, with ILUPP_PILUC_CHAIN=1 (the chain in LDS), and with
ILUPP_PILUC_CHAIN_MEM=1 on top (the chain with its vectors in global memory) -- and compared with the oracle array by array; then larger random matrices (n = 500 .. 4000, weak diagonals: several levels, working rows of hundreds of entries).
usage (GPU box): fuzz_chain.py FROM TO"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp
import torch  # noqa: F401
import fuzz_ml, ml_cases as C
import test_gpu_ml as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for chain in (0, 1, 2):
    if chain:
        os.environ["ILUPP_PILUC_CHAIN"] = "1"
    if chain == 2:
        os.environ["ILUPP_PILUC_CHAIN_MEM"] = "1"
    for seed in range(lo, hi):
        A, params = fuzz_ml.case(seed)
        try:
            T._against_oracle(A, params)
        except Exception as e:       # noqa: BLE001
            bad += 1
            print("FAIL seed %d flavour %d: %r" % (seed, chain, e), flush=True)
            traceback.print_exc(limit=2)
os.environ.pop("ILUPP_PILUC_CHAIN_MEM", None)
print("small cases %d..%d three ways: %d failures, %.0f s" % (lo, hi, bad, time.time() - t0), flush=True)
rng = np.random.default_rng(lo)
for it in range(24):
    n = int(rng.choice([500, 1200, 2500, 4000]))
    A = (sp.random(n, n, density=float(rng.choice([2.0, 4.0, 8.0])) / n, random_state=rng, format="csr") + sp.eye(n) * float(rng.choice([0.3, 0.6, 1.5]))).tocsr()
    thr = float(rng.choice([0.02, 0.05, 0.2]))
    knobs = {}
    r = int(rng.integers(0, 4))
    if r == 1:
        knobs.update(USE_INVERSE_DROPPING=True, USE_STANDARD_DROPPING=False)
    elif r == 2:
        knobs.update(USE_WEIGHTED_DROPPING=True, USE_STANDARD_DROPPING=False)
    elif r == 3:
        knobs.update(USE_INVERSE_DROPPING=True, USE_WEIGHTED_DROPPING2=True)
    if rng.random() < 0.3:
        knobs["fill_in"] = int(rng.choice([3, 10]))
    os.environ.pop("ILUPP_PILUC_CHAIN", None)
    if it % 2:
        os.environ["ILUPP_PILUC_CHAIN"] = "1"
    t1 = time.time()
    try:
        lv = T._against_oracle(A if it % 3 else A.tocsc(), (thr, T.PQ, knobs))
        print("larger %2d: n %4d thr %.2f knobs %s chain %d: %d levels ok (%.1f s)" % (it, n, thr, sorted(knobs), it % 2, lv, time.time() - t1), flush=True)
    except Exception as e:           # noqa: BLE001
        bad += 1
        print("FAIL larger %d n %d thr %g knobs %s chain %d: %r" % (it, n, thr, knobs, it % 2, e), flush=True)
print("total failures:", bad)
sys.exit(1 if bad else 0)
