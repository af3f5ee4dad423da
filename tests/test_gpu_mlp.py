"""GPU parity tests of the multilevel ILU++ preconditioner WITH pivoting (ilupp_amd/csrc/pilucdp.hip; reference partialILUCDP,
ILUCDP.hpp:268-1404 -- what the reference's default-constructed parameters, ILUppPreconditioner(A) and solve(A, b) run on).

Bit for bit, as everywhere in this package:
* the package's Python surface (LinearOperator protocol, `solve` with setter-built parameters) on the reference's vectors;
* the golden vectors of the REAL reference: tests/golden/ml.npz (default-constructed parameters and default_configuration(0, 1, 10, 11) on the
  reference's test matrices: 120 cases) and tests/golden/mlp.npz (tests/ml_cases.py PIVOT_PARAMS: windows of the row reordering and of the
  total pivoting, pivot tolerances, every FINAL_ROW_CRIT family, bounded fill, other dropping rules; CSR and CSC; up to 100 levels);
* the oracle on random matrices with random parameters, and on matrices of 10^4 rows.
"""
import hashlib
import os

import numpy as np
import pytest
import scipy.sparse as sp

import matgen
import ml_cases as C

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def _native_ml(a, params):
    from ilupp_amd import _native
    return _native.MultilevelILUCDPPreconditioner(a[0], a[1], a[2], a[3], params)


# ---- the package's Python surface on the reference's vectors ----
def _sp_matrix(z, key):
    a = (z[key + "/A_data"], z[key + "/A_indices"], z[key + "/A_indptr"])
    n = a[2].shape[0] - 1
    return (sp.csr_matrix if bool(z[key + "/A_is_csr"]) else sp.csc_matrix)(a, shape=(n, n))


@pytest.mark.parametrize("key", ["ml_laplace2d_csr", "ml_random_csc", "ml_p3d_6_7_5_csr"])
def test_operator_protocol_on_reference_vectors(key):
    """ILUppPreconditioner as a scipy LinearOperator (`@`, dot, apply in place, .T, total_nnz, memory), default-constructed parameters:
    the outputs the reference produced for the same matrix and right-hand side (tests/golden/ml.npz, config -1)"""
    import ilupp_amd as ilupp
    z = np.load(os.path.join(HERE, "golden", "ml.npz"))
    A = _sp_matrix(z, key)
    n = A.shape[0]
    cases = [c for c in C.ml_npz_cases(z) if c[0] == key and c[2] == -1 and (key + "/" + c[1] + "_info") in z.files]
    assert cases
    for _, tag, _, thr, fill in cases:
        kw = {"threshold": thr}
        if fill >= 0:
            kw["fill_in"] = fill
        P = ilupp.ILUppPreconditioner(A, **kw)
        name = "%s/%s" % (key, tag)
        assert P.total_nnz == int(z[name + "_info"][1]) and P.memory == 0.0 and P.shape == (n, n)
        b = C.rhs(n)
        want, want_t = z[name + "_apply"], z[name + "_apply_trans"]
        assert np.array_equal(P @ b, want, equal_nan=True) and np.array_equal(P.dot(b), want, equal_nan=True), name
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, want, equal_nan=True), name
        assert np.array_equal(P.T @ b, want_t, equal_nan=True), name


def test_solve_front_end_on_reference_vectors():
    """ilupp.solve(A, b, params=...) with a parameter object set up through the PREPROCESSING setters (the way callers of the reference
    do it) reproduces the reference's solution vectors of tests/golden/solve.npz (make_golden_solve.py)"""
    import ilupp_amd as ilupp
    z = np.load(os.path.join(HERE, "golden", "solve.npz"))
    key = "laplace2d_900_csr"
    a = (z[key + "/data"], z[key + "/indices"], z[key + "/indptr"])
    A, b = sp.csr_matrix(a, shape=(900, 900)), z[key + "/b"]
    for tag, setters in (("t0.01_pq", ("set_PQ",)), ("t0.01_mwm", ("set_MAX_WEIGHTED_MATCHING_ORDERING",)),
                         ("t0.01_mwm_spq", ("set_MAX_WEIGHTED_MATCHING_ORDERING_SYM_PQ",))):
        param = ilupp.iluplusplus_precond_parameter()
        param.default_configuration(1)
        for s_ in setters:
            getattr(param.PREPROCESSING, s_)()
        param.threshold = 1e-2
        x, info = ilupp.solve(A, b, atol=1e-8, rtol=1e-8, params=param, info=True)
        x_ref = z["%s/%s/x" % (key, tag)]
        assert abs(info[0] - int(z["%s/%s/info" % (key, tag)][1])) <= 1, tag
        assert np.linalg.norm(x - x_ref) <= 1e-6 * np.linalg.norm(x_ref), tag


# ---- golden vectors ----
def test_presets_of_the_reference():
    """tests/golden/ml.npz: ILUppPreconditioner(A, threshold, fill_in) of the reference with default-constructed parameters and
    default_configuration(0, 1, 10, 11), 120 cases: levels, total_nnz, apply, apply_trans"""
    import ilupp_amd as ilupp
    z = np.load(os.path.join(HERE, "golden", "ml.npz"))
    seen = refused = 0
    for key, tag, config, thr, fill in C.ml_npz_cases(z):
        a = (z[key + "/A_data"], z[key + "/A_indices"], z[key + "/A_indptr"], bool(z[key + "/A_is_csr"]))
        n = a[2].shape[0] - 1
        name = "%s/%s" % (key, tag)
        try:
            P = _native_ml(a, C.ml_npz_params(ilupp, config, thr, fill))
        except NotImplementedError as e:
            assert config == 11 and "undefined" in str(e), name
            refused += 1
            continue
        assert (P.levels(), P.total_nnz) == tuple(int(v) for v in z[name + "_info"]), name
        b = C.rhs(n)
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, z[name + "_apply"], equal_nan=True), name
        x = b.copy(); P.apply_trans(x)
        assert np.array_equal(x, z[name + "_apply_trans"], equal_nan=True), name
        seen += 1
    assert seen + refused == 120 and seen >= 100


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("name", [n for n, _ in C.matrices()])
def test_reference_vectors(name, fmt):
    import ilupp_amd as ilupp
    gold = np.load(os.path.join(HERE, "golden", "mlp.npz"))
    key = "%s_%s" % (name, fmt)
    a = (gold[key + "/data"], gold[key + "/indices"], gold[key + "/indptr"], fmt == "csr")
    b = C.rhs(a[2].shape[0] - 1)
    for tag, thr, pre, knobs in C.PIVOT_PARAMS:
        k2 = "%s/%s" % (key, tag)
        P = _native_ml(a, C.engine_params(ilupp, thr, pre, knobs))
        info = gold[k2 + "/info"]
        assert P.levels() == info[0] and P.total_nnz == info[1], k2
        for k in range(P.levels()):
            assert np.array_equal(_digest(C.level_arrays(P.level(k))), gold[k2 + "/levels_sha"][k]), (k2, k)
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, gold[k2 + "/apply"], equal_nan=True), k2
        x = b.copy(); P.apply_trans(x)
        assert np.array_equal(x, gold[k2 + "/apply_trans"], equal_nan=True), k2


# ---- the oracle ----
def _same(P, Q, n, what):
    assert P.levels() == Q.levels() and P.total_nnz == Q.total_nnz(), what
    for k in range(Q.levels()):
        for q, (x, y) in enumerate(zip(C.level_arrays(P.level(k)), C.level_arrays(Q.level(k)))):
            assert np.array_equal(x, y, equal_nan=(np.asarray(x).dtype.kind == "f")), (what, k, q)
    b = C.rhs(n)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, Q.apply(b), equal_nan=True), what
    from oracle import oracle as O
    x = b.copy(); P.apply_trans(x)
    assert np.array_equal(x, Q.apply(b, O.TRANSPOSE), equal_nan=True), what


def test_fuzz_against_the_oracle():
    import ilupp_amd as ilupp
    from oracle import oracle as O
    rng = np.random.default_rng(77 + int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")))          # other seeds: profiles/tools/fuzz_more.sh
    names = {v: k for k, v in C.PRE.items()}
    for it in range(60):
        n = int(rng.integers(3, 400))
        A = (sp.random(n, n, min(1.0, rng.uniform(2, 8) / n), random_state=rng, data_rvs=lambda k: rng.standard_normal(k))
             + sp.eye(n) * float(rng.choice([0.0, 0.5, 3.0]))).asformat("csr" if it % 2 else "csc")
        A.sort_indices()
        a = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), bool(it % 2))
        knobs = C.pivoting(piv_tol=float(rng.choice([1.0, 0.5, 0.1, 0.0])), PERMUTE_ROWS=int(rng.integers(0, 4)), TOTAL_PIV=int(rng.integers(0, 3)),
                           BEGIN_TOTAL_PIV=bool(rng.integers(0, 2)), FINAL_ROW_CRIT=int(rng.integers(-1, 10)),
                           MIN_ELIM_FACTOR=float(rng.choice([0.0, 0.3, 0.5])), SMALL_PIVOT_TERMINATES=bool(rng.integers(0, 2)),
                           MOVE_LEVEL_FACTOR=float(rng.choice([0.5, 2.0])))
        if rng.integers(0, 3) == 0:
            knobs["fill_in"] = int(rng.integers(1, 8))
        if rng.integers(0, 3) == 0:
            knobs.update(USE_STANDARD_DROPPING=bool(rng.integers(0, 2)), USE_PIVOT_DROPPING=bool(rng.integers(0, 2)), USE_ERR_PROP_DROPPING2=bool(rng.integers(0, 2)),
                         COMBINE_FACTOR=int(rng.integers(0, 4)))
        if rng.integers(0, 5) == 0:                                  # weighted dropping (accumulated weights), alone or beside the rules above
            knobs.update(USE_WEIGHTED_DROPPING=bool(rng.integers(0, 2)), USE_WEIGHTED_DROPPING2=bool(rng.integers(0, 2)),
                         WEIGHT_WEIGHTED_DROP=float(rng.choice([1.0, 0.2])), INIT_WEIGHTS_LU=float(rng.choice([1.0, 0.5, 2.0])))
        if rng.integers(0, 4) == 0:                                  # inverse-based dropping, alone or beside the rules above
            knobs.update(USE_INVERSE_DROPPING=True, USE_ERR_PROP_DROPPING=bool(rng.integers(0, 2)), WEIGHT_INVERSE_DROP=float(rng.choice([1.0, 0.3, 3.0])))
        pre = [("PQ_ORDERING",), ("MAX_WEIGHTED_MATCHING_ORDERING",), ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), ("SPARSE_FIRST_ORDERING",), ()][int(rng.integers(0, 5))]
        thr = float(rng.choice([0.0, 1e-3, 1e-2, 0.1, 1.0]))
        ep = C.engine_params(ilupp, thr, pre, knobs)
        if ep._uses_partial_iluc():
            continue
        Q = O.orc().ml(a, C.oracle_params(O, thr, pre, knobs))
        _same(_native_ml(a, ep), Q, n, (it, n, pre, thr, knobs))


def test_larger():
    """10^4 rows: thousands of sequential steps, stores that are enlarged, a second level from the fill criterion"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    n = 10000
    for A, thr in (((sp.random(n, n, 5.0 / n, random_state=rng) + sp.eye(n) * 2).tocsr(), 1e-2), (C.laplace2d_matrix(n), 1e-2),
                   (sp.csr_matrix(matgen.random_dd(n, k=7, diag=1.5), shape=(n, n)), 0.1)):
        A.sort_indices()
        a = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True)
        p = ilupp.iluplusplus_precond_parameter()          # default-constructed: PQ + the pivoting factorisation
        p.threshold = thr
        Q = O.orc().ml(a, C.block_to_oracle(O, p._to_ml_params()))
        _same(_native_ml(a, p), Q, n, (n, thr))
        p.default_configuration(10)                        # BASELINE config 5: maximum weighted matching + the pivoting factorisation
        p.threshold = thr
        Q = O.orc().ml(a, C.block_to_oracle(O, p._to_ml_params()))
        _same(_native_ml(a, p), Q, n, (n, thr, "config 10"))


def test_config10_full_size():
    """BASELINE config 5 AS NAMED at its full size: default_configuration(10) -- maximum weighted matching + the factorisation with pivoting
    (parameters_implementation.h:547-550) -- on the n = 10^6 unsymmetric matrix of the bench, against the REAL reference where oracle/_ref
    travelled with the snapshot (else against the oracle's restatement): levels, every level's arrays, apply and apply_trans bit for bit.
    (A chain of 10^6 sequential steps: ~20 s on the GPU, ~7-10 s on the host core beside it.)"""
    import ilupp_amd as ilupp
    from oracle import oracle as O
    n = 1000000
    d, i, p_ = matgen.random_dd(n, 8, 25.0, 12345)
    a = (d, i, p_, True)
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(10)
    p.threshold = 1e-3
    lib = O.ref() if O.ref_available() else O.orc()
    Q = lib.ml(a, C.block_to_oracle(O, p._to_ml_params()))
    _same(_native_ml(a, p), Q, n, ("config 10, n = 1e6", "reference" if O.ref_available() else "oracle"))


def test_stores_that_fill_up():
    """the kernel stops between two steps when a store has no room for another row, the store is enlarged and the kernel goes on with that
    step: with stores of a few hundred entries (ILUPP_DP_STORE) that happens dozens of times -- same bits"""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np, scipy.sparse as sp
sys.path[:0] = [%r, %r]
import ml_cases as C
import ilupp_amd as ilupp
from ilupp_amd import _native
from oracle import oracle as O
for name, A in C.matrices():
    if name not in ("laplace2d_400", "weak_200", "rdd_300"):
        continue
    A = A.tocsr(); A.sort_indices()
    a = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True)
    for tag, thr, pre, knobs in C.PIVOT_PARAMS[:6]:
        Q = O.orc().ml(a, C.oracle_params(O, thr, pre, knobs))
        P = _native.MultilevelILUCDPPreconditioner(a[0], a[1], a[2], True, C.engine_params(ilupp, thr, pre, knobs))
        assert P.levels() == Q.levels() and P.total_nnz == Q.total_nnz(), (name, tag)
        for k in range(Q.levels()):
            for x, y in zip(C.level_arrays(P.level(k)), C.level_arrays(Q.level(k))):
                assert np.array_equal(x, y, equal_nan=(np.asarray(x).dtype.kind == "f")), (name, tag, k)
print("ok")
''' % (os.path.dirname(HERE), HERE)
    env = dict(os.environ, ILUPP_DP_STORE="150", ILUPP_DEBUG="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stderr.count("status 1 at step") + r.stderr.count("status 2 at step") > 20 and "status 3 at step" in r.stderr


def test_batched_construction_is_the_same_objects_and_runs_side_by_side():
    """ILUppPreconditioner.batch (ilupp_hip_ml_create_batch: BASELINE config 5's many-matrices shape): every member identical to the
    object built alone -- levels, total_nnz, apply, apply_trans bit for bit, with the default (pivoting) parameters, with
    default_configuration(10) and with the family without pivoting; matrices of different sizes in one batch; and a batch of 16
    chains costs far less than 16 times one (they share one launch, one workgroup each)"""
    import time
    import ilupp_amd as ilupp
    mats = [sp.csr_matrix(matgen.random_dd(n, 8, 25.0, 100 + k), shape=(n, n)) for k, n in enumerate([3000, 2500, 4000, 3000, 1200, 3500])]
    for cfg in (None, 10, 1, 11):                       # (11: inverse-based dropping -- partialILUC as a chain, combined like the chains with pivoting)
        p = ilupp.iluplusplus_precond_parameter()
        if cfg is not None:
            p.default_configuration(cfg)
        p.threshold = 1e-2
        B = ilupp.ILUppPreconditioner.batch(mats, params=p)
        assert len(B) == len(mats)
        for A, Pb in zip(mats, B):
            P1 = ilupp.ILUppPreconditioner(A, params=p)
            b = C.rhs(A.shape[0])
            assert Pb.total_nnz == P1.total_nnz and Pb.pr.levels() == P1.pr.levels()
            assert np.array_equal(Pb @ b, P1 @ b) and np.array_equal(Pb.T @ b, P1.T @ b)
    assert ilupp.ILUppPreconditioner.batch([]) == []
    with pytest.raises(TypeError):
        ilupp.ILUppPreconditioner.batch([mats[0], mats[1].tocsc()])
    # a member the engine refuses (the move-to-corner ordering on a matrix where the reference's own result is undefined): reported by its
    # number, the other members' objects are not handed out half-done, nothing leaks (strict pool) -- and the next batch works
    import ml_cases
    cases = dict(ml_cases.matrices())
    q = ml_cases.engine_params(ilupp, 0.05, ("MAX_WEIGHTED_MATCHING_ORDERING", "DD_SYMM_MOVE_CORNER_ORDERING_IM"), {})
    with pytest.raises(NotImplementedError, match="matrix 1 of the batch"):
        ilupp.ILUppPreconditioner.batch([cases["laplace2d_400"], cases["rdd_300"], cases["p3d_6_7_5"]], params=q)
    ok = ilupp.ILUppPreconditioner.batch([cases["laplace2d_400"], cases["p3d_6_7_5"]], params=q)
    assert ok[0].total_nnz == ilupp.ILUppPreconditioner(cases["laplace2d_400"], params=q).total_nnz
    # side by side: 16 chains of n = 20000
    big = [sp.csr_matrix(matgen.random_dd(20000, 8, 25.0, 500 + k), shape=(20000, 20000)) for k in range(16)]
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(10)
    p.threshold = 1e-3
    ilupp.ILUppPreconditioner(big[0], params=p)                                        # (warm: pool, code objects)
    t0 = time.perf_counter(); one = ilupp.ILUppPreconditioner(big[0], params=p); t_one = time.perf_counter() - t0
    t0 = time.perf_counter(); B = ilupp.ILUppPreconditioner.batch(big, params=p); t_batch = time.perf_counter() - t0
    assert B[0].total_nnz == one.total_nnz and np.array_equal(B[0] @ C.rhs(20000), one @ C.rhs(20000))
    assert t_batch < 4.0 * t_one, (t_batch, t_one)
    # ... and so do 16 chains of the factorisation WITHOUT pivoting under inverse-based dropping (default_configuration(11))
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(11)
    p.threshold = 1e-3
    ilupp.ILUppPreconditioner(big[0], params=p)
    t0 = time.perf_counter(); one = ilupp.ILUppPreconditioner(big[0], params=p); t_one = time.perf_counter() - t0
    t0 = time.perf_counter(); B = ilupp.ILUppPreconditioner.batch(big, params=p); t_batch = time.perf_counter() - t0
    assert B[5].total_nnz == ilupp.ILUppPreconditioner(big[5], params=p).total_nnz and np.array_equal(B[0] @ C.rhs(20000), one @ C.rhs(20000))
    assert t_batch < 4.0 * t_one, (t_batch, t_one)
