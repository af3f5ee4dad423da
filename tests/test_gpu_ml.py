"""GPU parity tests of SURVEY section 8(f) rank 3: the multilevel ILU++ preconditioner without pivoting (ilupp_amd/csrc/piluc_df.hip,
ml.hip; reference preconditioner_implementation.h:1350-1665 over ILUCDP.hpp:1405-2231, apply :433-488).

Everything is compared bit for bit -- the bar of this package for all its factorisations; the 1e-12 of the north star is implied:
* against the golden vectors of the REAL reference (tests/golden/ml10.npz): levels, sizes, total_nnz, every level's factors, middle
  diagonal, permutations and scalings (sha256), apply and apply_trans, CSR and CSC input, twelve parameter sets;
* against the oracle on larger matrices (n up to 10^5: fill, levels ended by small pivots, Schur complements), and at n = 10^6 through
  the class API;
* the Python class (ilupp/__init__.py:171-203): LinearOperator protocol, total_nnz, repr, factors() == [] (binding.cpp:158-163),
  the refusal of the pivoting family, a vector of the wrong size;
* the device-pointer entry points, and BiCGstab on the GPU preconditioned with the multilevel object.
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


def _native_ml(M, params):
    from ilupp_amd import _native
    M = M.copy()
    M.sort_indices()
    return _native.MultilevelILUCDPPreconditioner(M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32), sp.isspmatrix_csr(M), params)


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("name", [n for n, _ in C.matrices()])
def test_against_reference_vectors(name, fmt):
    import ilupp_amd as ilupp
    gold = np.load(os.path.join(HERE, "golden", "ml10.npz"))
    key = "%s_%s" % (name, fmt)
    kind = sp.csr_matrix if fmt == "csr" else sp.csc_matrix
    n = gold[key + "/indptr"].shape[0] - 1
    M = kind((gold[key + "/data"], gold[key + "/indices"], gold[key + "/indptr"]), shape=(n, n))
    b = C.rhs(n)
    for tag, thr, pre, knobs in C.PARAMS:
        k2 = "%s/%s" % (key, tag)
        if k2 + "/refused" in gold.files:
            with pytest.raises(NotImplementedError, match="undefined"):
                _native_ml(M, C.engine_params(ilupp, thr, pre, knobs))
            continue
        P = _native_ml(M, C.engine_params(ilupp, thr, pre, knobs))
        info = gold[k2 + "/info"]
        assert P.levels() == info[0] and P.total_nnz == info[1], k2
        for k in range(P.levels()):
            lv = P.level(k)
            assert lv["n"] == info[2 + k], (k2, k)
            assert np.array_equal(_digest(C.level_arrays(lv)), gold[k2 + "/levels_sha"][k]), (k2, k)
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, gold[k2 + "/apply"], equal_nan=True), k2
        x = b.copy(); P.apply_trans(x)
        assert np.array_equal(x, gold[k2 + "/apply_trans"], equal_nan=True), k2


def _against_oracle(M, params_tuple):
    import ilupp_amd as ilupp
    from oracle import oracle as O
    thr, pre, knobs = params_tuple
    M = M.copy(); M.sort_indices()
    a = (M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32), sp.isspmatrix_csr(M))
    Q = O.orc().ml(a, C.oracle_params(O, thr, pre, knobs))
    P = _native_ml(M, C.engine_params(ilupp, thr, pre, knobs))
    assert P.levels() == Q.levels() and P.total_nnz == Q.total_nnz()
    for k in range(Q.levels()):
        for x, y in zip(C.level_arrays(P.level(k)), C.level_arrays(Q.level(k))):
            assert x.shape == y.shape and np.array_equal(x, y, equal_nan=True), k
    b = C.rhs(M.shape[0])
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, Q.apply(b), equal_nan=True)
    x = b.copy(); P.apply_trans(x)
    assert np.array_equal(x, Q.apply(b, O.TRANSPOSE), equal_nan=True)
    return Q.levels()


PQ = ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING")


def test_larger_matrices_against_oracle():
    """fill (threshold 1e-3 .. 1e-4 on diagonally dominant random rows), mesh matrices symmetric and with convection, CSC input"""
    for n, k in ((20000, 6), (100000, 8)):
        M = sp.csr_matrix(matgen.random_dd(n, k=k), shape=(n, n))
        for thr in (1e-3, 1e-4):
            assert _against_oracle(M, (thr, PQ, {})) == 1
    d, i, p = matgen.poisson3d(40, 40, 40)
    A = sp.csr_matrix((d, i, p))
    for thr in (0.05, 0.01):
        _against_oracle(A, (thr, PQ, {}))
    d, i, p = matgen.poisson3d(30, 30, 30)
    n = p.shape[0] - 1
    A = (sp.csr_matrix((d, i, p), shape=(n, n)) + 0.8 * sp.diags([np.ones(n - 1)], [1], shape=(n, n))).tocsr()
    _against_oracle(A, (0.005, PQ, {}))
    _against_oracle(A.tocsc(), (0.02, PQ, {}))


def test_many_levels_against_oracle():
    """weak diagonals: levels ended by small pivots one after the other, Schur complements that fill"""
    A = C.weak_random(700, 0.01, 0.3, 7)
    assert _against_oracle(A, (0.05, PQ, {})) >= 5
    assert _against_oracle(A.tocsc(), (0.2, PQ, {"THRESHOLD_SHIFT_SCHUR": 1e-2})) >= 2
    # a structurally missing diagonal is a pivot 0: level 0 ends at once (k > 0), the last level takes it as a "zero pivot"
    B = A.tolil(); B[5, 5] = 0.0; B[40, 40] = 0.0; B = B.tocsr(); B.eliminate_zeros()
    _against_oracle(B, (0.1, (), {}))
    _against_oracle(B, (0.1, (), {"SMALL_PIVOT_TERMINATES": False}))


def test_class_api():
    import ilupp_amd as ilupp
    from oracle import oracle as O
    n = 1000000
    A = sp.csr_matrix(matgen.random_dd(n, k=8), shape=(n, n))          # BASELINE config C5's shape: unsymmetric, n = 10^6
    A.indices = A.indices.astype(np.int32); A.indptr = A.indptr.astype(np.int32)
    params = ilupp.iluplusplus_precond_parameter()
    params.default_configuration(1)
    params.threshold = 1e-3
    P = ilupp.ILUppPreconditioner(A, params=params)
    Q = O.orc().ml((A.data, A.indices, A.indptr, True), O.ml_params(1e-3))
    assert P.total_nnz == Q.total_nnz() and P.pr.levels() == Q.levels()
    assert repr(P) == "<%dx%d ILUppPreconditioner with nnz=%d, dtype=float64>" % (n, n, Q.total_nnz())
    assert P.factors() == [] and P.memory == 0.0
    b = C.rhs(n)
    want = Q.apply(b)
    assert np.array_equal(P @ b, want) and np.array_equal(P.dot(b), want)
    assert np.array_equal(P.T @ b, Q.apply(b, O.TRANSPOSE))
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, want)
    X = np.stack([b, 2 * b], axis=1)
    assert (P @ X).shape == (n, 2) and np.array_equal((P @ X)[:, 0], want)
    with pytest.raises(RuntimeError, match="vector has wrong size for preconditioner!"):
        P.pr.apply(np.ones(n - 1))
    # default-constructed parameters: the factorisation with pivoting (tests/test_gpu_mlp.py), on a small matrix -- it is a chain of n steps
    A2 = sp.csr_matrix(matgen.random_dd(400, k=6, diag=2.0), shape=(400, 400))
    Pd = ilupp.ILUppPreconditioner(A2, threshold=0.1)
    Qd = O.orc().ml((A2.data, A2.indices, A2.indptr, True), O.ml_params(0.1, **O.PIVOTING_DEFAULTS))
    assert Pd.total_nnz == Qd.total_nnz() and np.array_equal(Pd @ C.rhs(400), Qd.apply(C.rhs(400)))


def test_device_entry_points_and_bicgstab():
    """the matrix and the vectors stay in HBM: construction from device pointers, apply on a device vector, and the reference's
    solver loop (BiCGstab, iterative_solvers_implementation.h:385-530) with the multilevel object as its preconditioner"""
    import torch
    import ilupp_amd as ilupp
    import ilupp_amd.device as ild
    from oracle import oracle as O
    d, i, p = matgen.poisson3d(24, 20, 22)
    n = p.shape[0] - 1
    A = (sp.csr_matrix((d, i, p), shape=(n, n)) + 0.5 * sp.diags([np.ones(n - 1)], [1], shape=(n, n))).tocsr()
    A.sort_indices()
    params = ilupp.iluplusplus_precond_parameter()
    params.default_configuration(1)
    params.threshold = 0.02
    dA = ild.DeviceCSR.from_scipy(A)
    M = ild.DevicePreconditioner("ILUpp", dA, params=params)
    Q = O.orc().ml((A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True), O.ml_params(0.02))
    b = C.rhs(n)
    tb = torch.from_numpy(b).cuda()
    assert np.array_equal((M @ tb).cpu().numpy(), Q.apply(b))
    tx = tb.clone(); M.apply_(tx, transpose=True); M.sync()
    assert np.array_equal(tx.cpu().numpy(), Q.apply(b, O.TRANSPOSE))
    x = ild.bicgstab(dA, tb, M, maxiter=25)
    r = b - A @ x.cpu().numpy()
    assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(b)


def test_solve_like_the_reference_tests():
    """ilupp.solve (ilupp/__init__.py:85-119): the reference's own tests of it (test/tests.py:344-383) with parameters of the built
    family -- x_exact recovered to np.allclose, convergence info returned, "did not converge" raised"""
    import ilupp_amd as ilupp
    A = C.laplace2d_matrix(900)
    n = A.shape[0]
    x_exact = np.random.default_rng(3).random(n)
    b = A @ x_exact
    for fmt in ("csr", "csc"):
        param = ilupp.iluplusplus_precond_parameter()
        param.default_configuration(1)
        param.threshold = 1e-2
        x, info = ilupp.solve(A.asformat(fmt), b, atol=1e-8, rtol=1e-8, params=param, info=True)
        assert np.allclose(x_exact, x)
        assert 1 <= info[0] <= 60 and info[1] < 1e-8 and info[2] < 1e-8
    A = sp.csr_matrix(matgen.random_dd(50, k=5, diag=10.0), shape=(50, 50))
    x_exact = np.linspace(1.0, 2.0, 50)
    param = ilupp.iluplusplus_precond_parameter()
    param.default_configuration(1)
    param.threshold = 0.1
    assert np.allclose(ilupp.solve(A, A @ x_exact, atol=1e-8, params=param), x_exact)
    with pytest.raises(RuntimeError, match="did not converge"):
        ilupp.solve(C.laplace2d_matrix(900), np.ones(900), atol=1e-14, rtol=1e-14, max_iter=2, params=param)
    assert np.allclose(ilupp.solve(A, A @ x_exact, atol=1e-8), x_exact)      # default-constructed parameters: the pivoting factorisation
    # a zero right-hand side: the reference's loop divides 0 by 0 (NaN: every comparison false), runs its min_iter iterations and
    # reports no convergence -- no ZeroDivisionError
    with pytest.raises(RuntimeError, match="did not converge"):
        ilupp.solve(A, np.zeros(50), params=param)


MWM = ("MAX_WEIGHTED_MATCHING_ORDERING",)


def test_matching_preprocessing_against_oracle():
    """the I-matrix preprocessing of default_configuration(10 .. 13): maximum-weight matching (its permutation and the two scalings come
    from the host, pmwm_implementation.h:385-537), alone and combined with the other steps; a matrix whose large entries lie OFF the
    diagonal, so that the matching permutes every row; n = 10^5 and n = 10^6 unsymmetric"""
    import ilupp_amd as ilupp
    n = 3000
    M = sp.csr_matrix(matgen.random_dd(n, k=6, diag=0.0), shape=(n, n))
    M = (M + sp.diags([np.full(n - 1, 2.0)], [1], shape=(n, n)) + sp.diags([np.full(1, 2.0)], [-(n - 1)], shape=(n, n))).tocsr()
    for pre in (MWM, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS") + MWM, ("SPARSE_FIRST_ORDERING",) + MWM, MWM + ("UNIT_OR_ZERO_DIAGONAL_SCALING",)):
        _against_oracle(M, (0.1, pre, {}))
    _against_oracle(M.tocsc(), (0.1, MWM, {}))
    M1 = sp.csr_matrix(matgen.random_dd(100000, k=8), shape=(100000, 100000))
    for pre in (MWM, MWM + ("PQ_ORDERING",), ("SPARSE_FIRST_ORDERING",) + MWM):
        _against_oracle(M1, (1e-3, pre, {}))
    M2 = sp.csr_matrix(matgen.random_dd(1000000, k=8), shape=(1000000, 1000000))
    _against_oracle(M2, (1e-3, MWM, {}))
    # default_configuration(11) / (12) through the parameter object
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(12)
    assert p.PREPROCESSING.to_names() == ["SPARSE_FIRST_ORDERING", "MAX_WEIGHTED_MATCHING_ORDERING"] and p._uses_partial_iluc()
    p.threshold = 1e-3
    P = ilupp.ILUppPreconditioner(M1, params=p)
    from oracle import oracle as O
    Q = O.orc().ml(O.from_scipy(M1), O.ml_params(1e-3, preprocessing=(7, 4)))
    b = C.rhs(M1.shape[0])
    assert P.total_nnz == Q.total_nnz() and np.array_equal(P @ b, Q.apply(b))
    # a matrix without a perfect matching: identity permutation, unit scalings (pmwm_implementation.h:460-471), then the factorisation
    # meets the empty column as zero pivots
    S = sp.random(200, 200, density=0.03, random_state=np.random.default_rng(1), format="csr").tolil()
    S[:, 7] = 0
    S = S.tocsr(); S.eliminate_zeros()
    _against_oracle(S, (0.05, MWM, {}))


def test_fuzz_against_oracle():
    """random small matrices of all kinds (missing diagonals, wild magnitudes, empty-ish rows, 1 x 1 ... 350 x 350) with random
    preprocessing sequences and knobs (tests/fuzz_ml.py): every level and both applies bit for bit, NaNs and infinities included"""
    import fuzz_ml
    off = int(os.environ.get("ILUPP_FUZZ_OFFSET", "0"))             # other seeds: profiles/tools/fuzz_more.sh
    for seed in range(off, off + 120):
        A, params = fuzz_ml.case(seed)
        try:
            _against_oracle(A, params)
        except AssertionError as e:
            raise AssertionError("fuzz case %d: %s" % (seed, e))


def test_bench_batch_member():
    """the unit of bench.py's multi-GPU C5 batch (`batch_ml`: one matrix per rank through ilupp_amd.batched.run_batch) on one GPU: the record
    is reproducible (what the bench asserts between the sharded and the single-rank run) and agrees with the oracle"""
    import sys
    import torch
    from oracle import oracle as O
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    from ilupp_amd.batched import run_batch
    dev = torch.device("cuda", 0)
    recs = run_batch(2, lambda m: bench.ml_batch_member(dev, m, n=200000))
    again = [bench.ml_batch_member(dev, m, n=200000) for m in range(2)]
    assert [r[:4] for r in recs] == [r[:4] for r in again] and recs[0][3] != recs[1][3]
    d, i, p = matgen.random_dd(200000, 8, 25.0, 12345)
    Q = O.orc().ml((d, i, p, True), O.ml_params(1e-3))
    import hashlib
    assert recs[0][1] == Q.levels() and recs[0][2] == Q.total_nnz()
    assert recs[0][3] == hashlib.sha256(Q.apply(np.ones(200000)).tobytes()).hexdigest()


def test_chain_kernel_with_vectors_in_memory(capfd):
    """working rows beyond the chain kernel's LDS capacity (2 048 entries): the same kernel body on global memory (k_piluc_chain_mem) takes the
    chain over AT the step that does not fit.  ILUPP_PILUC_CHAIN_MEM=1 runs it from the first step (fuzz cases, several levels, the rules that
    are recurrences); then a matrix whose rows do outgrow LDS in the middle of a level, under inverse-based dropping (which has no other path)"""
    import fuzz_ml
    os.environ["ILUPP_PILUC_CHAIN"] = "1"
    os.environ["ILUPP_PILUC_CHAIN_MEM"] = "1"
    try:
        for seed in range(40):
            A, params = fuzz_ml.case(2000 + seed)
            _against_oracle(A, params)
        assert _against_oracle(C.weak_random(700, 0.01, 0.3, 7), (0.05, PQ, {})) >= 5
        M = sp.csr_matrix(matgen.random_dd(3000, k=7, diag=2.0), shape=(3000, 3000))
        _against_oracle(M, (0.02, PQ, {"USE_WEIGHTED_DROPPING": True}))
        _against_oracle(M, (0.02, PQ, {"USE_INVERSE_DROPPING": True, "USE_WEIGHTED_DROPPING2": True, "fill_in": 10}))
    finally:
        del os.environ["ILUPP_PILUC_CHAIN"]
        del os.environ["ILUPP_PILUC_CHAIN_MEM"]
    # (the case of profiles/tools/fuzz_chain.py 7000 8000 that ran into the LDS capacity: its generator, up to that matrix)
    rng = np.random.default_rng(7000)
    for it in range(15):
        n = int(rng.choice([500, 1200, 2500, 4000]))
        A = (sp.random(n, n, density=float(rng.choice([2.0, 4.0, 8.0])) / n, random_state=rng, format="csr") + sp.eye(n) * float(rng.choice([0.3, 0.6, 1.5]))).tocsr()
        thr = float(rng.choice([0.02, 0.05, 0.2]))
        rng.integers(0, 4)
        if rng.random() < 0.3:
            rng.choice([3, 10])
    assert n == 4000 and thr == 0.2
    capfd.readouterr()
    os.environ["ILUPP_DEBUG"] = "1"
    try:
        assert _against_oracle(A, (0.2, PQ, {"USE_INVERSE_DROPPING": True, "USE_STANDARD_DROPPING": False})) >= 2
    finally:
        del os.environ["ILUPP_DEBUG"]
    assert "vectors in memory" in capfd.readouterr().err           # (the log of the level that was taken over)


def test_bench_multilevel_object():
    """the matrix of bench.py's "C5L" extra (random rows with a weak diagonal: several levels at n = 10^6) at n = 10^6 itself: levels, every
    level's arrays and the apply against the oracle -- the object the driver's bench line reports is a real multilevel one"""
    n = 1000000
    M = sp.csr_matrix(matgen.random_dd(n, 3, 0.6, 12345), shape=(n, n))
    assert _against_oracle(M, (0.3, PQ, {})) >= 3


def test_degenerate_inputs_against_oracle():
    """a matrix without entries, a 1 x 1 zero, empty rows and columns, an all-zero row with a stored zero: the reference divides by the
    zero norms and carries NaNs and infinities through every level; the engine and the oracle do the same, bit for bit"""
    Z = sp.csr_matrix((3, 3))
    E = sp.csr_matrix((np.array([0.0]), np.array([0], dtype=np.int32), np.array([0, 1], dtype=np.int32)), shape=(1, 1))
    R = sp.random(40, 40, density=0.1, random_state=np.random.default_rng(8), format="lil")
    R[7, :] = 0; R[:, 11] = 0; R[20, :] = 0
    R = R.tocsr(); R.eliminate_zeros()
    S = R.copy().tolil(); S[20, 3] = 1.0; S = S.tocsr(); S.data[S.indptr[20]] = 0.0          # a stored zero in an otherwise empty row
    for A in (Z, E, R, S, R.tocsc()):
        for pre in (PQ, (), MWM, ("NORMALIZE_ROWS", "SPARSE_FIRST_ORDERING", "UNIT_OR_ZERO_DIAGONAL_SCALING")):
            for thr in (0.0, 0.1):
                _against_oracle(A, (thr, pre, {}))


def test_chain_kernel_of_partial_iluc():
    """partialILUC as a sequential chain (pilucdp.hip: k_piluc_chain -- Crout's three lists kept as the reference keeps them, the working
    vectors in LDS): what the factorisation without pivoting runs on when its dropping rules are recurrences over all steps
    (inverse-based, weighted: presets 11, 12) and when its rows outgrow the dataflow kernel's LDS classes.  ILUPP_PILUC_CHAIN=1 sends every
    level there: the plain rules too, several levels, Schur complements, bounded fill, stores that fill up, n = 10^5 -- bit for bit; and the
    two presets through the parameter object"""
    import fuzz_ml
    import ilupp_amd as ilupp
    os.environ["ILUPP_PILUC_CHAIN"] = "1"
    try:
        A = C.weak_random(700, 0.01, 0.3, 7)
        assert _against_oracle(A, (0.05, PQ, {})) >= 5
        _against_oracle(A.tocsc(), (0.2, PQ, {"THRESHOLD_SHIFT_SCHUR": 1e-2, "fill_in": 6}))
        _against_oracle(C.laplace2d_matrix(2500), (0.01, MWM, {"USE_STANDARD_DROPPING": True, "COMBINE_FACTOR": 1}))
        _against_oracle(sp.csr_matrix(matgen.random_dd(100000, k=8), shape=(100000, 100000)), (1e-3, PQ, {}))
        os.environ["ILUPP_DP_STORE"] = "200"
        try:
            _against_oracle(sp.csr_matrix(matgen.random_dd(3000, k=7, diag=2.0), shape=(3000, 3000)), (0.02, PQ, {}))
        finally:
            del os.environ["ILUPP_DP_STORE"]
        for seed in range(40):
            A, params = fuzz_ml.case(1000 + seed)
            _against_oracle(A, params)
    finally:
        del os.environ["ILUPP_PILUC_CHAIN"]
    # presets 11 and 12 (parameters_implementation.h:935-945) through the parameter object, without the switch
    from oracle import oracle as O
    M = sp.csr_matrix(matgen.random_dd(20000, k=7, diag=4.0), shape=(20000, 20000))
    for preset in (11, 12):
        p = ilupp.iluplusplus_precond_parameter()
        p.init(ilupp.preprocessing_sequence(["NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"]), preset)
        p.threshold = 0.05
        P = ilupp.ILUppPreconditioner(M, params=p)
        Q = O.orc().ml(O.from_scipy(M), C.block_to_oracle(O, p._to_ml_params()))
        b = C.rhs(20000)
        assert P.total_nnz == Q.total_nnz() and np.array_equal(P @ b, Q.apply(b))
