"""`solve` behind the C ABI (ilupp_hip_solve: the reference's `_ilupp.solve`, binding.cpp:200-230 -- multilevel preconditioner, then
BiCGstab with SPLIT preconditioning from the zero vector, iterative_solvers_implementation.h:385-530) against the REAL reference's
outputs (tests/golden/solve.npz from make_golden_solve.py): converged or not, the number of iterations, the solution and the residual
measures.  The preconditioner is bit-identical to the reference's (test_gpu_ml / test_gpu_mlp); the inner products of the iteration
are tree reductions on the GPU and sequential sums in the reference, so the iterates agree to rounding, not bit for bit: the
tolerances below say how closely."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import ml_cases as C

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "solve.npz"))


def _matrix(gold, key, fmt):
    a = (gold[key + "/data"], gold[key + "/indices"], gold[key + "/indptr"])
    n = a[2].shape[0] - 1
    return (sp.csr_matrix if fmt == "csr" else sp.csc_matrix)(a, shape=(n, n))


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("name", [n for n, _, _ in C.solve_matrices()])
def test_solve_against_reference_vectors(gold, name, fmt):
    import ilupp_amd as ilupp
    key = "%s_%s" % (name, fmt)
    A, b = _matrix(gold, key, fmt), gold[key + "/b"]
    for tag, thr, pre, knobs, rtol, atol, max_iter in C.SOLVE_PARAMS:
        x_ref = gold["%s/%s/x" % (key, tag)]
        ok_ref, it_ref, rel_ref, res_ref = gold["%s/%s/info" % (key, tag)]
        p = C.engine_params(ilupp, thr, pre, knobs)
        if not ok_ref:
            with pytest.raises(RuntimeError, match="did not converge"):
                ilupp.solve(A, b, rtol=rtol, atol=atol, max_iter=max_iter, params=p)
            continue
        x, (it, rel, res) = ilupp.solve(A, b, rtol=rtol, atol=atol, max_iter=max_iter, params=p, info=True)
        # the stopping test sits on a residual that falls by orders of magnitude per iteration: the count is the reference's unless the
        # last residual lies within rounding of the tolerance
        assert abs(it - int(it_ref)) <= 1, (tag, it, it_ref)
        assert rel < rtol and res < atol
        if it == int(it_ref):
            assert rel <= 50 * rel_ref + 1e-300 and res <= 50 * res_ref + 1e-300, (tag, rel, rel_ref, res, res_ref)
        scale = np.linalg.norm(x_ref)
        assert np.linalg.norm(x - x_ref) <= 1e-6 * scale, (tag, np.linalg.norm(x - x_ref) / scale)
        r = b - A @ x
        assert np.linalg.norm(r) <= 1e3 * max(atol, rtol * np.linalg.norm(b))


def test_solve_is_deterministic_and_the_same_through_both_bindings(gold):
    """two runs give the same bits (fixed-shape reductions); the compiled shim's `solve` (pybind_module.cpp, the module that stands in
    for ilupp._ilupp) and the ctypes binding call the same entry point"""
    import ilupp_amd as ilupp
    from ilupp_amd import _ilupp_hip as shim, _native
    A, b = _matrix(gold, "p3d_shift_1320_csr", "csr"), gold["p3d_shift_1320_csr/b"]
    p = C.engine_params(ilupp, 0.02, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {})
    m = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True)
    r1 = _native.solve(*m, b, 1e-10, 1e-10, 500, p)
    r2 = _native.solve(*m, b, 1e-10, 1e-10, 500, p)
    r3 = shim.solve(*m, b, 1e-10, 1e-10, 500, p)
    assert np.array_equal(r1[0], r2[0]) and r1[1:] == r2[1:]
    assert np.array_equal(r1[0], r3[0]) and tuple(r1[1:]) == tuple(r3[1:])
    assert isinstance(r3, tuple) and len(r3) == 4 and r3[0].dtype == np.float64           # binding.cpp:200: (x, iterations, rel, abs)


def test_solve_matches_the_loop_on_device_tensors(gold):
    """the iteration inside the library and the same recurrence written over torch tensors (ilupp_amd.device.bicgstab_split, for callers
    whose vectors already live in HBM): same iteration count, solutions equal to rounding"""
    import torch
    import ilupp_amd as ilupp
    import ilupp_amd.device as ild
    A, b = _matrix(gold, "laplace2d_900_csr", "csr"), gold["laplace2d_900_csr/b"]
    p = C.engine_params(ilupp, 1e-2, ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING"), {})
    x, (it, rel, res) = ilupp.solve(A, b, rtol=1e-8, atol=1e-8, params=p, info=True)
    dA = ild.DeviceCSR.from_scipy(A)
    M = ild.DevicePreconditioner("ILUpp", dA, params=p)
    xt, it_t, rel_t, res_t = ild.bicgstab_split(dA, torch.from_numpy(b).cuda(), M, rtol=1e-8, atol=1e-8)
    assert it == it_t
    assert np.linalg.norm(x - xt.cpu().numpy()) <= 1e-9 * np.linalg.norm(x)


def test_solve_errors():
    """binding.cpp:209-210 "right-hand side has wrong size!", :227 "did not converge"; a parameter set outside the built family is
    refused before anything is copied; a zero right-hand side runs min_iter iterations on NaN and reports no convergence"""
    import ilupp_amd as ilupp
    from ilupp_amd import _ilupp_hip as shim, _native
    A = C.laplace2d_matrix(400)
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(1)
    m = (A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True)
    for mod in (_native, shim):
        with pytest.raises(RuntimeError, match="right-hand side has wrong size!"):
            mod.solve(*m, np.ones(399), 1e-4, 1e-4, 500, p)
        with pytest.raises(RuntimeError, match="did not converge"):
            mod.solve(*m, np.zeros(400), 1e-4, 1e-4, 500, p)
        q = ilupp.iluplusplus_precond_parameter()
        q.default_configuration(1)
        q.SCHUR_COMPLEMENT = 1
        with pytest.raises(NotImplementedError):
            mod.solve(*m, np.ones(400), 1e-4, 1e-4, 500, q)
    with pytest.raises(RuntimeError, match="right-hand side has wrong size!"):
        ilupp.solve(A, np.ones(399), params=p)


def test_solve_larger_system():
    """n = 10^5 (config 5's scale): converges like the small cases, the solution solves the system"""
    import ilupp_amd as ilupp
    import matgen
    d, i, p_ = matgen.poisson3d(50, 50, 40)
    n = p_.shape[0] - 1
    A = sp.csr_matrix((d, i, p_), shape=(n, n))
    x_exact = np.random.default_rng(5).random(n)
    b = A @ x_exact
    p = ilupp.iluplusplus_precond_parameter()
    p.default_configuration(1)
    p.threshold = 1e-2
    x, (it, rel, res) = ilupp.solve(A, b, rtol=1e-10, atol=1e-8, params=p, info=True)
    assert 1 <= it <= 100 and rel < 1e-10 and res < 1e-8
    assert np.linalg.norm(x - x_exact) <= 1e-7 * np.linalg.norm(x_exact)
