"""GPU-resident Krylov building blocks: the device BiCGstab against the iterates of the REFERENCE's own loop
(iterative_solvers_implementation.h:385-530, golden vectors of tests/golden/make_golden_ml.py), the bit-exact CSR product for long
rows, and ILUC in the device layer."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import golden_util as G
import matgen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_bicgstab_matches_reference_iterates():
    """ilupp_amd.device.bicgstab with an ILU(0) preconditioner: the iterate after k = 1..6 iterations agrees with the reference's
    bicgstab(P, LEFT, ...) to 1e-12 relative (the dot products are summed in another order on the GPU; everything else --
    the preconditioner, the matrix product -- is bit-identical)"""
    import torch
    import ilupp_amd.device as ild
    z = np.load(os.path.join(ROOT, "tests", "golden", "ml.npz"))
    d, i, p, csr = G.get_mat(z, "bicg/A")
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    Ad = ild.DeviceCSR.from_scipy(A)
    M = ild.DevicePreconditioner("ILU0", Ad)
    b = torch.from_numpy(z["bicg/b"]).cuda()
    hist = []
    ild.bicgstab(Ad, b, M, maxiter=6, history=hist)
    torch.cuda.synchronize()
    for k in range(1, 7):
        xr = z["bicg/x_%d" % k]
        x = hist[k - 1].cpu().numpy()
        assert np.max(np.abs(x - xr)) <= 1e-12 * np.max(np.abs(xr)), (k, np.max(np.abs(x - xr)) / np.max(np.abs(xr)))


def test_spmv_long_rows_bit_exact():
    """rows of more than 8 entries on average take the 8-lanes-per-row kernel: same sum, bit for bit, as scipy's csr_matvec"""
    import torch
    import ilupp_amd.device as ild
    rng = np.random.default_rng(3)
    for A in (sp.random(3000, 3000, density=0.02, random_state=rng, format="csr") + sp.identity(3000, format="csr"),
              sp.csr_matrix(matgen.box_stencil((20, 20, 20)), shape=(8000, 8000))):
        A = sp.csr_matrix(A); A.sort_indices()
        n = A.shape[0]
        assert A.nnz > 8 * n
        Ad = ild.DeviceCSR.from_scipy(A)
        x = G.rhs(n)
        y = Ad.matvec(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), A @ x)


def test_device_layer_iluc_and_cg():
    """ILUC through the device layer (the object of test_gpu_iluc.py behind DevicePreconditioner), and a few CG steps on it"""
    import torch
    import ilupp_amd as ilupp
    import ilupp_amd.device as ild
    d, i, p = matgen.poisson3d(14, 12, 10)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    Ad = ild.DeviceCSR.from_scipy(A)
    M = ild.DevicePreconditioner("ILUC", Ad, fill_in=8, threshold=1e-2)
    b = G.rhs(n)
    y = M.matvec(torch.from_numpy(b).cuda())
    M.sync()
    P = ilupp.ILUCPreconditioner(A, fill_in=8, threshold=1e-2)
    x = b.copy(); P.apply(x)
    assert np.array_equal(y.cpu().numpy(), x)
    xs = ild.cg(Ad, torch.from_numpy(b).cuda(), M, maxiter=25)
    torch.cuda.synchronize()
    r = b - A @ xs.cpu().numpy()
    assert np.linalg.norm(r) <= 1e-6 * np.linalg.norm(b)
