"""GPU-resident Krylov building blocks: the preconditioners of this package and a CSR
operator that work on torch tensors living in HBM, so that an iteration of CG / BiCGstab / GMRES never crosses PCIe.

    import torch, ilupp_amd.device as ild
    A = ild.DeviceCSR.from_scipy(A_scipy)                  # one H2D copy
    M = ild.DevicePreconditioner("ICholT", A, add_fill_in=0, threshold=0.0)
    x = ild.cg(A, b, M, maxiter=50)                        # b, x: torch.float64 tensors on the GPU

Everything is ordered on torch's current stream (ilupp_hip_set_caller_stream): no host synchronisation per call.
The reference's counterpart is the loop of iterative_solvers_implementation.h:385-530 around
matrix_sparse::matrix_vector_multiplication (sparse_implementation.h:2733-2760) and apply_preconditioner_only.
"""
import torch

from . import _native

_FACTORIES = {
    "ILU0": (_native.ILU0Preconditioner_device, ()),
    "ILUT": (_native.ILUTPreconditioner_device, ("fill_in", "threshold")),
    "IChol0": (_native.IChol0Preconditioner_device, ()),
    "ICholT": (_native.ICholTPreconditioner_device, ("add_fill_in", "threshold")),
    "ILUC": (_native.ILUCPreconditioner_device, ("fill_in", "threshold")),
    "ILUpp": (_native.MultilevelILUCDPPreconditioner_device, ("params",)),     # params: an iluplusplus_precond_parameter (no default: see ILUppPreconditioner)
}
_DEFAULTS = {"ILUT": {"fill_in": 100, "threshold": 0.1}, "ICholT": {"add_fill_in": 0, "threshold": 0.0},
             "ILUC": {"fill_in": 100, "threshold": 0.1}}


def _on_current_stream():
    _native.set_caller_stream(torch.cuda.current_stream().cuda_stream, True)


class DeviceCSR:
    """a square CSR matrix in HBM (fp64 values, int32 indices) with a bit-exact matvec"""

    def __init__(self, data, indices, indptr):
        assert data.is_cuda and data.dtype == torch.float64 and indices.dtype == torch.int32 and indptr.dtype == torch.int32
        self.data, self.indices, self.indptr = data.contiguous(), indices.contiguous(), indptr.contiguous()
        self.n = indptr.numel() - 1
        self.nnz = data.numel()
        self.shape = (self.n, self.n)

    @classmethod
    def from_scipy(cls, A, device=None):
        import numpy as np
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        A.sort_indices()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        return cls(torch.from_numpy(np.ascontiguousarray(A.data, dtype=np.float64)).to(dev),
                   torch.from_numpy(A.indices.astype(np.int32)).to(dev), torch.from_numpy(A.indptr.astype(np.int32)).to(dev))

    def matvec(self, x, out=None):
        y = torch.empty_like(x) if out is None else out
        rc = _native.lib().ilupp_hip_spmv_device(self.data.data_ptr(), self.indices.data_ptr(), self.indptr.data_ptr(), self.n,
                                                 self.nnz, x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc:
            _native._raise(rc)
        return y

    __matmul__ = matvec


class DevicePreconditioner:
    """ILU0 / ILUT / ILUC / IChol0 / ICholT / multilevel ILU++ of a DeviceCSR, applied to device tensors in place or out of place"""

    def __init__(self, kind, A, **params):
        make, names = _FACTORIES[kind]
        p = dict(_DEFAULTS.get(kind, {}))
        p.update(params)
        _on_current_stream()
        self.pr = make(A.data.data_ptr(), A.indices.data_ptr(), A.indptr.data_ptr(), A.n, True, *[p[k] for k in names])
        self.n = A.n
        self.shape = A.shape

    def apply_(self, x, transpose=False):
        """in place on a contiguous fp64 device tensor; asynchronous, ordered on torch's current stream"""
        assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous() and x.numel() == self.n
        _on_current_stream()
        self.pr.apply_device(x.data_ptr(), self.n, transpose=transpose, sync=False)
        return x

    def matvec(self, x):
        return self.apply_(x.clone())

    __matmul__ = matvec

    def apply_part_(self, x, left, transpose=False):
        """the left or the right half of a split preconditioner in place (multilevel ILU++ objects only)"""
        assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous() and x.numel() == self.n
        _on_current_stream()
        self.pr.apply_part_device(x.data_ptr(), self.n, left, transpose=transpose, sync=False)
        return x

    def sync(self):
        self.pr.sync()


def cg(A, b, M=None, x0=None, maxiter=100, rtol=0.0, check_every=0):
    """preconditioned conjugate gradients on device tensors.  No host round trip per iteration: the scalars stay 0-dim
    device tensors; the residual is only looked at every `check_every` iterations (0 = never: run maxiter iterations)."""
    x = torch.zeros_like(b) if x0 is None else x0.clone()
    r = b - A.matvec(x) if x0 is not None else b.clone()
    z = M.matvec(r) if M is not None else r.clone()
    p = z.clone()
    rz = torch.dot(r, z)
    bnorm = torch.linalg.vector_norm(b)
    Ap = torch.empty_like(b)
    for it in range(maxiter):
        A.matvec(p, out=Ap)
        alpha = rz / torch.dot(p, Ap)
        x.add_(p * alpha)
        r.sub_(Ap * alpha)
        if check_every and (it + 1) % check_every == 0 and rtol > 0.0:
            if float(torch.linalg.vector_norm(r) / bnorm) <= rtol:      # the only device-to-host read
                break
        z = M.matvec(r) if M is not None else r
        rz_new = torch.dot(r, z)
        p = z + p * (rz_new / rz)
        rz = rz_new
    if M is not None:
        M.sync()
    return x


def bicgstab(A, b, M=None, x0=None, maxiter=100, rtol=0.0, check_every=0, history=None):
    """left-preconditioned BiCGstab on device tensors, statement for statement the loop of the reference
    (iterative_solvers_implementation.h:385-530 with a LEFT preconditioner application): the residual recurrence runs on
    r = M^-1 (b - A x), Ap = M^-1 (A p), As = M^-1 (A s).  No host round trip per iteration (the scalars stay 0-dim device tensors;
    the residual norm is looked at every `check_every` iterations only); `history`, when a list, receives a copy of the iterate after
    every iteration (tests)."""
    def prec(v):
        return M.matvec(v) if M is not None else v.clone()
    y = torch.zeros_like(b) if x0 is None else x0.clone()
    r0star = b.clone() if x0 is None else b - A.matvec(y)
    r = prec(r0star)
    r0star = r.clone()
    p = r.clone()
    initial_res = torch.linalg.vector_norm(r)
    Ap = torch.empty_like(b)
    As = torch.empty_like(b)
    for it in range(maxiter):
        Ap = prec(A.matvec(p))
        dot_r_r0star = torch.dot(r, r0star)
        alpha = dot_r_r0star / torch.dot(Ap, r0star)
        s = r - alpha * Ap
        As = prec(A.matvec(s))
        omega = torch.dot(As, s) / torch.dot(As, As)
        y.add_(alpha * p)
        y.add_(omega * s)
        r = s - omega * As
        beta = (torch.dot(r, r0star) / dot_r_r0star) * (alpha / omega)
        p.sub_(omega * Ap)
        p = beta * p + r
        if history is not None:
            history.append(y.clone())
        if check_every and (it + 1) % check_every == 0 and rtol > 0.0:
            if float(torch.linalg.vector_norm(r) / initial_res) <= rtol:      # the only device-to-host read
                break
    if M is not None:
        M.sync()
    return y


def bicgstab_split(A, b, M, min_iter=1, max_iter=500, rtol=1e-4, atol=1e-4):
    """BiCGstab with SPLIT preconditioning as the reference's `solve` runs it (solving_routines_implementation.h:81 ->
    iterative_solvers_implementation.h:385-530 from the zero vector): r = L' b, Ap = L'(A(R' p)), x = R' y at the end; the loop goes on
    while (res / initial_res > rtol or res > atol) and iter < max_iter, or iter < min_iter -- the residual is looked at after every
    iteration, as there.  Returns (x, iterations, res / initial_res, res).  M: a DevicePreconditioner of the "ILUpp" kind."""
    def pmv(v):
        w = M.apply_part_(v.clone(), left=False)
        return M.apply_part_(A.matvec(w), left=True)
    r = M.apply_part_(b.clone(), left=True)
    r0star = r.clone()
    p = r.clone()
    y = torch.zeros_like(b)
    initial_res = float(torch.linalg.vector_norm(r))
    res = initial_res
    it = 0
    # (IEEE division as the reference's doubles do it: a zero right-hand side gives 0 / 0 = NaN, every comparison with it is false, the
    # loop runs its min_iter iterations and the caller reports "did not converge" -- not a ZeroDivisionError)
    rel = lambda a, b: a / b if b != 0.0 else (float("nan") if (a == 0.0 or a != a) else float("inf"))
    while (((rel(res, initial_res) > rtol) or res > atol) and it < max_iter) or it < min_iter:
        it += 1
        Ap = pmv(p)
        dot_r_r0star = torch.dot(r, r0star)
        alpha = dot_r_r0star / torch.dot(Ap, r0star)
        s = r - alpha * Ap
        As = pmv(s)
        omega = torch.dot(As, s) / torch.dot(As, As)
        y.add_(alpha * p)
        y.add_(omega * s)
        r = s - omega * As
        beta = (torch.dot(r, r0star) / dot_r_r0star) * (alpha / omega)
        p.sub_(omega * Ap)
        p = beta * p + r
        res = float(torch.linalg.vector_norm(r))
    x = M.apply_part_(y, left=False)
    M.sync()
    return x, it, rel(res, initial_res), res
