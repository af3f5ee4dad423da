"""GPU tests added in round 2 (run with -m gpu on an MI355X), all through the C ABI:

* the headline config C2 (256^3) and C3 (n = 1e6) at FULL size, array-compared with the reference's own C++ (oracle/_ref,
  when it travelled) or its C restatement;
* the static level-major path (st.hip) on meshes it accepts and on matrices it must hand to the other generations;
* device-resident constructors of ILUT / IChol0 / ICholT, caller-stream ordering, numeric re-factorisation on every
  kernel generation, error paths, the batched multi-GPU driver.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import golden_util as G
import matgen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle():
    from oracle import oracle as O
    return O, (O.ref() if O.ref_available() else O.orc())


def _dev(*arrays):
    import torch
    dev = torch.device("cuda", 0)
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrays]


def _ptrs(ts):
    return [t.data_ptr() for t in ts]


def test_ilu0_config_c2_full_size():
    """BASELINE config C2 at its full size (256^3 7-point Poisson): L, U (indices and values) and apply(ones) are
    array-equal to the reference; the construction took the static level-major path"""
    import torch
    from ilupp_amd import _native
    O, ref = _oracle()
    d, i, p = matgen.poisson3d(256)
    n = p.shape[0] - 1
    t = _dev(d, i, p)
    torch.cuda.synchronize()
    P = _native.ILU0Preconditioner_device(*_ptrs(t), n, True)
    assert P.path() == "ilu0:static-direct"
    x = torch.ones(n, dtype=torch.float64, device=t[0].device)
    torch.cuda.synchronize()
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    Lo, Uo = ref.ilu0((d, i, p, True))
    xo = ref.trisolve(Uo, O.UPPER, O.ID, ref.trisolve(Lo, O.LOWER, O.ID, np.ones(n)))
    assert np.array_equal(x.cpu().numpy(), xo)
    (Ld, Li, Lp, Lcsr, _, _), (Ud, Ui, Up, Ucsr, _, _) = P.factors_info()
    assert Lcsr and Ucsr and Li.dtype == np.int32
    assert np.array_equal(Lp, Lo[2]) and np.array_equal(Li, Lo[1]) and np.array_equal(Ld, Lo[0])
    assert np.array_equal(Up, Uo[2]) and np.array_equal(Ui, Uo[1]) and np.array_equal(Ud, Uo[0])
    assert P.total_nnz == 2 * 66912256
    # the transposed solves on the same object (transposed storages built from the unpacked factors)
    x.fill_(1.0)
    torch.cuda.synchronize()
    P.apply_device(x.data_ptr(), n, transpose=True, sync=True)
    xt = ref.trisolve(Lo, O.LOWER, O.TRANSPOSE, ref.trisolve(Uo, O.UPPER, O.TRANSPOSE, np.ones(n)))
    assert np.array_equal(x.cpu().numpy(), xt)


def test_ilut_config_c3_full_size():
    """BASELINE config C3 at its full size (random diagonally dominant, n = 1e6, nnz = 2e7, ILUT(10, 1e-4)) against the
    reference (about a minute of CPU): L and U array-equal, through the device-resident constructor"""
    import torch
    from ilupp_amd import _native
    O, ref = _oracle()
    n = 1000000
    d, i, p = matgen.random_dd(n, 19, 25.0, 12345)
    t = _dev(d, i, p)
    torch.cuda.synchronize()
    P = _native.ILUTPreconditioner_device(*_ptrs(t), n, True, 10, 1e-4)
    L, U = P.factors_info()
    Lo, Uo = ref.ilut((d, i, p, True), 10, 1e-4)
    assert np.array_equal(L[2], Lo[2]) and np.array_equal(L[1], Lo[1]) and np.array_equal(L[0], Lo[0])
    assert np.array_equal(U[2], Uo[2]) and np.array_equal(U[1], Uo[1]) and np.array_equal(U[0], Uo[0])
    b = G.rhs(n)
    x = torch.from_numpy(b.copy()).to(t[0].device)
    torch.cuda.synchronize()
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    assert np.array_equal(x.cpu().numpy(), O.orc().apply_lu(Lo, Uo, b, O.ID))


_ILUT_TIERS = r"""
import sys, hashlib
sys.path[:0] = [%r, %r]
import numpy as np, matgen
from ilupp_amd import _native
from oracle import oracle as O
d, i, p = matgen.random_dd(60000, 19, 25.0, 4711)
P = _native.ILUTPreconditioner(d, i, p, True, 10, 1e-4)
L, U = P.factors_info()
Lo, Uo = O.orc().ilut((d, i, p, True), 10, 1e-4)
assert np.array_equal(L[0], Lo[0]) and np.array_equal(L[1], Lo[1]) and np.array_equal(L[2], Lo[2])
assert np.array_equal(U[0], Uo[0]) and np.array_equal(U[1], Uo[1]) and np.array_equal(U[2], Uo[2])
h = hashlib.sha256()
for f in (L, U):
    for a in f[:3]:
        h.update(np.ascontiguousarray(a).tobytes())
print(h.hexdigest())
"""


def test_ilut_rows_of_every_tier():
    """a C3-shaped matrix (19 entries per row, ILUT(10, 1e-4)): most working rows outgrow the LDS pieces (tier 1) by their U part and go
    on with the pool in LDS and the U part in global memory (tier 2), some outgrow that too (tier 3: ILUPP_DEBUG=1 prints the counts);
    ILUPP_ILUT_NO_TIER2=1 sends them to global memory at once.  Array-equal to the oracle, and the same bits, either way"""
    code = _ILUT_TIERS % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for env in ({"ILUPP_DEBUG": "1"}, {"ILUPP_DEBUG": "1", "ILUPP_ILUT_NO_TIER2": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(r.stdout.strip().splitlines()[-1])
        import re
        m = re.search(r"ilut_wp: (\d+) of (\d+) rows outgrew LDS .*?, (\d+) of them the pool-in-LDS tier too", r.stderr)
        assert m and int(m.group(1)) > 1000, r.stderr[-400:]
        if "ILUPP_ILUT_NO_TIER2" in env:
            assert int(m.group(3)) == 0
        else:
            assert 0 < int(m.group(3)) < int(m.group(1))
    assert outs[0] == outs[1]


@pytest.mark.parametrize("shape", [(40, 40, 40), (64, 24, 16), (17, 33, 65), (300, 300)])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_static_path_meshes(shape, fmt):
    """meshes of several aspect ratios (lines longer / shorter than a 16 x 16 patch of lines, 2-D): bit-exact, and where
    the static analysis accepts the structure the static kernels ran"""
    import ilupp_amd as ilupp
    O, ref = _oracle()
    d, i, p = matgen.poisson3d(*shape) if len(shape) == 3 else matgen.poisson2d(*shape)
    n = p.shape[0] - 1
    # nonsymmetric values on the symmetric pattern (the transposed entries really are other numbers)
    rng = np.random.default_rng(7)
    d = d * (1.0 + 0.3 * rng.random(d.shape[0]))
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    A = A if fmt == "csr" else A.tocsc()
    P = ilupp.ILU0Preconditioner(A)
    Lo, Uo = ref.ilu0((A.data, A.indices, A.indptr, fmt == "csr"))
    L, U = P.factors()
    assert G.mat_equal((L.data, L.indices, L.indptr, fmt == "csr"), Lo) and G.mat_equal((U.data, U.indices, U.indptr, fmt == "csr"), Uo)
    b = G.rhs(n)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, O.ID))
    xt = b.copy(); P.apply_trans(xt)
    assert np.array_equal(xt, O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE))


def test_static_path_more_lines_than_lanes():
    """a mesh with more lines (90 000) than the chip has lanes (65 536): one lane per line all the same -- the static kernels take
    workgroups in ticket order and only wait for lower tickets, so the schedule need not be resident at once (before: blocks
    that straddled lines, 600x slower at 260^3, a time-out at 288^3) -- and a grid of more than 131 008 slots (the older
    generations' descriptor limit)"""
    import ilupp_amd as ilupp
    O, ref = _oracle()
    for shape in ((64, 300, 300), (48, 272, 500)):
        d, i, p = matgen.poisson3d(*shape)
        n = p.shape[0] - 1
        rng = np.random.default_rng(11)
        d = d * (1.0 + 0.3 * rng.random(d.shape[0]))
        A = sp.csr_matrix((d, i, p), shape=(n, n))
        P = ilupp.ILU0Preconditioner(A)
        assert P.pr.path().startswith("ilu0:static-")
        Lo, Uo = ref.ilu0((A.data, A.indices, A.indptr, True))
        L, U = P.factors()
        assert G.mat_equal((L.data, L.indices, L.indptr, True), Lo) and G.mat_equal((U.data, U.indices, U.indptr, True), Uo)
        b = G.rhs(n)
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, O.ID))
        xt = b.copy(); P.apply_trans(xt)
        assert np.array_equal(xt, O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE))


def test_more_lines_than_lanes_other_kernels(monkeypatch):
    """the same kind of mesh (78 400 lines) through IChol0, ICholT(0, 0) and the record-decoding generation of ILU(0): array-equal
    to the reference.  (With the chip's 65 536 lanes as the limit of a schedule, blocks straddled lines and these ran into their
    spin limits from 288^3 on: every kernel takes tickets and only waits for lower ones, so the limit never was one.)"""
    import ilupp_amd as ilupp
    O, ref = _oracle()
    d, i, p = matgen.poisson3d(48, 280, 280)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    b = G.rhs(n)
    P = ilupp.IChol0Preconditioner(A)
    Lo = ref.ichol0((A.data, A.indices, A.indptr, True))
    (L,) = P.factors()
    assert G.mat_equal((L.data, L.indices, L.indptr, isinstance(L, sp.csr_matrix)), Lo)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, O.orc().apply_llt(Lo, b, O.ID))
    del P
    P = ilupp.ICholTPreconditioner(A, add_fill_in=0, threshold=0.0)
    Lo = ref.icholt((A.data, A.indices, A.indptr, True), 0, 0.0)
    (L,) = P.factors()
    assert G.mat_equal((L.data, L.indices, L.indptr, isinstance(L, sp.csr_matrix)), Lo)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, O.orc().apply_llt(Lo, b, O.ID))
    del P
    monkeypatch.setenv("ILUPP_NO_STATIC", "1")
    code = """
import sys, numpy as np, scipy.sparse as sp
sys.path[:0] = [%r, %r]
import matgen, golden_util as G, ilupp_amd as ilupp
from oracle import oracle as O
d, i, p = matgen.poisson3d(48, 280, 280)
n = p.shape[0] - 1
A = sp.csr_matrix((d, i, p), shape=(n, n))
P = ilupp.ILU0Preconditioner(A)
assert not P.pr.path().startswith("ilu0:static-"), P.pr.path()
ref = O.ref() if O.ref_available() else O.orc()
Lo, Uo = ref.ilu0((A.data, A.indices, A.indptr, True))
b = G.rhs(n)
for use, f in ((O.ID, P.apply), (O.TRANSPOSE, P.apply_trans)):
    x = b.copy(); f(x)
    assert np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, use))
print("ok", P.pr.path())
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def _mesh_with_holes(g, seed):
    """7-point mesh with random points removed (rows/columns deleted): chains of irregular length, templates that do not hold"""
    d, i, p = matgen.poisson3d(g)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    keep = np.random.default_rng(seed).random(n) > 0.03
    idx = np.flatnonzero(keep)
    B = A[idx][:, idx].tocsr()
    B.sort_indices()
    return B


@pytest.mark.parametrize("case", ["holes", "nonsym_pattern", "wide_rows"])
def test_static_path_rejects_and_falls_back(case):
    """structures outside the static form (irregular chains, a structurally nonsymmetric stencil, 9-point rows) take the
    other kernel generations and stay bit-exact"""
    import ilupp_amd as ilupp
    O, ref = _oracle()
    if case == "holes":
        A = _mesh_with_holes(32, 5)
    elif case == "nonsym_pattern":
        d, i, p = matgen.poisson3d(32)
        n = p.shape[0] - 1
        A = sp.csr_matrix((d, i, p), shape=(n, n)).tolil()
        rng = np.random.default_rng(11)
        for r in rng.integers(40, n - 40, size=400):         # drop some upper entries only
            if A[r, r + 1] != 0:
                A[r, r + 1] = 0
        A = A.tocsr(); A.eliminate_zeros(); A.sort_indices()
    else:
        g = 96
        T = sp.diags([-1.0, 2.5, -1.0], [-1, 0, 1], shape=(g, g))
        E = sp.diags([1.0, 1.0, 1.0], [-1, 0, 1], shape=(g, g))
        A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) - 0.1 * sp.kron(E, E)).tocsr()
        A.sort_indices()
    n = A.shape[0]
    P = ilupp.ILU0Preconditioner(A)
    Lo, Uo = ref.ilu0((A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), True))
    L, U = P.factors()
    assert G.mat_equal((L.data, L.indices, L.indptr, True), Lo) and G.mat_equal((U.data, U.indices, U.indptr, True), Uo)
    b = G.rhs(n)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, O.ID))


@pytest.mark.parametrize("permute", [False, True])
def test_mesh_with_holes_takes_the_level_order_when_its_chains_are_deep(permute):
    """A box mesh with 3 % of its points removed is no box grid: the static form rejects it.  Its dependency levels are as deep as the
    box's (64^3: 190), so the factorisation and both sweeps run by level (round 6: in natural order the same matrix at 256^3 took
    111 s and its sweeps ran into their time limit); the same matrix randomly permuted has few, wide levels and stays on the
    CSR-streaming kernels.  Both bit-identical to the reference's C++ (`ILU0.hpp:26-106`), factors and apply."""
    import ilupp_amd as ilupp
    O, ref = _oracle()
    d, i, p = matgen.mesh_with_holes(64, 5, 0.03, permute=permute)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    P = ilupp.ILU0Preconditioner(A)
    assert P.pr.path() == ("ilu0:csr" if permute else "ilu0:level-order"), P.pr.path()
    Lo, Uo = ref.ilu0((d, i, p, True))
    L, U = P.factors()
    assert G.mat_equal((L.data, L.indices, L.indptr, True), Lo) and G.mat_equal((U.data, U.indices, U.indptr, True), Uo)
    b = G.rhs(n)
    want = O.orc().apply_lu(Lo, Uo, b, O.ID)
    for _ in range(2):                      # (the first apply builds the sweeps' level order, the second runs on it)
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, want)


def test_device_constructors_match_host_constructors():
    """ILUT / IChol0 / ICholT from device-resident inputs (include/ilupp_hip.h *_create_device) = the host-pointer ones"""
    import torch
    from ilupp_amd import _native
    d, i, p = matgen.random_dd(20000, 12, 25.0, 99)
    n = p.shape[0] - 1
    t = _dev(d, i, p)
    torch.cuda.synchronize()
    a = _native.ILUTPreconditioner_device(*_ptrs(t), n, True, 8, 1e-3).factors_info()
    b = _native.ILUTPreconditioner(d, i, p, True, 8, 1e-3).factors_info()
    for fa, fb in zip(a, b):
        assert all(np.array_equal(u, v) for u, v in zip(fa[:3], fb[:3]))
    ds, is_, ps = matgen.poisson3d(20)
    ns = ps.shape[0] - 1
    ts = _dev(ds, is_, ps)
    torch.cuda.synchronize()
    for dev_fn, host_fn, extra in ((_native.IChol0Preconditioner_device, _native.IChol0Preconditioner, ()),
                                   (_native.ICholTPreconditioner_device, _native.ICholTPreconditioner, (3, 1e-3))):
        fa = dev_fn(*_ptrs(ts), ns, True, *extra)
        fb = host_fn(ds, is_, ps, True, *extra)
        assert all(np.array_equal(u, v) for u, v in zip(fa.factors_info()[0][:3], fb.factors_info()[0][:3]))
        x = torch.ones(ns, dtype=torch.float64, device=ts[0].device)
        torch.cuda.synchronize()
        fa.apply_device(x.data_ptr(), ns, transpose=False, sync=True)
        y = np.ones(ns); fb.apply(y)
        assert np.array_equal(x.cpu().numpy(), y)


def test_caller_stream_ordering():
    """with the caller's stream registered, producer work queued on it is seen by create / apply without a host
    synchronisation, and an asynchronous apply's result is ordered before the caller's next work on that stream"""
    import torch
    from ilupp_amd import _native
    O, ref = _oracle()
    d, i, p = matgen.poisson3d(48)
    n = p.shape[0] - 1
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    _native.set_caller_stream(s.cuda_stream, True)
    try:
        with torch.cuda.stream(s):
            td = torch.from_numpy(d).to(dev, non_blocking=True) * 1.0          # produced on s
            ti = torch.from_numpy(i).to(dev); tp = torch.from_numpy(p).to(dev)
            P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
            x = torch.zeros(n, dtype=torch.float64, device=dev)
            x += 1.0                                                            # producer of x on s, no sync
            P.apply_device(x.data_ptr(), n, transpose=False, sync=False)
            y = x * 2.0                                                         # consumer on s, no sync
        s.synchronize()
        P.sync()
    finally:
        _native.set_caller_stream(0, False)
    Lo, Uo = ref.ilu0((d, i, p, True))
    xo = O.orc().apply_lu(Lo, Uo, np.ones(n), O.ID)
    assert np.array_equal(x.cpu().numpy(), xo) and np.array_equal(y.cpu().numpy(), 2.0 * xo)


_REFACTOR_SCRIPT = r'''
import sys
sys.path[:0] = [%(root)r, %(tests)r]
import numpy as np, torch
import matgen
from oracle import oracle as O
from ilupp_amd import _native
dev = torch.device("cuda", 0)
def check(d, i, p, tag):
    n = p.shape[0] - 1
    t = [torch.from_numpy(a).to(dev) for a in (d, i, p)]
    torch.cuda.synchronize()
    P = _native.ILU0Preconditioner_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, True)
    x = torch.ones(n, dtype=torch.float64, device=dev); torch.cuda.synchronize()
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)          # builds whatever sweep forms this object has
    d2 = d * (1.0 + 0.25 * np.cos(np.arange(d.shape[0], dtype=np.float64)))
    t2 = torch.from_numpy(d2).to(dev); torch.cuda.synchronize()
    P.refactor_device(t2.data_ptr(), t[1].data_ptr(), t[2].data_ptr())
    L2, U2 = O.orc().ilu0((d2, i, p, True))
    for tr in (False, True):
        x.fill_(1.0); torch.cuda.synchronize()
        P.apply_device(x.data_ptr(), n, transpose=tr, sync=True)
        assert np.array_equal(x.cpu().numpy(), O.orc().apply_lu(L2, U2, np.ones(n), O.TRANSPOSE if tr else O.ID)), (tag, tr)
    F = P.factors_info()
    assert np.array_equal(F[0][0], L2[0]) and np.array_equal(F[1][0], U2[0]), tag
    print("ok", tag, P.path())
check(*matgen.poisson3d(40), "mesh")
# short L rows, long U rows: only one of the two sweeps has a level-major form
n = 6000
rng = np.random.default_rng(3)
rows, cols, vals = [], [], []
for r in range(n):
    cs = {r}
    if r > 0: cs.add(r - 1)
    if r > 70: cs.add(r - 64)
    for c in rng.integers(r + 1, min(n, r + 400), size=6) if r + 1 < n else []:
        cs.add(int(c))
    for c in sorted(cs):
        rows.append(r); cols.append(c); vals.append(20.0 if c == r else -rng.random())
import scipy.sparse as sp
A = sp.csr_matrix((vals, (rows, cols)), shape=(n, n)); A.sort_indices()
check(A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), "short_L_long_U")
print("refactor ok")
'''


@pytest.mark.parametrize("env", [{}, {"ILUPP_NO_DIRECT": "1"}, {"ILUPP_NO_STATIC": "1"}, {"ILUPP_CLASSIC_ANALYSIS": "1"}, {"ILUPP_NO_PACKED": "1"}],
                         ids=["static_direct", "static_records", "no_static_form", "csr_program_packed_sweeps", "csr_only"])
def test_refactor_on_every_generation(env):
    """numeric re-factorisation (same pattern, new values) followed by apply / apply_trans / factors() on every kernel
    generation, incl. an object of which only one sweep has a level-major form (stale packed values)"""
    code = _REFACTOR_SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "refactor ok" in r.stdout, r.stdout + r.stderr


def test_refactor_rejects_other_objects_and_patterns():
    import torch
    from ilupp_amd import _native
    d, i, p = matgen.poisson3d(16)
    n = p.shape[0] - 1
    t = _dev(d, i, p)
    torch.cuda.synchronize()
    T = _native.ILUTPreconditioner_device(*_ptrs(t), n, True, 5, 0.1)
    with pytest.raises(RuntimeError, match="not an ILU\\(0\\) object"):
        T.refactor_device(*_ptrs(t))
    P = _native.ILU0Preconditioner_device(*_ptrs(t), n, True)
    A2 = sp.csr_matrix((d, i, p), shape=(n, n)).tolil()
    A2[5, 6] = 0.0                                   # one stored entry fewer behind the same object
    A2 = A2.tocsr(); A2.eliminate_zeros(); A2.sort_indices()
    t2 = _dev(A2.data, A2.indices.astype(np.int32), A2.indptr.astype(np.int32))
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="analysed pattern"):
        P.refactor_device(*_ptrs(t2))


def test_error_paths_of_the_reference():
    """IChol.hpp:105-107 "must be in triangular form" (a zero on the diagonal position) and
    sparse_implementation.h:3178-3179 "insufficient memory reserved" do not exist for valid inputs; what can be provoked:
    a structurally missing diagonal in ICholT's input"""
    import ilupp_amd as ilupp
    n = 50
    A = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n)).tolil()
    A[10, 10] = 0.0
    A = A.tocsr(); A.eliminate_zeros(); A.sort_indices()
    with pytest.raises(RuntimeError, match="triangular form"):
        ilupp.ICholTPreconditioner(A)
    with pytest.raises(RuntimeError, match="missing diagonal"):
        ilupp.ILU0Preconditioner(A)


_BATCH_SCRIPT = r'''
import json, os, subprocess, sys
root = %(root)r
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "48", "--no-cpu"],
                   capture_output=True, text=True, timeout=600, cwd=root)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
assert r.returncode == 0 and line, r.stdout + r.stderr
out = json.loads(line[-1])
assert out["n_gpus"] == 2 and out["batch"]["identical_to_single_rank"] and len(out["batch"]["records"]) == 2
print("batch ok")
'''


def test_batched_two_gpus():
    """the N > 1 path on the HIP kernels: bench.py --gpus 2 spawns one rank per GPU, every rank factors its own matrix of the
    batch through ilupp_amd.batched.run_batch, outputs byte-identical to the single-rank run (skipped with < 2 GPUs)"""
    from ilupp_amd import _native
    if _native.lib().ilupp_hip_device_count() < 2:
        pytest.skip("needs 2 GPUs")
    r = subprocess.run([sys.executable, "-c", _BATCH_SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "batch ok" in r.stdout, r.stdout + r.stderr


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 code path where only one GPU is visible: two ranks share GPU 0 over gloo (BENCH_SHARE_GPU=1; the measured runs use
    one rank per GPU over RCCL): both batches -- ILU(0) on shifted meshes, the multilevel preconditioner on n = 10^6 matrices -- gathered
    through run_batch and byte-identical to rank 0's own single-rank run, one JSON line with the whole-job value"""
    import json
    e = dict(os.environ, BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "48", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=e)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(line) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(line[-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["batch"]["identical_to_single_rank"] and len(out["batch"]["records"]) == 2
    assert out["batch_ml"]["identical_to_single_rank"] and [x["matrix"] for x in out["batch_ml"]["records"]] == [0, 1]
    assert out["batch_ml"]["records"][0]["sha256_apply"] != out["batch_ml"]["records"][1]["sha256_apply"]
    # BASELINE config 5 as named: default_configuration(10)
    b10 = out["batch_ml_config10"]
    assert "default_configuration(10)" in b10["what"] and b10["identical_to_single_rank"] and [x["matrix"] for x in b10["records"]] == [0, 1]
    assert b10["records"][0]["sha256_apply"] != b10["records"][1]["sha256_apply"]
    # ... at the size BASELINE.json names: n = 10^6 for both presets
    assert b10["n"] == 1000000 and b10["preset"] == 10 and out["batch_ml"]["n"] == 1000000


def test_bench_single_gpu_line():
    """bench.py's contract on a small grid: one JSON line with roofline, cpu_baseline and an ARRAY parity verdict"""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--grid", "48", "--no-extra"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, r.stdout + r.stderr
    out = json.loads(line[-1])
    assert out["n_gpus"] == 1 and out["parity"]["apply_ones_equal"] is True
    assert out["roofline"]["kernel"] in ("k_ilu0_wa<0, 4, 4>", "k_ilu0_wx", "k_ilu0_sd") and out["roofline"]["frac"] > 0 and out["roofline"]["step_frac"] > 0
    assert out["headline_fraction"] == out["roofline"]["step_frac"] == out["hbm_fraction_factor_plus_apply"]
    assert [ph["name"] for ph in out["roofline"]["phases"]][0] in ("k_ilu0_wa<0, 4, 4>", "k_ilu0_wx", "k_ilu0_sd")
    assert out["cpu_baseline"]["kind"] in ("reference", "port") and out["cpu_baseline"]["apply_paths_agree"]


def test_bench_extras_of_every_kind_of_object():
    """the default run's extra configs build three kinds of objects (single-level ones with a path word, multilevel ones without, the
    speculative ICholT path): one of each through bench.py, with their roofline and path fields"""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--grid", "48", "--no-extra", "--no-cpu",
                        "--config", "C4", "--config", "C5", "--config", "ILUC"], capture_output=True, text=True, timeout=1800, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, r.stdout + r.stderr
    ex = json.loads(line[-1])["extra"]
    assert ex["C4"]["path"] == "icholt:grid-static" and ex["C4"]["roofline"]["kernel"] == "k_icholt_grid" and ex["C4"]["construct_s"] < 0.02
    assert ex["C5"]["path"] is None and ex["C5"]["levels"] >= 1 and ex["C5"]["roofline"]["kernel"] == "k_piluc_df"
    assert ex["ILUC"]["path"] == "iluc" and ex["ILUC"]["roofline"]["frac"] > 0


def test_spmv_device_bit_exact():
    """the CSR product on device tensors = scipy's csr_matvec bit for bit (one lane per row, stored order, from 0;
    reference: sparse_implementation.h:2733-2760), short rows and long ones"""
    import torch
    import ilupp_amd.device as ild
    for A in (sp.csr_matrix((matgen.poisson3d(40)[0], matgen.poisson3d(40)[1], matgen.poisson3d(40)[2]), shape=(64000, 64000)),
              sp.random(5000, 5000, density=0.01, random_state=np.random.default_rng(5), format="csr") + sp.identity(5000, format="csr")):
        A = sp.csr_matrix(A); A.sort_indices()
        n = A.shape[0]
        Ad = ild.DeviceCSR.from_scipy(A)
        x = G.rhs(n)
        y = Ad.matvec(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), A @ x)


def test_device_resident_cg_matches_host_iterates():
    """preconditioned CG with everything in HBM (ilupp_amd.device): the iterate after k steps agrees to 1e-12 with the same
    recurrence run on the host with scipy's product and the oracle's (bit-identical) preconditioner"""
    import torch
    import ilupp_amd.device as ild
    O, ref = _oracle()
    d, i, p = matgen.poisson3d(48)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    b = G.rhs(n)
    Lo = ref.icholt((d, i, p, True), 0, 0.0)
    Mh = lambda r: O.orc().apply_llt(Lo, r, O.ID)
    k = 6
    # host recurrence
    x = np.zeros(n); r = b.copy(); z = Mh(r); pv = z.copy(); rz = r @ z
    for _ in range(k):
        Ap = A @ pv
        alpha = rz / (pv @ Ap)
        x += alpha * pv; r -= alpha * Ap
        z = Mh(r); rzn = r @ z
        pv = z + (rzn / rz) * pv; rz = rzn
    Ad = ild.DeviceCSR.from_scipy(A)
    M = ild.DevicePreconditioner("ICholT", Ad, add_fill_in=0, threshold=0.0)
    xd = ild.cg(Ad, torch.from_numpy(b).cuda(), M, maxiter=k)
    torch.cuda.synchronize()
    xd = xd.cpu().numpy()
    assert np.max(np.abs(xd - x)) <= 1e-12 * np.max(np.abs(x))
    # and it converges: 60 iterations bring the residual down by 1e8
    xs = ild.cg(Ad, torch.from_numpy(b).cuda(), M, maxiter=60).cpu().numpy()
    assert np.linalg.norm(b - A @ xs) <= 1e-8 * np.linalg.norm(b)


_PYBIND_SCRIPT = r'''
import sys
sys.path[:0] = [%(root)r, %(tests)r]
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import golden_util as G, matgen
import ilupp_amd as ilupp
from oracle import oracle as O
assert ilupp._backend.__name__.endswith("_ilupp_hip")
d, i, p = matgen.poisson3d(20)
n = p.shape[0] - 1
for fmt in ("csr", "csc"):
    A = sp.csr_matrix((d, i, p), shape=(n, n)).asformat(fmt)
    M = (A.data, A.indices, A.indptr, fmt == "csr")
    P = ilupp.ILU0Preconditioner(A)
    Lo, Uo = O.orc().ilu0(M)
    L, U = P.factors()
    assert G.mat_equal((L.data, L.indices, L.indptr, fmt == "csr"), Lo) and G.mat_equal((U.data, U.indices, U.indptr, fmt == "csr"), Uo)
    b = G.rhs(n)
    assert np.array_equal(P @ b, O.orc().apply_lu(Lo, Uo, b, O.ID)) and np.array_equal(P.T @ b, O.orc().apply_lu(Lo, Uo, b, O.TRANSPOSE))
    assert repr(P) == "<%%dx%%d ILU0Preconditioner with nnz=%%d, dtype=float64>" %% (n, n, P.total_nnz)
    T = ilupp.ILUTPreconditioner(A, fill_in=5, threshold=0.1)
    Lt, Ut = O.orc().ilut(M, 5, 0.1)
    Lg, Ug = T.factors()
    assert G.mat_equal((Lg.data, Lg.indices, Lg.indptr, fmt == "csr"), Lt) and G.mat_equal((Ug.data, Ug.indices, Ug.indptr, fmt == "csr"), Ut)
    Q = ilupp.ILUCPreconditioner(A, fill_in=5, threshold=0.1)
    Lq, Uq = O.orc().iluc(M, 5, 0.1)
    Lh, Uh = Q.factors()
    assert G.mat_equal((Lh.data, Lh.indices, Lh.indptr, False), Lq) and G.mat_equal((Uh.data, Uh.indices, Uh.indptr, True), Uq)
    assert np.array_equal(Q @ b, O.orc().apply_lu(Lq, Uq, b, O.ID)) and np.array_equal(Q.T @ b, O.orc().apply_lu(Lq, Uq, b, O.TRANSPOSE))
    C = ilupp.ICholTPreconditioner(A, add_fill_in=2, threshold=1e-3)
    Lc, = C.factors()
    assert G.mat_equal((Lc.data, Lc.indices, Lc.indptr, False), O.orc().icholt(M, 2, 1e-3))
    assert G.mat_equal(tuple(getattr(ilupp.ichol0(A), k) for k in ("data", "indices", "indptr")) + (True,), O.orc().ichol0(M))
    # the multilevel class of the shim (binding.cpp:284-298)
    prm = ilupp.iluplusplus_precond_parameter(); prm.default_configuration(1); prm.threshold = 0.02
    ML = ilupp.ILUppPreconditioner(A, params=prm)
    assert type(ML.pr).__module__.endswith("_ilupp_hip") and ML.factors() == []
    Qm = O.orc().ml(M, O.ml_params(0.02))
    assert ML.total_nnz == Qm.total_nnz() and ML.pr.levels() == Qm.levels()
    assert np.array_equal(ML @ b, Qm.apply(b)) and np.array_equal(ML.T @ b, Qm.apply(b, O.TRANSPOSE))
    MLd = ilupp.ILUppPreconditioner(A)                                  # default-constructed parameters: the factorisation with pivoting
    Qd = O.orc().ml(M, O.ml_params(1.0, **O.PIVOTING_DEFAULTS))
    assert MLd.total_nnz == Qd.total_nnz() and np.array_equal(MLd @ b, Qd.apply(b))
    # the two classes with column pivoting of the shim (binding.cpp:313-326, :343-356) next to the ctypes ones and the checker
    from ilupp_amd import _native
    for cls, orc_cls, nat in ((ilupp.ILUCPPreconditioner, O.ILUCP, _native.ILUCPPreconditioner), (ilupp.ILUTPPreconditioner, O.ILUTP, _native.ILUTPPreconditioner)):
        V = cls(A, fill_in=4, threshold=1e-2, piv_tol=1.0)
        assert type(V.pr).__module__.endswith("_ilupp_hip")
        Qv = orc_cls(O.orc(), M, fill_in=4, threshold=1e-2, piv_tol=1.0)
        Nv = nat(*M, 4, 1e-2, 1.0, -1, 10.0)
        pl, pr = V.permutations()
        npl, npr = Nv.permutations()
        assert (pl is None) == (npl is None) and (pr is None) == (npr is None)
        assert np.array_equal(pl if pr is None else pr, Qv.perm)
        assert V.total_nnz == len(Qv.L[0]) + len(Qv.U[0]) and V.pr.zero_pivots == Qv.zero_pivots
        for (fa, fb) in zip(V.pr.factors_info(), Nv.factors_info()):
            assert fa[3:] == fb[3:] and all(np.array_equal(x, y) for x, y in zip(fa[:3], fb[:3]))
        assert np.array_equal(V @ b, Qv.apply(b)) and np.array_equal(V.T @ b, Qv.apply(b, O.TRANSPOSE))
        assert repr(V).startswith("<%%dx%%d %%s with nnz=%%d" %% (n, n, cls.__name__, V.total_nnz))
    x, info = spla.gmres(A, b, M=P, atol=1e-10)
    assert info == 0
try:
    P.apply(np.ones(n + 1))
    raise SystemExit("no size check")
except RuntimeError as e:
    assert "wrong size" in str(e)
# a batch through the compiled shim: the objects the constructor gives one at a time (its own native class, not the ctypes one)
mats = [sp.csr_matrix(matgen.random_dd(600 + 50 * k, 6, 25.0, 500 + k), shape=(600 + 50 * k, 600 + 50 * k)) for k in range(3)]
prm = ilupp.iluplusplus_precond_parameter()
prm.default_configuration(10)
Ps = ilupp.ILUppPreconditioner.batch(mats, params=prm)
for M, Pb in zip(mats, Ps):
    P1 = ilupp.ILUppPreconditioner(M, params=prm)
    assert type(Pb.pr) is type(P1.pr) and type(Pb.pr).__module__.endswith("_ilupp_hip"), type(Pb.pr)
    v = np.random.default_rng(3).random(M.shape[0])
    a1 = v.copy(); P1.pr.apply(a1)
    a2 = v.copy(); Pb.pr.apply(a2)
    assert np.array_equal(a1, a2) and Pb.pr.total_nnz == P1.pr.total_nnz
print("pybind ok")
'''


def test_pybind11_shim_end_to_end():
    """the whole Python surface on top of the compiled pybind11 shim instead of ctypes (ILUPP_AMD_BINDING=pybind)"""
    code = _PYBIND_SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    e = dict(os.environ); e["ILUPP_AMD_BINDING"] = "pybind"
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "pybind ok" in r.stdout, r.stdout + r.stderr


def test_icholt_long_dense_tail_does_not_time_out():
    """fuzz seed 1245: n = 2500, 25 entries per row, add_fill_in = 40 -- the last hundred columns are reached by more than a thousand
    earlier ones each and form a chain that takes seconds; the waiting waves' limit is on the time without progress anywhere, not on
    the wait itself (the factorisation used to end with "dependency wait timed out")"""
    import fuzz_util
    assert fuzz_util.run(1, first_seed=1245, verbose=False) == 0


def test_constructions_from_two_threads():
    """the reference lets two Python threads build preconditioners at once (the GIL is released during a factorisation, binding.cpp:292-294,
    :371); here the constructions of a process take turns inside the library (the pool knows nothing of streams), so whatever the
    threads do, every object is the one a lone construction gives -- several kinds of objects, many rounds, both bindings"""
    import threading
    import ilupp_amd as ilupp
    import matgen
    d, i, p = matgen.poisson3d(40, 36, 30)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    B = sp.csr_matrix(matgen.random_dd(20000, 8, 25.0, 77), shape=(20000, 20000))
    prm = ilupp.iluplusplus_precond_parameter()
    prm.default_configuration(1)
    prm.threshold = 1e-2
    makers = [lambda: ilupp.ILU0Preconditioner(A), lambda: ilupp.ILUTPreconditioner(B, 10, 1e-3), lambda: ilupp.ILUppPreconditioner(B, params=prm),
              lambda: ilupp.ILUCPreconditioner(A, 6, 1e-2), lambda: ilupp.IChol0Preconditioner(A)]
    rhs = {n: np.linspace(1.0, 2.0, n), 20000: np.linspace(1.0, 2.0, 20000)}
    want = [m() @ rhs[m().shape[0]] for m in makers]
    errors = []

    def worker(seed):
        rng = np.random.default_rng(seed)
        try:
            for _ in range(12):
                k = int(rng.integers(0, len(makers)))
                P = makers[k]()
                if not np.array_equal(P @ rhs[P.shape[0]], want[k]):
                    errors.append((seed, k))
        except Exception as e:          # noqa: BLE001
            errors.append((seed, repr(e)))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
