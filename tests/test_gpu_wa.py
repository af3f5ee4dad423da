"""GPU parity tests of round 6's factor kernel (ilupp_amd/csrc/st_wave.hip: k_ilu0_wa; reference ILU0.hpp:26-106): the wave-exchange
factor kernel fed by LDS-DMA, without a barrier in its loop, with prefetcher waves reading ahead for the tiles at work.

* every variant of it gives the bits of the oracle AND of the kernel it replaces (k_ilu0_wx, ILUPP_NO_WA=1): the default, without
  the prefetchers (ILUPP_NO_PREFETCH=1), with the replays (ILUPP_REPLAY=1: the chains store a quarter of the records, workgroups whose
  tile has ended write the rest), with tiles handed out by XCD (ILUPP_XCD_TICKETS=1);
* on box grids whose patches are cut by the domain, with more tiles than the chip holds workgroups (the prefetchers must not stay
  behind their tile's end then), 2-D and 3-D, CSR and CSC, nonsymmetric values, a value array that is only 8-byte aligned;
* the kernel the object reports (ilupp_hip_kernel_names) is the one that ran.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import sys, hashlib, numpy as np, scipy.sparse as sp
sys.path[:0] = [%(root)r, %(tests)r]
import torch
import matgen, ilupp_amd as ilupp
from ilupp_amd import _native
from oracle import oracle as O
rng = np.random.default_rng(17)
big = sys.argv[1] == "big"
cases = []
shapes = ((24, 24, 24), (70, 45, 37), (100, 60, 40), (17, 33, 65), (40, 272, 272)) if not big else ((96, 160, 144),)
for shape in shapes:
    d, i, p = matgen.poisson3d(*shape)
    cases.append(("7pt%%dx%%dx%%d" %% shape, sp.csr_matrix((d * (1.0 + 0.3 * rng.random(d.shape[0])), i, p))))
if not big:
    d, i, p = matgen.poisson2d(300, 300)
    cases.append(("5pt", sp.csr_matrix((d * (1.0 + 0.3 * rng.random(d.shape[0])), i, p))))
if not big:
    d, i, p = matgen.poisson3d(33, 20, 50)
    cases.append(("csc", sp.csr_matrix((d * (1.0 + 0.3 * rng.random(d.shape[0])), i, p)).tocsc()))
for name, A in cases:
    P = ilupp.ILU0Preconditioner(A)
    L, U = P.factors()
    h = hashlib.sha256()
    for M in (L, U):
        for a in (M.data, M.indices, M.indptr):
            h.update(np.ascontiguousarray(a).tobytes())
    x = np.linspace(-1.0, 2.0, A.shape[0])
    y = x.copy(); P.apply(y)
    h.update(y.tobytes())
    names = ";".join(P.pr.kernel_names()) if hasattr(P.pr, "kernel_names") else ""
    ok = ""
    if A.shape[0] <= 300000 and sp.isspmatrix_csr(A):
        Lo, Uo = O.orc().ilu0((A.data, A.indices, A.indptr, True))
        ok = " oracle=%%s" %% (np.array_equal(Lo[0], L.data) and np.array_equal(Uo[0], U.data))
    print("CASE %%s %%s path=%%s kernels=%%s%%s" %% (name, h.hexdigest()[:24], P.pr.path() if hasattr(P.pr, "path") else "", names, ok), flush=True)
# a value array that is only 8-byte aligned (the DMA windows are 16-byte pieces)
d, i, p = matgen.poisson3d(40, 40, 40)
d = d * (1.0 + 0.3 * rng.random(d.shape[0]))
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
buf = torch.zeros(d.shape[0] + 1, dtype=torch.float64, device=dev)
buf[1:] = torch.from_numpy(d).to(dev)
ti, tp = torch.from_numpy(i).to(dev), torch.from_numpy(p).to(dev)
P = _native.ILU0Preconditioner_device(buf[1:].data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
fi = P.factors_info()
h = hashlib.sha256()
for k in range(2):
    h.update(np.ascontiguousarray(fi[k][0]).tobytes())
print("CASE misaligned %%s path=%%s kernels=%%s" %% (h.hexdigest()[:24], P.path(), ";".join(P.kernel_names())), flush=True)
'''

_VARIANTS = {
    "default": {},
    "old": {"ILUPP_NO_WA": "1"},
    "no-prefetch": {"ILUPP_NO_PREFETCH": "1"},
    "replay": {"ILUPP_REPLAY": "1"},
    "xcd": {"ILUPP_XCD_TICKETS": "1"},
    "format1": {"ILUPP_NO_COMPACT_L": "1"},      # L's records as {lA, 1} pairs (round 6's default leaves the constant out: format 2)
}


def _run(variant, size="small"):
    code = _SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    e = dict(os.environ)
    e.update(_VARIANTS[variant])
    r = subprocess.run([sys.executable, "-c", code, size], env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("CASE "):
            f = ln.split()
            out[f[1]] = (f[2], ln)
    return out


@pytest.fixture(scope="module")
def reference_run():
    return _run("old")


@pytest.mark.parametrize("variant", ["default", "no-prefetch", "replay", "xcd", "format1"])
def test_every_variant_gives_the_old_kernels_bits(variant, reference_run):
    got = _run(variant)
    assert set(got) == set(reference_run)
    for name, (digest, line) in got.items():
        assert digest == reference_run[name][0], (variant, line, reference_run[name][1])
        assert "oracle=False" not in line, line
    # the kernel the objects name is the variant's
    some = next(v[1] for k, v in got.items() if k.startswith("7pt"))
    want = {"default": "k_ilu0_wa<0, 4, 4>", "no-prefetch": "k_ilu0_wa<0, 4, 4>", "xcd": "k_ilu0_wa<0, 4, 4>", "replay": "k_ilu0_wa<2, 4, 4>", "format1": "k_ilu0_wa<0, 4, 4>"}[variant]
    assert want in some, some


def test_old_kernel_is_named_when_it_runs(reference_run):
    assert any("k_ilu0_wx" in v[1] for v in reference_run.values())


def test_medium_grid_default_against_old():
    a, b = _run("default", "big"), _run("old", "big")
    assert {k: v[0] for k, v in a.items()} == {k: v[0] for k, v in b.items()}


def test_random_box_shapes_predicted_sizes_and_edge_tiles():
    """Box grids of random dimensions (patches cut by the domain in y and z, lines of 16 .. 90 rows, more or fewer tiles than the chip
    has workgroups): the closed-form analysis' predicted sizes must be what the device finds (a mismatch redoes the construction the
    general way and shows as another path), and factors + apply are the oracle's bits.  ILUPP_FUZZ_OFFSET shifts the seed."""
    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import matgen
    import scipy.sparse as sp
    import ilupp_amd as ilupp
    from oracle import oracle as O
    rng = np.random.default_rng(4242 + int(os.environ.get("ILUPP_FUZZ_OFFSET", "0")))
    done = 0
    while done < 10:
        nx, ny, nz = int(rng.integers(16, 91)), int(rng.integers(8, 71)), int(rng.integers(8, 71))
        if nx * ny * nz < (1 << 16) or ny * nz < 512:
            continue
        done += 1
        d, i, p = matgen.poisson3d(nx, ny, nz)
        d = d * (1.0 + 0.3 * rng.random(d.shape[0]))
        n = p.shape[0] - 1
        A = sp.csr_matrix((d, i, p), shape=(n, n))
        P = ilupp.ILU0Preconditioner(A)
        assert P.pr.path() == "ilu0:static-direct", ((nx, ny, nz), P.pr.path())
        assert P.pr.analysis_path() == "grid", ((nx, ny, nz), P.pr.analysis_path())
        # (fewer than 16 lines in y: the z-neighbour is not lane - 16, the wave-exchange kernels decline and k_ilu0_sd runs)
        assert P.pr.kernel_names()[0] == ("k_ilu0_wa<0, 4, 4>" if ny >= 16 else "k_ilu0_sd"), ((nx, ny, nz), P.pr.kernel_names())
        Lo, Uo = O.orc().ilu0((d, i, p, True))
        L, U = P.factors()
        assert np.array_equal(L.indptr, Lo[2]) and np.array_equal(L.indices, Lo[1]) and np.array_equal(U.indptr, Uo[2]) and np.array_equal(U.indices, Uo[1]), (nx, ny, nz)
        assert np.array_equal(L.data, Lo[0]) and np.array_equal(U.data, Uo[0]), (nx, ny, nz)
        b = rng.random(n)
        x = b.copy(); P.apply(x)
        assert np.array_equal(x, O.orc().apply_lu(Lo, Uo, b, O.ID)), (nx, ny, nz)
