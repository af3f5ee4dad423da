"""GPU parity tests of the speculative static path of ICholT(add_fill_in = 0, threshold = 0) on box grids (ilupp_amd/csrc/icholt_grid.hip:
the recurrence over A's pattern plus one-step fill on the wavefront x + 2y + 3z, every column's cut verified, anything else rebuilt by
the dataflow kernel).  Everything is compared, as arrays, with the oracle's restatement of the reference (IChol.hpp:78-155,
dropping.hpp:8-34, ILUC.hpp:37-63)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import matgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unsym(d, seed):
    """values that differ from entry to entry (ICholT reads the upper triangle only; the pattern stays the stencil's)"""
    rng = np.random.default_rng(seed)
    return d * (1.0 + 0.25 * rng.random(d.shape[0]))


def _check(a, want_path, fill=0, tau=0.0):
    from oracle import oracle as O
    from ilupp_amd import _native
    d, i, p = a
    n = p.shape[0] - 1
    P = _native.ICholTPreconditioner(d, i, p, True, fill, tau)
    assert P.path() == want_path
    L = O.orc().icholt((d, i, p, True), fill, tau)
    (ld, li, lp, _, _, _), = P.factors_info()[:1]
    assert np.array_equal(lp, L[2]) and np.array_equal(li, L[1])
    assert np.array_equal(ld, L[0])
    b = np.random.default_rng(11).random(n)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, O.orc().apply_llt(L, b))
    xt = b.copy(); P.apply_trans(xt)
    assert np.array_equal(xt, O.orc().apply_llt(L, b, O.TRANSPOSE))
    return P


@pytest.mark.parametrize("dims", [(64, 48, 40), (100, 37, 19), (33, 50, 40), (17, 64, 64), (16, 70, 66), (129, 520), (300, 600)])
def test_box_grids_take_the_static_path(dims):
    """full and partial (sheared) patches, several patches in y and z, 2-D grids"""
    d, i, p = matgen.poisson3d(*dims) if len(dims) == 3 else matgen.poisson2d(*dims)
    _check((_unsym(d, 7), i, p), "icholt:grid-static")
    _check((d, i, p), "icholt:grid-static")


def test_other_parameters_and_small_grids_keep_the_general_path():
    d, i, p = matgen.poisson3d(64, 48, 40)
    _check((_unsym(d, 3), i, p), "icholt", fill=1)
    _check((_unsym(d, 3), i, p), "icholt", tau=1e-3)
    d, i, p = matgen.poisson3d(20, 20, 20)
    _check((_unsym(d, 3), i, p), "icholt")


def _entry(i, p, r, c):
    q = p[r] + int(np.searchsorted(i[p[r]:p[r + 1]], c))
    assert i[q] == c
    return q


def test_a_column_where_fill_wins_the_cut_is_rebuilt():
    """one entry of A made so small that a fill entry of its column is larger: the cut of dropping.hpp keeps the fill entry, the premise
    of the static kernel is violated (its verdict word), the dataflow kernel builds the object -- the reference's factor"""
    nx, ny, nz = 64, 48, 40
    d, i, p = matgen.poisson3d(nx, ny, nz)
    d = _unsym(d, 5)
    r = 17 * nx * ny + 20 * nx + 30
    d = d.copy()
    d[_entry(i, p, r, r + nx)] *= 1e-9              # a(j, j+nx): |e2| falls below the fill entries of column j
    P = _check((d, i, p), "icholt")
    (ld, li, lp, _, _, _), = P.factors_info()[:1]
    assert (r + nx) not in li[lp[r]:lp[r + 1]]      # (the column really lost that entry to a fill entry)
    # an entry of A that is exactly zero is no candidate of the cut at all
    d2 = _unsym(matgen.poisson3d(nx, ny, nz)[0], 5)
    d2[_entry(i, p, r, r + 1)] = 0.0
    _check((d2, i, p), "icholt")


def test_not_positive_definite_reports_as_before():
    from ilupp_amd import _native
    d, i, p = matgen.poisson3d(64, 48, 40)
    d = d.copy()
    r = 5 * 64 * 48 + 7 * 64 + 9
    d[_entry(i, p, r, r)] = -1.0
    with pytest.raises(RuntimeError, match="not positive definite"):
        _native.ICholTPreconditioner(d, i, p, True, 0, 0.0)


def test_a_matrix_that_only_begins_like_a_grid():
    d, i, p = matgen.poisson3d(64, 48, 40)
    i = i.copy()
    r = 20 * 64 * 48 + 24 * 64 + 32
    i[p[r] + 1] += 1                               # column r - nx -> r - nx + 1 (sorted, same count): k_grid_check's verdict
    _check((_unsym(d, 5), i, p), "icholt")


def test_device_resident_construction():
    import torch
    from ilupp_amd import _native
    d, i, p = matgen.poisson3d(96, 64, 48)
    d = _unsym(d, 9)
    n = p.shape[0] - 1
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    torch.cuda.synchronize()
    Pd = _native.ICholTPreconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, 0, 0.0)
    Ph = _native.ICholTPreconditioner(d, i, p, True, 0, 0.0)
    assert Pd.path() == Ph.path() == "icholt:grid-static"
    for f, g in zip(Pd.factors_info(), Ph.factors_info()):
        assert all(np.array_equal(a, c) for a, c in zip(f[:3], g[:3]))
    x = torch.from_numpy(np.random.default_rng(4).random(n)).to(dev)
    xh = x.cpu().numpy().copy()
    Pd.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    Ph.apply(xh)
    assert np.array_equal(x.cpu().numpy(), xh)


_SWITCH_SCRIPT = r"""
import hashlib, sys
sys.path[:0] = [%r, %r]
import numpy as np, matgen
from ilupp_amd import _native
h = hashlib.sha256()
for dims in ((100, 37, 19), (64, 64, 64)):
    d, i, p = matgen.poisson3d(*dims)
    d = d * (1.0 + 0.25 * np.random.default_rng(7).random(d.shape[0]))
    P = _native.ICholTPreconditioner(d, i, p, True, 0, 0.0)
    for f in P.factors_info():
        for a in f[:3]:
            h.update(np.ascontiguousarray(a).tobytes())
    x = np.random.default_rng(1).random(p.shape[0] - 1)
    y = x.copy(); P.apply(y); h.update(y.tobytes())
    y = x.copy(); P.apply_trans(y); h.update(y.tobytes())
    print(P.path())
print(h.hexdigest())
"""


def _run_with(env):
    r = subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, **env), cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return lines[:-1], lines[-1]


def test_the_static_and_the_general_path_give_the_same_bits():
    info, ref = _run_with({})
    assert info == ["icholt:grid-static"] * 2
    info2, dig = _run_with({"ILUPP_NO_ICHOLT_GRID": "1"})
    assert info2 == ["icholt"] * 2 and dig == ref


def test_the_schedule_from_the_dimensions_is_the_general_one():
    """ILUPP_IG_SCHED=verify builds the sweeps' schedule both ways (from the grid's dimensions; by the pass over L's pattern) and
    compares blocks, starts, patches, the longest row; =general uses the pass.  Same factor, same applies either way."""
    info, ref = _run_with({})
    for mode in ("verify", "general"):
        info2, dig = _run_with({"ILUPP_IG_SCHED": mode})
        assert info2 == info and dig == ref, mode
