"""GPU parity tests of round 5's headline path: box-grid matrices take the analysis of ilupp_amd/csrc/grid.hip (row blocks guessed from row 0,
proven for every row by k_grid_check, slot tables / lane templates / skews from the dimensions) and the sweeps with the vector wave
(st_wave.hip: k_sptrsv_wv).  Everything is compared, as arrays, with the oracle's restatement of the reference (ILU0.hpp:26-106,
sparse_implementation.h:4040-4087): the factors' values and index arrays, apply and apply_trans."""
import os
import subprocess
import sys

import numpy as np
import pytest

import matgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unsym(d, seed):
    """values that are not symmetric and not constant along the lines (the pattern stays the stencil's)"""
    rng = np.random.default_rng(seed)
    return d * (1.0 + 0.25 * rng.random(d.shape[0]))


def _check_against_oracle(a, want_analysis):
    from oracle import oracle as O
    from ilupp_amd import _native
    d, i, p = a
    n = p.shape[0] - 1
    P = _native.ILU0Preconditioner(d, i, p, True)
    assert P.path() == "ilu0:static-direct" and P.analysis_path() == want_analysis
    L, U = O.orc().ilu0((d, i, p, True))
    (ld, li, lp, _, _, _), (ud, ui, up, _, _, _) = P.factors_info()
    for got, want in ((lp, L[2]), (li, L[1]), (up, U[2]), (ui, U[1])):
        assert np.array_equal(got, want)
    assert np.array_equal(ld, L[0]) and np.array_equal(ud, U[0])
    rng = np.random.default_rng(11)
    b = rng.random(n)
    x = b.copy(); P.apply(x)
    y = O.orc().trisolve(U, O.UPPER, O.ID, O.orc().trisolve(L, O.LOWER, O.ID, b))
    assert np.array_equal(x, y)
    xt = b.copy(); P.apply_trans(xt)
    yt = O.orc().trisolve(L, O.LOWER, O.TRANSPOSE, O.orc().trisolve(U, O.UPPER, O.TRANSPOSE, b))
    assert np.array_equal(xt, yt)
    # a second apply (armed behind the first one's wait) and an apply on another vector give the same bits
    x2 = b.copy(); P.apply(x2)
    assert np.array_equal(x2, y)
    return P


@pytest.mark.parametrize("dims", [(64, 48, 40), (100, 37, 19), (33, 50, 40), (17, 64, 64), (1024, 700), (129, 520)])
def test_box_grids_take_the_grid_analysis(dims):
    """full and partial 16 x 16 patches, odd line lengths (the vector wave's 8-byte tail stores), 2-D grids (identity placement)"""
    d, i, p = matgen.poisson3d(*dims) if len(dims) == 3 else matgen.poisson2d(*dims)
    _check_against_oracle((_unsym(d, 7), i, p), "grid")


def test_small_and_thin_grids_keep_the_general_analysis():
    """below 2^16 rows, lines shorter than 16 rows or fewer than 512 lines: the general pass (grid_guess declines)"""
    for dims in ((33, 20, 30), (8, 100, 100), (300, 300), (301, 300)):
        d, i, p = matgen.poisson3d(*dims) if len(dims) == 3 else matgen.poisson2d(*dims)
        _check_against_oracle((_unsym(d, 3), i, p), "general")


def test_a_matrix_that_only_begins_like_a_grid():
    """row 0 and the entry count say 64 x 48 x 40; one interior row has a column moved (sorted, same count): k_grid_check's verdict drops
    everything built on the guess and the general pass runs -- the result is the reference's for THIS matrix"""
    from oracle import oracle as O
    from ilupp_amd import _native
    d, i, p = matgen.poisson3d(64, 48, 40)
    i = i.copy()
    r = 20 * 64 * 48 + 24 * 64 + 32
    row = i[p[r]:p[r + 1]].copy()
    assert row[3] == r and row.shape[0] == 7
    row[1] += 1                                    # column r - nx -> r - nx + 1
    i[p[r]:p[r + 1]] = row
    d = _unsym(d, 5)
    P = _native.ILU0Preconditioner(d, i, p, True)
    assert P.analysis_path() == "general"
    L, U = O.orc().ilu0((d, i, p, True))
    (ld, li, lp, _, _, _), (ud, ui, up, _, _, _) = P.factors_info()
    assert np.array_equal(li, L[1]) and np.array_equal(ui, U[1]) and np.array_equal(ld, L[0]) and np.array_equal(ud, U[0])
    b = np.random.default_rng(2).random(p.shape[0] - 1)
    x = b.copy(); P.apply(x)
    assert np.array_equal(x, O.orc().trisolve(U, O.UPPER, O.ID, O.orc().trisolve(L, O.LOWER, O.ID, b)))
    # ... and a wrong LAST row pointer / a missing diagonal in the grid's place are not taken for the grid either
    d2, i2, p2 = matgen.poisson3d(64, 48, 40)
    i2 = i2.copy()
    q = p2[r] + 3
    i2[q] = r + 1; i2[q + 1] = r + 2                # the diagonal of row r replaced (columns stay sorted)
    with pytest.raises(RuntimeError, match="missing diagonal"):
        _native.ILU0Preconditioner(d2, i2, p2, True)


_SWITCH_SCRIPT = r"""
import hashlib, sys
sys.path[:0] = [%r, %r]
import numpy as np, matgen
from ilupp_amd import _native
h = hashlib.sha256()
for dims in ((100, 37, 19), (64, 64, 64)):
    d, i, p = matgen.poisson3d(*dims)
    d = d * (1.0 + 0.25 * np.random.default_rng(7).random(d.shape[0]))
    P = _native.ILU0Preconditioner(d, i, p, True)
    for f in P.factors_info():
        for a in f[:3]:
            h.update(np.ascontiguousarray(a).tobytes())
    x = np.random.default_rng(1).random(p.shape[0] - 1)
    y = x.copy(); P.apply(y); h.update(y.tobytes())
    y = x.copy(); P.apply(y); h.update(y.tobytes())
    print(P.analysis_path(), ";".join(P.kernel_names()))
print(h.hexdigest())
"""


def _run_with(env):
    r = subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, **env), cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return lines[:-1], lines[-1]


def test_every_switch_of_the_grid_path_gives_the_same_bits():
    """the general pass (ILUPP_NO_GRID), the general lane-table kernels on the grid's row blocks (ILUPP_GRID_TABLES=0, ILUPP_GRID_LINK=0), the
    waiting construction (ILUPP_NO_SPEC), the proof at its three places (ILUPP_GRID_CHECK_AT), the sweeps through k_st_vec
    (ILUPP_NO_VECWAVE) and an unarmed apply (ILUPP_NO_ARM): one digest over factors and applies"""
    info, ref = _run_with({})
    assert all(l.startswith("grid") and "k_sptrsv_wv<1, false>" in l for l in info), info
    for env in ({"ILUPP_NO_GRID": "1"}, {"ILUPP_GRID_TABLES": "0"}, {"ILUPP_GRID_LINK": "0"}, {"ILUPP_NO_SPEC": "1"}, {"ILUPP_GRID_CHECK_AT": "1"},
                {"ILUPP_GRID_CHECK_AT": "2"}, {"ILUPP_NO_VECWAVE": "1"}, {"ILUPP_NO_ARM": "1"}, {"ILUPP_NO_COMPACT_L": "1"}):
        info2, dig = _run_with(env)
        assert dig == ref, (env, info2)
        if "ILUPP_NO_GRID" in env:
            assert all(l.startswith("general") for l in info2)
        if "ILUPP_NO_VECWAVE" in env:
            assert all("k_sptrsv_wx<1, false>" in l for l in info2)


def test_device_resident_construction_and_the_head_read():
    """ilupp_hip_ilu0_create_device: indptr[n] and the head of the matrix come back in one read-back; same object as from host arrays"""
    import torch
    from ilupp_amd import _native
    d, i, p = matgen.poisson3d(96, 64, 48)
    d = _unsym(d, 9)
    n = p.shape[0] - 1
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    torch.cuda.synchronize()
    Pd = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    Ph = _native.ILU0Preconditioner(d, i, p, True)
    assert Pd.analysis_path() == Ph.analysis_path() == "grid" and Pd.total_nnz == Ph.total_nnz
    x = torch.from_numpy(np.random.default_rng(4).random(n)).to(dev)
    xh = x.cpu().numpy().copy()
    for rep in range(3):                         # (the second and third apply start armed)
        Pd.apply_device(x.data_ptr(), n, transpose=False, sync=True)
        Ph.apply(xh)
        assert np.array_equal(x.cpu().numpy(), xh)
    # re-factorisation with new values on the analysed pattern, then an apply (the exchange buffers are re-armed)
    td2 = td * 1.5
    Pd.refactor_device(td2.data_ptr(), ti.data_ptr(), tp.data_ptr())
    P2 = _native.ILU0Preconditioner(d * 1.5, i, p, True)
    xb = np.random.default_rng(6).random(n)
    x = torch.from_numpy(xb).to(dev)
    Pd.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    y = xb.copy(); P2.apply(y)
    assert np.array_equal(x.cpu().numpy(), y)


def test_construction_with_the_number_of_entries_handed_in():
    """ilupp_hip_ilu0_create_device_nnz: a shape remembered from an earlier proven grid is guessed again without reading the head (the proof
    still covers every row and the last row pointer); a matrix of the same size that is no grid falls back to the reading way; a wrong
    count is refused"""
    import torch
    from ilupp_amd import _native
    d, i, p = matgen.poisson3d(80, 64, 32)
    d = _unsym(d, 13)
    n, nnz = p.shape[0] - 1, int(p[-1])
    dev = torch.device("cuda", 0)
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    torch.cuda.synchronize()
    Ph = _native.ILU0Preconditioner(d, i, p, True)                     # (remembers the shape)
    assert Ph.analysis_path() == "grid"
    b = np.random.default_rng(8).random(n)
    want = b.copy(); Ph.apply(want)
    for rep in range(2):
        Pd = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, nnz=nnz)
        assert Pd.analysis_path() == "grid" and Pd.total_nnz == Ph.total_nnz
        x = torch.from_numpy(b).to(dev)
        Pd.apply_device(x.data_ptr(), n, transpose=False, sync=True)
        assert np.array_equal(x.cpu().numpy(), want)
        for f, g in zip(Pd.factors_info(), Ph.factors_info()):
            assert all(np.array_equal(a, c) for a, c in zip(f[:3], g[:3]))
    # same n and nnz, one interior row with a column moved: the recalled guess fails its proof, the reading way takes the general pass
    i2 = i.copy()
    r = 10 * 80 * 64 + 20 * 80 + 40
    i2[p[r] + 1] += 1
    ti2 = torch.from_numpy(i2).to(dev)
    torch.cuda.synchronize()
    P2 = _native.ILU0Preconditioner_device(td.data_ptr(), ti2.data_ptr(), tp.data_ptr(), n, True, nnz=nnz)
    Pg = _native.ILU0Preconditioner(d, i2, p, True)
    assert P2.analysis_path() == Pg.analysis_path() == "general"
    x = torch.from_numpy(b).to(dev)
    P2.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    y = b.copy(); Pg.apply(y)
    assert np.array_equal(x.cpu().numpy(), y)
    # (the failed proof dropped the remembered shape; the grid itself is taken for a grid again, the reading way)
    P3 = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, nnz=nnz)
    assert P3.analysis_path() == "grid"
    # a count that is not indptr[n]
    with pytest.raises(RuntimeError, match="number of stored entries"):
        _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, nnz=nnz - 7)


def test_two_grids_of_equal_size_and_entry_count_keep_their_shapes():
    """64 x 32 x 40 and 32 x 64 x 40 have the same n and nnz: the remembered shapes are kept per index array, so that
    alternating between the two neither evicts the other's shape nor ends anywhere but on the grid analysis with the right bits"""
    import torch
    from ilupp_amd import _native
    dev = torch.device("cuda", 0)
    mats = []
    for dims in ((64, 32, 40), (32, 64, 40)):
        d, i, p = matgen.poisson3d(*dims)
        d = _unsym(d, 3)
        n, nnz = p.shape[0] - 1, int(p[-1])
        t = tuple(torch.from_numpy(a).to(dev) for a in (d, i, p))
        Ph = _native.ILU0Preconditioner(d, i, p, True)
        b = np.random.default_rng(2).random(n)
        want = b.copy(); Ph.apply(want)
        mats.append((t, n, nnz, b, want))
    assert mats[0][1] == mats[1][1] and mats[0][2] == mats[1][2]
    torch.cuda.synchronize()
    for rep in range(3):
        for (t, n, nnz, b, want) in mats:
            P = _native.ILU0Preconditioner_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, True, nnz=nnz)
            assert P.analysis_path() == "grid"
            x = torch.from_numpy(b).to(dev)
            P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
            assert np.array_equal(x.cpu().numpy(), want)
