"""ICholT on a matrix that is not positive definite (reference IChol.hpp:78-164, dropping.hpp:8-34).

The reference does not notice: the square root of a negative pivot is NaN, every entry of that column becomes NaN, none of them
passes `|w| > norm * tau` (comparisons with NaN are false), so the column is stored EMPTY; its NaNs reach the diagonals of the rows
it touched, and those columns end the same way.  This library reports `ILUPP_ERR_NOT_SPD` by default; with ILUPP_REFERENCE_NANS=1 it
returns the factor the reference returns -- compared here, array for array, with the reference's own C++ (`oracle/_ref`)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import sys, numpy as np, scipy.sparse as sp
sys.path[:0] = [%(root)r, %(tests)r]
import matgen
from ilupp_amd import _native
from oracle import oracle as O
ref = O.ref() if O.ref_available() else O.orc()
bad = 0
for (shape, row, fill, tau) in (((12, 12, 12), 700, 0, 0.0), ((12, 12, 12), 700, 3, 1e-2), ((30, 20, 10), 2500, 0, 0.0), ((40, 40), 333, 5, 1e-3), ((64, 48, 40), 5 * 64 * 48 + 7 * 64 + 9, 0, 0.0)):
    d, i, p = matgen.poisson3d(*shape) if len(shape) == 3 else matgen.poisson2d(*shape)
    d = d.copy()
    n = p.shape[0] - 1
    for q in range(p[row], p[row + 1]):
        if i[q] == row:
            d[q] = -1.0
    Lo = ref.icholt((d, i, p, True), fill, tau)
    P = _native.ICholTPreconditioner(d, i, p, True, fill, tau)
    (ld, li, lp, _, _, _), = P.factors_info()[:1]
    same = np.array_equal(lp, Lo[2]) and np.array_equal(li, Lo[1]) and np.array_equal(ld, Lo[0], equal_nan=True)
    empty = int(np.sum(np.diff(Lo[2]) == 0))
    print("CASE", shape, fill, tau, "empty columns in the reference's factor:", empty, "same:", same, flush=True)
    bad += 0 if (same and empty > 0) else 1
sys.exit(1 if bad else 0)
'''


def test_reference_nans_switch_gives_the_reference_factor():
    e = dict(os.environ)
    e["ILUPP_REFERENCE_NANS"] = "1"
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count("same: True") == 5, r.stdout
