"""Helpers shared by the golden-vector tests (CPU oracle tests and GPU parity tests)."""
import hashlib
import json
import os

import numpy as np

import matgen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_cache = {}


def load(name):
    if name not in _cache:
        if name.endswith(".json"):
            with open(os.path.join(GOLDEN, name)) as f:
                _cache[name] = json.load(f)
        else:
            _cache[name] = np.load(os.path.join(GOLDEN, name))
    return _cache[name]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def get_mat(z, key):
    return (z[key + "_data"], z[key + "_indices"], z[key + "_indptr"], bool(int(z[key + "_is_csr"])))


def has_mat(z, key):
    return (key + "_data") in z.files


def mat_equal(a, b):
    """bit-exact on indices/indptr/orientation AND on values (NaN == NaN)."""
    return (bool(a[3]) == bool(b[3]) and np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])
            and np.array_equal(np.asarray(a[0]), np.asarray(b[0]), equal_nan=True))


def mat_close(a, b, rtol=1e-12):
    """indices bit-exact, values within rtol relative (north_star tolerance: 1e-12)."""
    if not (bool(a[3]) == bool(b[3]) and np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])):
        return False
    x, y = np.asarray(a[0]), np.asarray(b[0])
    return bool(np.all(np.abs(x - y) <= rtol * np.abs(y)) or np.array_equal(x, y, equal_nan=True))


def digest_of(M):
    return {"data": sha(M[0]), "indices": sha(M[1]), "indptr": sha(M[2]), "is_csr": bool(M[3]),
            "nnz": int(M[2][-1])}


def rhs(n):
    return 1.0 + (np.arange(n, dtype=np.float64) % 17) / 16.0 - (np.arange(n, dtype=np.float64) % 5) / 8.0


CONFIG_CASES = {
    "poisson2d_20": lambda: matgen.poisson2d(20),
    "poisson3d_8": lambda: matgen.poisson3d(8),
    "poisson3d_16": lambda: matgen.poisson3d(16),
    "poisson3d_5x7x3": lambda: matgen.poisson3d(5, 7, 3),
    "random_dd_2000": lambda: matgen.random_dd(2000, 19, 25.0, 12345),
    "random_dd_300_k6": lambda: matgen.random_dd(300, 6, 4.0, 99),
}

CONFIG_ILUT = ((5, 0.1), (10, 1e-4), (1, 0.0), (3, 0.0))
REFTEST_ILUT = ((5, 0.1), (100, 0.0), (10, 1e-4))
ICHOLT = ((0, 0.0), (5, 1e-3), (2, 0.05))


def config_inputs(name, fmt):
    """(M, S): general and symmetrised input of a config case in the given format; checks the
    generator still produces the bytes the golden file was made from."""
    d, i, p = CONFIG_CASES[name]()
    z = load("configs.npz")
    want = bytes(z[name + "/input_sha"]).hex()
    assert want == sha(d) + sha(i) + sha(p), "tests/matgen.py drifted from the golden inputs"
    sd, si, spp = matgen.symmetrize(d, i, p)
    if fmt == "csr":
        return (d, i, p, True), (sd, si, spp, True)
    return matgen.to_csc(d, i, p) + (False,), matgen.to_csc(sd, si, spp) + (False,)
