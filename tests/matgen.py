"""Deterministic synthetic sparse matrices for tests, golden fixtures and bench.py.

Pure numpy (no scipy RNG: ``scipy.sparse.random``'s stream is version dependent), so the
same arrays come out in the build container and on the GPU box.  A matrix is returned as
``(data f64, indices i32, indptr i32)`` in CSR with sorted column indices.

Shapes follow BASELINE.json's configs:
  poisson2d(nx, ny)   5-point  Laplacian, diag 4, off-diag -1           (C1: 200x200)
  poisson3d(g)        7-point  Laplacian, diag 6, off-diag -1, lexicographic (C2/C4: g=256)
  random_dd(n, k, d)  k pseudo-random off-diagonals U[0,1) per row + diagonal d (C3: n=1e6, k=19, d=25)
"""
import numpy as np


def _stencil(dims, diag):
    """Generic lexicographic Laplacian stencil on a box grid, built without Python loops."""
    dims = tuple(int(d) for d in dims)
    n = int(np.prod(dims))
    nd = len(dims)
    strides = [1]
    for d in dims[:-1]:
        strides.append(strides[-1] * d)          # x fastest
    ii = np.arange(n, dtype=np.int64)
    coords = []
    rem = ii
    for d in dims:
        coords.append(rem % d)
        rem = rem // d
    # candidate columns in ascending order: -s_{nd-1}, ..., -s_0, 0, +s_0, ..., +s_{nd-1}
    offs, valid = [], []
    for a in reversed(range(nd)):
        offs.append(-strides[a]); valid.append(coords[a] > 0)
    offs.append(0); valid.append(np.ones(n, dtype=bool))
    for a in range(nd):
        offs.append(strides[a]); valid.append(coords[a] < dims[a] - 1)
    valid = np.stack(valid, axis=1)                         # n x (2nd+1)
    counts = valid.sum(axis=1)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    cols = (ii[:, None] + np.asarray(offs, dtype=np.int64)[None, :])[valid]
    vals = np.broadcast_to(np.where(np.asarray(offs) == 0, float(diag), -1.0)[None, :], valid.shape)[valid]
    return np.ascontiguousarray(vals, dtype=np.float64), cols.astype(np.int32), indptr.astype(np.int32)


def poisson2d(nx, ny=None):
    ny = nx if ny is None else ny
    return _stencil((nx, ny), 4.0)


def poisson3d(g, gy=None, gz=None):
    gy = g if gy is None else gy
    gz = g if gz is None else gz
    return _stencil((g, gy, gz), 6.0)


def laplace1d(n, scaled=True):
    """tridiag(-1,2,-1)*(n+1)^2 as in the reference's tests (test/tests.py:9-12)."""
    s = float((n + 1) ** 2) if scaled else 1.0
    data, indices, indptr = _stencil((n,), 2.0)
    return data * s, indices, indptr


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def random_dd(n, k=19, diag=25.0, seed=12345, symmetric_pattern=False):
    """Diagonally dominant pseudo-random CSR: per row up to k off-diagonals with values in [0,1),
    plus ``diag + u`` on the diagonal (duplicates removed), hash-based so it is reproducible anywhere."""
    n = int(n)
    with np.errstate(over="ignore"):
        rows = np.repeat(np.arange(n, dtype=np.uint64), k)
        slot = np.tile(np.arange(k, dtype=np.uint64), n)
        h = _splitmix64(rows * np.uint64(0x100000001B3) + slot * np.uint64(0x1000193) + np.uint64(seed))
        cols = (h % np.uint64(n)).astype(np.int64)
        h2 = _splitmix64(h)
        vals = (h2 >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        hd = _splitmix64(np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(seed + 7))
        dvals = diag + (hd >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    rows = rows.astype(np.int64)
    keep = cols != rows
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    if symmetric_pattern:
        rows, cols, vals = np.concatenate([rows, cols]), np.concatenate([cols, rows]), np.concatenate([vals, vals])
    rows = np.concatenate([rows, np.arange(n, dtype=np.int64)])
    cols = np.concatenate([cols, np.arange(n, dtype=np.int64)])
    vals = np.concatenate([vals, dvals])
    key = rows * n + cols
    order = np.argsort(key, kind="stable")
    key, vals = key[order], vals[order]
    first = np.ones(key.shape[0], dtype=bool)
    first[1:] = key[1:] != key[:-1]
    key, vals = key[first], vals[first]
    r = key // n
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=n), out=indptr[1:])
    return np.ascontiguousarray(vals), (key % n).astype(np.int32), indptr.astype(np.int32)


def to_csc(data, indices, indptr):
    """CSR arrays -> CSC arrays of the same matrix (sorted row indices)."""
    n = indptr.shape[0] - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    order = np.argsort(indices.astype(np.int64) * n + rows, kind="stable")
    cptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(indices, minlength=n), out=cptr[1:])
    return np.ascontiguousarray(data[order]), rows[order].astype(np.int32), cptr.astype(np.int32)


def symmetrize(data, indices, indptr):
    """(A + A^T)/2 in CSR (tests.py:219-220)."""
    import scipy.sparse as sp
    n = indptr.shape[0] - 1
    A = sp.csr_matrix((data, indices, indptr), shape=(n, n))
    S = ((A + A.T) / 2).tocsr()
    S.sort_indices()
    return S.data.astype(np.float64), S.indices.astype(np.int32), S.indptr.astype(np.int32)


def box_stencil(dims, diag=None):
    """full (2 radius-1 box) stencil on a lexicographic grid: 9-point in 2-D, 27-point in 3-D; diagonally dominant, nonsymmetric values"""
    dims = tuple(int(d) for d in dims)
    n = int(np.prod(dims))
    nd = len(dims)
    strides = [1]
    for d in dims[:-1]:
        strides.append(strides[-1] * d)
    ii = np.arange(n, dtype=np.int64)
    coords, rem = [], ii
    for d in dims:
        coords.append(rem % d); rem = rem // d
    offs, valid = [], []
    for delta in np.ndindex(*([3] * nd)):
        dl = [v - 1 for v in delta][::-1]                      # slowest dimension first => ascending column order
        o = sum(dl[nd - 1 - a] * strides[a] for a in range(nd))
        ok = np.ones(n, dtype=bool)
        for a in range(nd):
            c = coords[a] + dl[nd - 1 - a]
            ok &= (c >= 0) & (c < dims[a])
        offs.append(o); valid.append(ok)
    order = np.argsort(offs)
    offs = [offs[k] for k in order]; valid = np.stack([valid[k] for k in order], axis=1)
    counts = valid.sum(axis=1)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    cols = (ii[:, None] + np.asarray(offs, dtype=np.int64)[None, :])[valid]
    w = float(3 ** nd) if diag is None else float(diag)
    base = np.where(np.asarray(offs) == 0, w, -1.0 + 0.1 * np.sin(np.arange(len(offs))))
    vals = np.broadcast_to(base[None, :], valid.shape)[valid]
    return np.ascontiguousarray(vals, dtype=np.float64), cols.astype(np.int32), indptr.astype(np.int32)


def mesh_with_holes(g, seed=5, frac=0.03, permute=False):
    """The 7-point matrix of a g^3 grid with a fraction of its points removed (their rows and columns deleted): x-lines of irregular
    length, what a mesh that is not a box looks like to the kernels.  permute: the same matrix under a random symmetric permutation
    (no ordering left for a schedule to find).  Returns (data, indices, indptr) with int32 indices, columns sorted."""
    import scipy.sparse as sp
    d, i, p = poisson3d(g)
    n = p.shape[0] - 1
    A = sp.csr_matrix((d, i, p), shape=(n, n))
    rng = np.random.default_rng(seed)
    idx = np.flatnonzero(rng.random(n) > frac)
    if permute:
        idx = idx[rng.permutation(idx.shape[0])]
    B = A[idx][:, idx].tocsr()
    B.sort_indices()
    return np.ascontiguousarray(B.data, dtype=np.float64), B.indices.astype(np.int32), B.indptr.astype(np.int32)
