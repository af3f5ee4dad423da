"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs oracle/_ref/libilupp_ref.so, i.e. /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Every output array below comes out of the reference's own C++ (through oracle/ref_shim.cpp, which
calls the same entry points as src/binding.cpp:366-447 and the same preconditioner classes for
apply).  Inputs built with scipy.sparse.random (the reference's own test matrices,
test/tests.py:9-36) are stored as arrays because scipy's stream is version dependent; inputs from
tests/matgen.py are regenerated at test time and guarded by a sha256 of their bytes.

Files written:
  reftests.npz     the reference's own test matrices (tests.py) and every in-scope function on them
  configs.npz      small config-shaped cases (2-D/3-D Poisson, diag-dominant random), full arrays
  edges.npz        edge cases (1x1, fill_in=1, explicit zeros, zero pivot row, ties in the top-k cut)
  digests.json     sha256 of outputs for medium sizes (C1 200x200, 64^3, random n=50 000)
"""
import hashlib
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import matgen  # noqa: E402
from oracle import oracle as O  # noqa: E402

ref = O.ref()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def put_mat(out, key, M):
    out[key + "_data"], out[key + "_indices"], out[key + "_indptr"] = M[0], M[1], M[2]
    out[key + "_is_csr"] = np.array(int(M[3]), dtype=np.int8)


def rhs(n):
    # non-trivial, deterministic, exactly representable pattern
    return 1.0 + (np.arange(n, dtype=np.float64) % 17) / 16.0 - (np.arange(n, dtype=np.float64) % 5) / 8.0


def run_all(out, key, M, S, full=True, ilut_params=((5, 0.1), (100, 0.0), (10, 1e-4)),
            icholt_params=((0, 0.0), (5, 1e-3), (2, 0.05))):
    """All in-scope reference functions on matrix M (general) and S (symmetrised); results into out."""
    n = M[2].shape[0] - 1
    b = rhs(n)
    if full:
        out[key + "/b"] = b
    L, U = ref.ilu0(M)
    rec = {}
    rec["ilu0"] = (L, U)
    put = (lambda k, M_: put_mat(out, key + "/" + k, M_)) if full else (lambda k, M_: None)
    put("ilu0_L", L); put("ilu0_U", U)
    out[key + "/ilu0_apply"] = ref.apply_lu(L, U, b, O.ID)
    out[key + "/ilu0_apply_trans"] = ref.apply_lu(L, U, b, O.TRANSPOSE)
    out[key + "/ilu0_total_nnz"] = np.array(int(L[2][-1]) + int(U[2][-1]))
    for (p, t) in ilut_params:
        tag = "ilut_%d_%g" % (p, t)
        L, U = ref.ilut(M, p, t)
        put(tag + "_L", L); put(tag + "_U", U)
        out[key + "/" + tag + "_apply"] = ref.apply_lu(L, U, b, O.ID)
        out[key + "/" + tag + "_apply_trans"] = ref.apply_lu(L, U, b, O.TRANSPOSE)
        out[key + "/" + tag + "_total_nnz"] = np.array(O.ref_total_nnz_ilut(M, p, t))
        rec[tag] = (L, U)
    if S is not None:
        L = ref.ichol0(S)
        put("ichol0_L", L)
        out[key + "/ichol0_apply"] = ref.apply_llt(L, b, O.ID)
        out[key + "/ichol0_apply_trans"] = ref.apply_llt(L, b, O.TRANSPOSE)
        rec["ichol0"] = (L,)
        for (a, t) in icholt_params:
            tag = "icholt_%d_%g" % (a, t)
            L = ref.icholt(S, a, t)
            put(tag + "_L", L)
            out[key + "/" + tag + "_apply"] = ref.apply_llt(L, b, O.ID)
            rec[tag] = (L,)
    return rec


# ---------------------------------------------------------------------------------------------
# (1) the reference's own test matrices (test/tests.py:9-36)
# ---------------------------------------------------------------------------------------------
def laplace_matrix(n):
    h = 1.0 / (n + 1)
    d = np.ones(n) / (h ** 2)
    return sp.diags((-d[:-1], 2 * d, -d[:-1]), (-1, 0, 1)).tocsr()


def laplace2d_matrix(n_total):
    n = int(np.sqrt(n_total))
    A, I = laplace_matrix(n), sp.eye(n)
    return (sp.kron(A, I) + sp.kron(I, A)).tocsr()


def random_matrix(n, eye_factor=10.0):
    density = min(1.0, 5 / n)
    return (sp.random(n, n, density=density, random_state=39273) + eye_factor * sp.eye(n)).tocsr()


def make_reftests():
    out = {}
    for name, A in (("laplace", laplace_matrix(50)), ("laplace2d", laplace2d_matrix(50)),
                    ("random", random_matrix(50))):
        for fmt in ("csr", "csc"):
            M = O.from_scipy(A.asformat(fmt))
            S = O.from_scipy(((A + A.T) / 2).asformat(fmt))
            key = "%s_%s" % (name, fmt)
            put_mat(out, key + "/A", M)
            put_mat(out, key + "/S", S)
            run_all(out, key, M, S, icholt_params=((0, 0.0), (5, 1e-3)))
    np.savez_compressed(os.path.join(HERE, "reftests.npz"), **out)
    print("reftests.npz:", len(out), "arrays")


# ---------------------------------------------------------------------------------------------
# (2) small config-shaped cases, inputs from tests/matgen.py
# ---------------------------------------------------------------------------------------------
CONFIG_CASES = {
    "poisson2d_20": lambda: matgen.poisson2d(20),
    "poisson3d_8": lambda: matgen.poisson3d(8),
    "poisson3d_16": lambda: matgen.poisson3d(16),
    "poisson3d_5x7x3": lambda: matgen.poisson3d(5, 7, 3),
    "random_dd_2000": lambda: matgen.random_dd(2000, 19, 25.0, 12345),
    "random_dd_300_k6": lambda: matgen.random_dd(300, 6, 4.0, 99),
}


def make_configs():
    out = {}
    for name, gen in CONFIG_CASES.items():
        d, i, p = gen()
        out[name + "/input_sha"] = np.frombuffer(bytes.fromhex(sha(d) + sha(i) + sha(p)), dtype=np.uint8)
        sd, si, spp = matgen.symmetrize(d, i, p)
        for fmt in ("csr", "csc"):
            if fmt == "csr":
                M, S = (d, i, p, True), (sd, si, spp, True)
            else:
                M, S = matgen.to_csc(d, i, p) + (False,), matgen.to_csc(sd, si, spp) + (False,)
            run_all(out, "%s_%s" % (name, fmt), M, S, ilut_params=((5, 0.1), (10, 1e-4), (1, 0.0), (3, 0.0)))
    np.savez_compressed(os.path.join(HERE, "configs.npz"), **out)
    print("configs.npz:", len(out), "arrays")


# ---------------------------------------------------------------------------------------------
# (3) edge cases
# ---------------------------------------------------------------------------------------------
def make_edges():
    out = {}
    # 1x1
    M = (np.array([4.0]), np.array([0], dtype=np.int32), np.array([0, 1], dtype=np.int32), True)
    put_mat(out, "one/A", M)
    run_all(out, "one", M, M, ilut_params=((5, 0.1),), icholt_params=((0, 0.0),))
    # explicit stored zeros (ILUT strips them through compress(0.0); ILU0 keeps them)
    A = sp.csr_matrix(np.array([[4.0, 1.0, 0.0, 2.0],
                                [1.0, 5.0, 1.0, 0.0],
                                [0.5, 1.0, 6.0, 1.0],
                                [2.0, 0.0, 1.0, 7.0]]))
    d, i, p = A.data.copy(), A.indices.astype(np.int32), A.indptr.astype(np.int32)
    # insert explicit zeros at (0,2) and (3,1)
    dense_mask = np.ones((4, 4), dtype=bool)
    rows, cols = np.nonzero(dense_mask)
    vals = np.asarray(A.todense())[rows, cols]
    M = (vals.astype(np.float64), cols.astype(np.int32), np.arange(0, 17, 4, dtype=np.int32), True)
    put_mat(out, "zeros/A", M)
    run_all(out, "zeros", M, None, ilut_params=((5, 0.1), (100, 0.0)))
    # zero pivot in ILUT: row 2 becomes singular
    Z = np.array([[1.0, 2.0, 0.0], [2.0, 4.0, 1.0], [0.0, 1.0, 3.0]])
    zr, zc = np.nonzero(Z)
    Zs = sp.csr_matrix(Z)
    Zm = O.from_scipy(Zs)
    put_mat(out, "zeropivot/A", Zm)
    try:
        ref.ilut(Zm, 100, 0.0)
        out["zeropivot/err_row"] = np.array(-1)
    except O.OracleError as e:
        out["zeropivot/err_row"] = np.array(e.row)
    # structured-grid fill case with top-k magnitude ties (126 of 4.2M cuts at 128^3)
    d, i, p = matgen.poisson3d(12)
    M = (d, i, p, True)
    for (pp, t) in ((10, 1e-4), (4, 0.0), (20, 1e-6)):
        L, U = ref.ilut(M, pp, t)
        put_mat(out, "ties/ilut_%d_%g_L" % (pp, t), L)
        put_mat(out, "ties/ilut_%d_%g_U" % (pp, t), U)
    sd, si, spp = matgen.symmetrize(d, i, p)
    for (a, t) in ((5, 1e-3), (3, 0.0), (12, 0.0)):
        L = ref.icholt((sd, si, spp, True), a, t)
        put_mat(out, "ties/icholt_%d_%g_L" % (a, t), L)
    np.savez_compressed(os.path.join(HERE, "edges.npz"), **out)
    print("edges.npz:", len(out), "arrays")


# ---------------------------------------------------------------------------------------------
# (4) medium sizes as digests
# ---------------------------------------------------------------------------------------------
def digest_mat(M):
    return {"data": sha(M[0]), "indices": sha(M[1]), "indptr": sha(M[2]), "is_csr": bool(M[3]),
            "nnz": int(M[2][-1])}


def make_digests():
    dg = {}
    # C1: 2-D 5-point 200x200, ILU0 (BASELINE config 1), full digest + sampled rows live in configs via 20x20
    for name, gen in (("poisson2d_200", lambda: matgen.poisson2d(200)),
                      ("poisson3d_64", lambda: matgen.poisson3d(64))):
        d, i, p = gen()
        M = (d, i, p, True)
        n = p.shape[0] - 1
        L, U = ref.ilu0(M)
        x = ref.apply_lu(L, U, np.ones(n), O.ID)
        xt = ref.apply_lu(L, U, np.ones(n), O.TRANSPOSE)
        entry = {"input": digest_mat(M), "ilu0_L": digest_mat(L), "ilu0_U": digest_mat(U),
                 "ilu0_apply_ones": sha(x), "ilu0_apply_trans_ones": sha(xt),
                 "ilu0_apply_ones_sum": float(x.sum())}
        Mc = matgen.to_csc(d, i, p) + (False,)
        L, U = ref.ilu0(Mc)
        entry["csc_ilu0_L"] = digest_mat(L); entry["csc_ilu0_U"] = digest_mat(U)
        entry["csc_ilu0_apply_ones"] = sha(ref.apply_lu(L, U, np.ones(n), O.ID))
        sd, si, spp = d, i, p   # already symmetric
        Lc = ref.ichol0(M)
        entry["ichol0_L"] = digest_mat(Lc)
        entry["ichol0_apply_ones"] = sha(ref.apply_llt(Lc, np.ones(n), O.ID))
        for (a, t) in ((0, 0.0), (5, 1e-3)):
            Lt = ref.icholt(M, a, t)
            entry["icholt_%d_%g_L" % (a, t)] = digest_mat(Lt)
            entry["icholt_%d_%g_apply_ones" % (a, t)] = sha(ref.apply_llt(Lt, np.ones(n), O.ID))
        if name == "poisson2d_200":
            for (pp, t) in ((10, 1e-4), (5, 0.1)):
                L, U = ref.ilut(M, pp, t)
                entry["ilut_%d_%g_L" % (pp, t)] = digest_mat(L)
                entry["ilut_%d_%g_U" % (pp, t)] = digest_mat(U)
        dg[name] = entry
        print(name, "done")
    d, i, p = matgen.random_dd(50000, 19, 25.0, 12345)
    M = (d, i, p, True)
    entry = {"input": digest_mat(M)}
    for (pp, t) in ((10, 1e-4), (5, 0.1)):
        L, U = ref.ilut(M, pp, t)
        entry["ilut_%d_%g_L" % (pp, t)] = digest_mat(L)
        entry["ilut_%d_%g_U" % (pp, t)] = digest_mat(U)
        entry["ilut_%d_%g_apply_ones" % (pp, t)] = sha(ref.apply_lu(L, U, np.ones(50000), O.ID))
    L, U = ref.ilu0(M)
    entry["ilu0_L"] = digest_mat(L); entry["ilu0_U"] = digest_mat(U)
    entry["ilu0_apply_ones"] = sha(ref.apply_lu(L, U, np.ones(50000), O.ID))
    dg["random_dd_50000"] = entry
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(dg, f, indent=1, sort_keys=True)
    print("digests.json written")


if __name__ == "__main__":
    make_reftests()
    make_configs()
    make_edges()
    make_digests()
