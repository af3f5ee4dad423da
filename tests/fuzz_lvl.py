"""Randomized matrices through the level-ordered ILU(0) kernel (ilu0_lvl.hip: one wave per row, up to 64 entries per row, several
batches of 16 pivots), the level-ordered sweeps, IChol0 and ILUC, bit-exact against the C restatement of the reference
(tests/test_gpu_level_sweeps.py::test_fuzz_level_order, profiles/tools/fuzz_lvl.py)."""
import numpy as np
import scipy.sparse as sp


def _eq(M, Mo):
    return (np.array_equal(M.indptr, Mo[2]) and np.array_equal(M.indices, Mo[1])
            and np.array_equal(M.data.view(np.int64), Mo[0].view(np.int64)))


def run(nseeds, first_seed=0, verbose=True):
    import ilupp_amd as ilupp
    from oracle import oracle as O
    orc = O.orc()
    bad = 0
    for seed in range(first_seed, first_seed + nseeds):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1024, 1100, 2500, 6000]))
        per_row = float(rng.choice([5.0, 9.0, 14.0, 22.0, 40.0]))
        R = sp.random(n, n, density=min(1.0, per_row / n), random_state=rng, format='csr')
        if rng.random() < 0.3:
            # a band part: chains between consecutive rows, long dependency paths
            R = R + sp.diags([rng.random(n - 1) + 0.1, rng.random(n - 1) + 0.1], [-1, 1], format='csr')
        A = (R + sp.identity(n) * float(rng.choice([4.0, 25.0, 60.0]))).tocsr(); A.sort_indices()
        S = ((A + A.T) * 0.5 + sp.identity(n) * 30.0).tocsr(); S.sort_indices()
        b = np.cos(np.arange(n, dtype=np.float64)) + 1.5
        for fmt in ('csr', 'csc'):
            Af = (A if fmt == 'csr' else A.tocsc()); Af.sort_indices()
            Sf = (S if fmt == 'csr' else S.tocsc()); Sf.sort_indices()
            Mi = (Af.data.astype(np.float64), Af.indices.astype(np.int32), Af.indptr.astype(np.int32), fmt == 'csr')
            Ms = (Sf.data.astype(np.float64), Sf.indices.astype(np.int32), Sf.indptr.astype(np.int32), fmt == 'csr')
            # ILU(0)
            Lo, Uo = orc.ilu0(Mi)
            P = ilupp.ILU0Preconditioner(Af.copy())
            L, U = P.factors()
            ok = _eq(L, Lo) and _eq(U, Uo)
            for rep in range(2):
                x = b.copy(); P.apply(x); xt = b.copy(); P.apply_trans(xt)
                ok = ok and np.array_equal(x, orc.apply_lu(Lo, Uo, b, O.ID), equal_nan=True) \
                        and np.array_equal(xt, orc.apply_lu(Lo, Uo, b, O.TRANSPOSE), equal_nan=True)
            if not ok:
                bad += 1; print('ILU0 MISMATCH seed', seed, fmt, n, per_row, P.pr.path(), flush=True)
            # IChol(0)
            Lo = orc.ichol0(Ms)
            P = ilupp.IChol0Preconditioner(Sf.copy())
            (L,) = P.factors()
            x = b.copy(); P.apply(x)
            if not (_eq(L, Lo) and np.array_equal(x, orc.apply_llt(Lo, b, O.ID), equal_nan=True)):
                bad += 1; print('ICHOL0 MISMATCH seed', seed, fmt, n, per_row, flush=True)
            # ILUC
            fill = int(rng.choice([3, 8, 20])); tau = float(rng.choice([0.0, 1e-4, 1e-2]))
            try:
                Lo, Uo = orc.iluc(Mi, fill, tau)
            except O.OracleError:
                continue
            P = ilupp.ILUCPreconditioner(Af.copy(), fill_in=fill, threshold=tau)
            L, U = P.factors()
            ok = _eq(L, Lo) and _eq(U, Uo)
            x = b.copy(); P.apply(x); xt = b.copy(); P.apply_trans(xt)
            ok = ok and np.array_equal(x, orc.apply_lu(Lo, Uo, b, O.ID), equal_nan=True) \
                    and np.array_equal(xt, orc.apply_lu(Lo, Uo, b, O.TRANSPOSE), equal_nan=True)
            if not ok:
                bad += 1; print('ILUC MISMATCH seed', seed, fmt, n, per_row, fill, tau, flush=True)
    if verbose:
        print('fuzz (level order): %d seeds, %d mismatches' % (nseeds, bad))
    return bad
