"""Randomized small/medium matrices through the ICholT dataflow kernel and the ILUT wave kernel, bit-exact against the
C restatement of the reference (tests/test_gpu_parity.py::test_fuzz_new_kernels, profiles/tools/fuzz_kernels.py)."""
import numpy as np
import scipy.sparse as sp


def eq(M, Mo):
    return (np.array_equal(M.indptr, Mo[2]) and np.array_equal(M.indices, Mo[1])
            and np.array_equal(M.data.view(np.int64), Mo[0].view(np.int64)))

def run(nseeds, first_seed=0, verbose=True):
    import ilupp_amd as ilupp
    from oracle import oracle as O
    orc = O.orc()
    bad = 0
    for seed in range(first_seed, first_seed + nseeds):
        rng = np.random.default_rng(seed)
        if verbose == 2:
            print('seed', seed, flush=True)
        n = int(rng.choice([1, 2, 3, 7, 20, 64, 65, 150, 400, 1500, 2500]))
        dens = float(rng.choice([0.5, 2.0, 5.0, 12.0])) / max(n, 1)
        R = sp.random(n, n, density=min(1.0, dens), random_state=rng, format='csr')
        if rng.random() < 0.3:
            R.data = np.round(R.data * 4) / 4          # many equal magnitudes: ties at the top-k cut
        A = (R + sp.identity(n) * float(rng.choice([1.0, 4.0, 25.0]))).tocsr(); A.sort_indices()
        S = ((A + A.T) * 0.5 + sp.identity(n) * float(rng.choice([2.0, 2.0, 0.0, -3.0]))).tocsr(); S.sort_indices()      # sometimes indefinite: NaN columns
        for fmt in ('csr', 'csc'):
            Af = A if fmt == 'csr' else A.tocsc(); Sf = S if fmt == 'csr' else S.tocsc()
            Af.sort_indices(); Sf.sort_indices()
            Mi = (Af.data.astype(np.float64), Af.indices.astype(np.int32), Af.indptr.astype(np.int32), fmt == 'csr')
            Ms = (Sf.data.astype(np.float64), Sf.indices.astype(np.int32), Sf.indptr.astype(np.int32), fmt == 'csr')
            fill = int(rng.choice([1, 2, 3, 5, 10, 17, 70, 100])); tau = float(rng.choice([0.0, 1e-6, 1e-3, 0.05, 0.3]))
            try:
                Lo, Uo = orc.ilut(Mi, fill, tau)
                P = ilupp.ILUTPreconditioner(Af.copy(), fill_in=fill, threshold=tau)
                L, U = P.factors()
                ok = eq(L, Lo) and eq(U, Uo)
                if ok:
                    b = np.cos(np.arange(n, dtype=np.float64)) + 1.5
                    x = b.copy(); P.apply(x)
                    xt = b.copy(); P.apply_trans(xt)
                    ok = (np.array_equal(x, orc.apply_lu(Lo, Uo, b, O.ID), equal_nan=True)
                          and np.array_equal(xt, orc.apply_lu(Lo, Uo, b, O.TRANSPOSE), equal_nan=True))
            except O.OracleError:
                try:
                    ilupp.ILUTPreconditioner(Af.copy(), fill_in=fill, threshold=tau); ok = False
                except RuntimeError:
                    ok = True
            if not ok:
                bad += 1; print('ILUT MISMATCH seed', seed, fmt, n, fill, tau, flush=True)
            add = int(rng.choice([0, 1, 2, 5, 9, 40])); tau = float(rng.choice([0.0, 1e-6, 1e-3, 0.05, 0.3]))
            Lo = orc.icholt(Ms, add, tau)
            # a column of the reference's factor that does not start with its diagonal (dropped, or the whole column empty: the
            # pivot's square root was NaN) = indefinite input or a budget below one entry: the engine reports it instead of
            # returning the reference's NaN-filled factor (deviations)
            lens = np.diff(Lo[2])
            lost = bool(np.any(lens == 0)) or bool(np.any(Lo[1][Lo[2][:-1][lens > 0]] != np.arange(n)[lens > 0]))
            try:
                P = ilupp.ICholTPreconditioner(Sf.copy(), add_fill_in=add, threshold=tau)
            except RuntimeError as e:
                if not (lost and ('pivot of a column is NaN' in str(e) or 'diagonal entry of a column was dropped' in str(e))):
                    bad += 1; print('ICHOLT UNEXPECTED ERROR seed', seed, fmt, n, add, tau, e, flush=True)
                continue
            if lost:
                bad += 1; print('ICHOLT MISSING ERROR seed', seed, fmt, n, add, tau, flush=True)
                continue
            L, = P.factors()
            b = np.cos(np.arange(n, dtype=np.float64)) + 1.5
            x = b.copy(); P.apply(x)
            if not (np.array_equal(L.indptr, Lo[2]) and np.array_equal(L.indices, Lo[1]) and np.array_equal(L.data, Lo[0], equal_nan=True)
                    and np.array_equal(x, orc.apply_llt(Lo, b, O.ID), equal_nan=True)):
                bad += 1; print('ICHOLT MISMATCH seed', seed, fmt, n, add, tau, flush=True)
    if verbose:
        print('fuzz: %d seeds, %d mismatches' % (nseeds, bad))
    return bad
