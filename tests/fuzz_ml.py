"""Randomised cases for the multilevel ILU++ preconditioner (no pivoting): small matrices of all kinds -- dense-ish, nearly empty rows,
missing diagonals, huge and tiny values, symmetric patterns, CSR / CSC -- with random parameter sets; used by tests/test_gpu_ml.py
(GPU against the oracle) and tests/test_oracle_ml.py (oracle against the real reference where it is built)."""
import numpy as np
import scipy.sparse as sp

STEPS = ["NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING", "MAX_WEIGHTED_MATCHING_ORDERING", "UNIT_OR_ZERO_DIAGONAL_SCALING", "SPARSE_FIRST_ORDERING", "SYMM_PQ"]


def case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 3, 5, 8, 17, 40, 90, 200, 350]))
    dens = float(rng.choice([0.02, 0.05, 0.15, 0.4])) if n > 8 else 0.6
    A = sp.random(n, n, density=min(1.0, dens), random_state=rng, format="lil")
    kind = int(rng.integers(0, 6))
    if kind == 0:
        A = A + sp.eye(n) * float(rng.choice([0.05, 0.5, 3.0]))
    elif kind == 1:                                             # symmetric pattern, no guaranteed diagonal
        A = A + A.T
    elif kind == 2:                                             # wild magnitudes
        A = A.tocsr(); A.data = A.data * np.exp(rng.normal(0, 6, A.data.shape[0])); A = A + sp.eye(n) * 1e-3
    elif kind == 3:                                             # some rows with the diagonal only, some missing diagonals
        A = A + sp.diags([np.where(rng.random(n) < 0.7, 1.0 + rng.random(n), 0.0)], [0])
    elif kind == 4:                                             # negative entries
        A = A.tocsr(); A.data = A.data - 0.5; A = A + sp.eye(n) * float(rng.choice([0.2, 2.0]))
    else:
        A = A + sp.diags([rng.random(n - 1) * 2], [1], shape=(n, n)) if n > 1 else A + sp.eye(n)
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    if A.nnz == 0:
        A = sp.eye(n, format="csr")
    A.sort_indices()
    if rng.random() < 0.4:
        A = A.tocsc()
    k = int(rng.integers(0, 4))
    pre = tuple(rng.choice(STEPS, size=k, replace=False)) if k else ()
    if rng.random() < 0.4:
        pre = ("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING")
    thr = float(rng.choice([0.0, 1e-3, 0.02, 0.1, 0.5]))
    knobs = {}
    if rng.random() < 0.3:
        knobs["MAX_LEVELS"] = int(rng.choice([1, 2, 4]))
    if rng.random() < 0.3:
        knobs["MIN_PIVOT"] = float(rng.choice([1e-6, 0.05, 0.3]))
    if rng.random() < 0.2:
        knobs["THRESHOLD_SHIFT_SCHUR"] = float(rng.choice([1e-3, 0.1]))
    if rng.random() < 0.2:
        knobs["SMALL_PIVOT_TERMINATES"] = False
    if rng.random() < 0.2:
        knobs["MIN_ELIM_FACTOR"] = float(rng.choice([0.1, 0.5]))
    if rng.random() < 0.3:
        knobs["fill_in"] = int(rng.choice([1, 2, 3, 6, 15]))
    if rng.random() < 0.35:                                     # other dropping rules, combined at random
        for name in ("USE_STANDARD_DROPPING", "USE_STANDARD_DROPPING2", "USE_ERR_PROP_DROPPING", "USE_ERR_PROP_DROPPING2", "USE_PIVOT_DROPPING"):
            knobs[name] = bool(rng.random() < 0.4)
        knobs["COMBINE_FACTOR"] = int(rng.integers(0, 4))
        if rng.random() < 0.5:
            knobs["NEUTRAL_ELEMENT"] = float(rng.choice([0.0, 0.5, 1.0]))
            knobs["WEIGHT_STANDARD_DROP"] = float(rng.choice([0.2, 1.0, 3.0]))
            knobs["WEIGHT_PIVOT_DROP"] = float(rng.choice([0.1, 1.0]))
        if rng.random() < 0.3:
            knobs["SCALE_WEIGHT_INVDIAG"] = True
    if rng.random() < 0.2:                                      # the rules that are recurrences over all steps: the factorisation runs as a chain
        if rng.random() < 0.5:
            knobs["USE_INVERSE_DROPPING"] = True
            knobs["WEIGHT_INVERSE_DROP"] = float(rng.choice([1.0, 0.3]))
        else:
            knobs["USE_WEIGHTED_DROPPING"] = True
            knobs["INIT_WEIGHTS_LU"] = float(rng.choice([1.0, 0.5]))
        if rng.random() < 0.5:
            knobs["USE_ERR_PROP_DROPPING"] = False
    return A, (thr, pre, knobs)
