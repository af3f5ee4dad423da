#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Metric: ILU(0) factor + one L/U apply on the 3-D 7-point Poisson 256^3 CSR matrix (fp64 values,
int32 indices), reported as nnz(A)/s, with the achieved fraction of the HBM roofline for the
dominant kernel and the reference's own CPU path timed beside it.

A "step" = one complete ILU(0) factorisation of the device-resident CSR matrix (pattern analysis
+ row scheduling + numeric factorisation) followed by one apply() on a device-resident vector,
i.e. exactly what `P = ilupp.ILU0Preconditioner(A); P.apply(x)` does, with A and x already in HBM
when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid G] [--no-cpu] [--no-extra] [--config C2COLD|C2H|C2HP|C3|C4|ILUC|S27|S9]

N > 1: the path does not shard (a single factorisation is one dependency chain), so every rank
factors its own matrix of a batch (weak scaling, no data-path collective; RCCL only for the barrier,
the max-over-ranks time and one gather of per-matrix records).  Launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU);
run without a launcher, `--gpus N` starts that launcher itself -- before anything touches the GPU.

The default run (one GPU) also measures what a matrix that is NOT the benchmark's box gets (C2COLD: the headline step in a fresh process
through the plain entry; C2H: the 256^3 mesh with 3 % of its points removed; C2HP: the same under a random symmetric permutation), C3, C4,
one ILUC config and the multilevel configs (C5: one level, C5M: with the matching, C5L: a 5-level object, C5P: config 5 as named, C5PB: 64
of those side by side) after the headline one (about a minute; --no-extra skips them), the same step through the entry without the
entry count ("plain_entry") and a re-factorisation with new values on the analysed pattern ("refactor").  Every extra carries path, roofline,
the reference's CPU figure and construct_plus_first_apply_s.
--config C3 / C4 add the other BASELINE configs as extra keys of the same JSON line ("extra"):
ILUT(10, 1e-4) on the random diagonally dominant matrix with n = 1e6, ICholT(0, 0) on the 256^3 matrix
(bytes = read A + write the factors actually produced).  --config S27 / S9: ILU(0) beyond 7-point rows (27-point box
stencil 128^3, 9-point 2048^2: the level-ordered kernels), same keys.  first_apply_ms: the apply right after construction (it may
build what the sweeps need); apply_ms: the apply after that.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)

# kernel that dominates the numeric factorisation, per construction path (ilupp_hip_path)
FACTOR_KERNEL = {"ilu0:static-direct": "k_ilu0_sd", "ilu0:static-level-major": "k_ilu0_st", "ilu0:level-order": "k_ilu0_lvl",
                 "ilu0:csr-program": "k_ilu0_numeric_lc", "ilu0:csr": "k_ilu0_numeric"}


def algorithmic_bytes(n, nnz):
    """factor = read A once + write L,U once; apply = read L,U once + read/write x per solve."""
    factor = (12 * nnz + 4 * (n + 1)) + (12 * (nnz + n) + 8 * (n + 1))
    apply_ = 12 * (nnz + n) + 8 * (n + 1) + 32 * n
    return factor, apply_


def _same_kernel(profiled, wanted):
    """a profiler's name of a kernel ('ilupp::k_sptrsv_wv<1, false, false>': namespace, defaulted template arguments spelled out)
    against the library's ('k_sptrsv_wv<1, false>')"""
    a = profiled.replace("ilupp::", "").replace("void ", "").split("(")[0].strip()
    if a == wanted:
        return True
    if "<" in a and "<" not in wanted:
        return a.split("<", 1)[0] == wanted          # (a kernel named without its template arguments: any instance)
    if "<" not in a or "<" not in wanted:
        return False
    an, aa = a.split("<", 1)
    wn, wa = wanted.split("<", 1)
    aa = [x.strip() for x in aa.rstrip(">").split(",")]
    wa = [x.strip() for x in wa.rstrip(">").split(",")]
    return an == wn and aa[:len(wa)] == wa and all(x in ("false", "0") for x in aa[len(wa):])


def measured_traffic(kernel, g):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r*_pmc_hbm.json, collected with
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs of this same command).  gfx950
    correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports one half of the loaded bytes, WRITE_SIZE
    is taken as reported; both counters are in KiB.  Only valid for the 256^3 workload it was measured on."""
    import glob
    if g != 256:
        return None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")), reverse=True):
        try:
            ks = json.load(open(f))["kernels"]
            for name, k in ks.items():
                if _same_kernel(name, kernel):
                    TRAFFIC_SOURCE.add(os.path.relpath(f, ROOT))
                    return (2.0 * k["FETCH_SIZE_KiB_avg_per_launch"] + k["WRITE_SIZE_KiB_avg_per_launch"]) * 1024.0
        except Exception:
            continue
    return None


TRAFFIC_SOURCE = set()      # the committed counter files the traffic figures of this line come from (they are NOT measured in this run)


def cpu_extra(name, d, i, p):
    """The reference's own C++ (oracle/_ref, one host core) on an extra config's workload, construction only -- beside the GPU's construct_s.
    C3 (102 s on one core) is quoted from BASELINE.md, not run."""
    n, nnz = p.shape[0] - 1, int(p[-1])
    if name == "C3":
        return {"value": nnz / 102.1, "unit": "nnz/s", "cores": 1, "kind": "quoted", "seconds": 102.1,
                "sample": "BASELINE.md section 3, row C3: ILUT_heap on one core of the survey container (not run here: 102 s)"}
    try:
        import ctypes
        from oracle import oracle as O
        if not O.ref_available():
            return None
        ref = O.ref()
        A = (d, i, p, True)
        t0 = time.perf_counter()
        if name in ("C2H", "C2HP", "S27", "S9"):
            L, U = ref.ilu0(A)
            x = np.ones(n)
            x = ref.trisolve(L, O.LOWER, O.ID, x)
            x = ref.trisolve(U, O.UPPER, O.ID, x)
        elif name == "C4":
            ref.icholt(A, 0, 0.0)
        elif name == "ILUC":
            ref.iluc(A, 8, 1e-2)
        elif name in ("C5", "C5P", "C5L"):
            I32P, F64P = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)
            f = ref.lib.ref_ilupp_apply
            f.argtypes = [ctypes.c_int32, I32P, I32P, F64P, ctypes.c_int, ctypes.c_int32, ctypes.c_double, ctypes.c_int32, ctypes.c_int, F64P, I32P, I32P]
            f.restype = ctypes.c_int
            x = np.ones(n)
            lev, tn = ctypes.c_int32(0), ctypes.c_int32(0)
            dd, ii, pp = (np.ascontiguousarray(d, dtype=np.float64), np.ascontiguousarray(i, dtype=np.int32), np.ascontiguousarray(p, dtype=np.int32))
            rc = f(ctypes.c_int32(n), pp.ctypes.data_as(I32P), ii.ctypes.data_as(I32P), dd.ctypes.data_as(F64P), ctypes.c_int(1),
                   ctypes.c_int32(10 if name == "C5P" else 1), ctypes.c_double(0.3 if name == "C5L" else 1e-3), ctypes.c_int32(-1), ctypes.c_int(0),
                   x.ctypes.data_as(F64P), ctypes.byref(lev), ctypes.byref(tn))
            if rc:
                return None
        else:
            return None
        sec = time.perf_counter() - t0
        return {"value": nnz / sec, "unit": "nnz/s", "cores": 1, "kind": "reference", "seconds": sec,
                "sample": "the same matrix once through oracle/_ref (the reference's C++ compiled from /root/reference), construction" +
                          (" + one apply" if name in ("C5", "C5P", "C5L", "C2H", "C2HP", "S27", "S9") else "") + ", one host core"}
    except Exception as e:                     # the CPU leg never takes the line down
        return {"value": None, "kind": "failed", "sample": repr(e)}


EXTRA_KERNEL = {"C2H": None, "C2HP": None, "C3": "k_ilut_rows_wp", "C4": "k_icholt_df", "ILUC": "k_iluc_df", "C5": "k_piluc_df", "C5M": "k_piluc_df", "C5L": "k_piluc_df",
                "C5P": "k_pilucdp_lds", "S27": "k_ilu0_lvl", "S9": "k_ilu0_lvl"}


def cpu_baseline(g, want_ref=True):
    """The reference's own C++ path (oracle/_ref, kind "reference") or the plain-C restatement (kind "port") on
    ONE host core (the reference is single-threaded): ILU(0) factor + one apply (the two triangular solves, no
    copies) on the same workload.  Bounded: one pass over the workload.  Returns the record and apply(ones)."""
    import matgen
    from oracle import oracle as O
    kind = "reference" if (want_ref and O.ref_available()) else "port"
    lib = O.ref() if kind == "reference" else O.orc()
    d, i, p = matgen.poisson3d(g)
    n, nnz = p.shape[0] - 1, int(p[-1])
    t0 = time.perf_counter()
    L, U = lib.ilu0((d, i, p, True))
    t1 = time.perf_counter()
    # preconditioner_implementation.h:321-334: L.triangular_solve(LOWER) then U.triangular_solve(UPPER), in place
    x = np.ones(n)
    x = lib.trisolve(L, O.LOWER, O.ID, x)
    x = lib.trisolve(U, O.UPPER, O.ID, x)
    t2 = time.perf_counter()
    x2 = lib.apply_lu(L, U, np.ones(n), O.ID)          # through the preconditioner object: copies L and U first
    t3 = time.perf_counter()
    try:
        cpu = open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip()
    except Exception:
        cpu = "unknown"
    rec = {"value": nnz / (t2 - t0), "unit": "nnz/s", "cores": 1, "kind": kind,
           "sample": "the full workload once: 3-D 7-pt Poisson %d^3 (n=%d, nnz=%d); factor %.3f s + apply %.3f s (two "
                     "triangular solves in place); host CPU: %s, %d cores present" % (g, n, nnz, t1 - t0, t2 - t1, cpu, os.cpu_count() or 0),
           "factor_s": t1 - t0, "apply_s": t2 - t1, "apply_with_copies_s": t3 - t2,
           "checksum": float(np.sum(x)), "apply_paths_agree": bool(np.array_equal(x, x2))}
    return rec, x


def spawn_ranks(args):
    """--gpus N without a launcher: start `torch.distributed.run` as a child (nothing here has touched the GPU)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def batch_matrix(g, member):
    """matrix `member` of the batch: the 7-point matrix with its diagonal shifted by 0.01 * member"""
    import matgen
    d, i, p = matgen.poisson3d(g)
    if member:
        d = d + np.where(d > 0, 0.01 * member, 0.0)
    return d, i, p


def ml_batch_member(dev, member, n=1000000, preset=1):
    """matrix `member` of the C5 batch (unsymmetric random CSR, seed 12345 + member) through the multilevel preconditioner
    (default_configuration(preset), threshold 1e-3) + one apply: (member, levels, total_nnz, sha256 of apply(ones), ms).
    preset 10 is BASELINE config 5 as named (maximum weighted matching + the factorisation with pivoting); preset 1 the family without
    pivoting."""
    import torch
    import matgen
    import ilupp_amd as ilupp
    from ilupp_amd import _native
    prm = ilupp.iluplusplus_precond_parameter()
    prm.default_configuration(preset)
    prm.threshold = 1e-3
    dm, im, pm = matgen.random_dd(n, 8, 25.0, 12345 + member)
    a = [torch.from_numpy(v).to(dev) for v in (dm, im, pm)]
    xb = torch.ones(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Pm = _native.MultilevelILUCDPPreconditioner_device(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), n, True, prm)
    Pm.apply_device(xb.data_ptr(), n, transpose=False, sync=True)
    ms = 1e3 * (time.perf_counter() - t0)
    return (member, int(Pm.levels()), int(Pm.total_nnz), hashlib.sha256(xb.cpu().numpy().tobytes()).hexdigest(), ms)


def _cpu_batch_one(a):
    member, n, preset = a
    import matgen
    dm, im, pm = matgen.random_dd(n, 8, 25.0, 12345 + member)
    t0 = time.perf_counter()
    r = cpu_extra("C5P" if preset == 10 else "C5", dm, im, pm)
    return (time.perf_counter() - t0) if (r and r.get("value")) else None


def cpu_batch(world, n, preset):
    """BASELINE.md section 4.3's CPU figure for the batch: `world` single-thread instances of the reference (oracle/_ref), one matrix each, on
    `world` host cores at once"""
    try:
        import multiprocessing as mp
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(world) as pool:
            secs = pool.map(_cpu_batch_one, [(m, n, preset) for m in range(world)])
        wall = time.perf_counter() - t0
        if any(v is None for v in secs):
            return None
        return {"kind": "reference", "cores": world, "wall_s": wall, "per_matrix_s": secs, "matrices_per_s": world / wall,
                "sample": "%d processes, one matrix (n=%d, default_configuration(%d)) each: construction + one apply through oracle/_ref" % (world, n, preset)}
    except Exception as e:
        return {"kind": "failed", "sample": repr(e)}


def batch_config(steps, members=64, n=100000):
    """C5PB: BASELINE config 5's many-matrices shape on ONE GPU: `members` matrices (n = 1e5 each), default_configuration(10) -- the
    factorisation with pivoting, a sequential chain per matrix -- built side by side (ilupp_hip_ml_create_batch: the chains of all
    members in one launch, one workgroup each), against the same construction for one matrix alone"""
    import matgen
    import ilupp_amd as ilupp
    from ilupp_amd import _native
    prm = ilupp.iluplusplus_precond_parameter()
    prm.default_configuration(10)
    prm.threshold = 1e-3
    mats = [matgen.random_dd(n, 8, 25.0, 12345 + m) for m in range(members)]
    one, bat = [], []
    nnz_f = 0
    for rep in range(max(2, min(steps, 3))):
        t0 = time.perf_counter()
        P = _native.MultilevelILUCDPPreconditioner(*mats[0], True, prm)
        t1 = time.perf_counter()
        Ps = _native.MultilevelILUCDPPreconditioner_batch(mats, True, prm)
        t2 = time.perf_counter()
        assert Ps[0].total_nnz == P.total_nnz
        nnz_f = sum(int(q.total_nnz) for q in Ps)
        if rep:
            one.append(t1 - t0); bat.append(t2 - t1)
        P = Ps = None
    s1, sb = float(np.median(one)), float(np.median(bat))
    return {"workload": "C5PB: %d x ILUppPreconditioner(default_configuration(10), threshold=1e-3), random unsymmetric CSR n=%d each, host arrays in, "
                        "built side by side (one launch for all chains)" % (members, n),
            "members": members, "n": n, "construct_one_s": s1, "construct_batch_s": sb, "batch_over_one": sb / s1, "matrices_per_s": members / sb,
            "factor_nnz_total": nnz_f}


_COLD = r"""
import sys, time, json
sys.path[:0] = [%r, %r]
import numpy as np, torch
import matgen
from ilupp_amd import _native
g = int(sys.argv[1])
d, i, p = matgen.poisson3d(g)
n, nnz = p.shape[0] - 1, int(p[-1])
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
x = torch.ones(n, dtype=torch.float64, device=dev)
y = torch.ones(1 << 20, device=dev).sum().item()          # (the runtime and its first kernel are up)
torch.cuda.synchronize()
out = []
for rep in range(3):
    t0 = time.perf_counter()
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
    P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
    out.append(1e3 * (time.perf_counter() - t0))
    path = P.path(); P = None
print("COLD " + json.dumps({"ms": out, "path": path, "n": n, "nnz": nnz}))
"""


def cold_config(g):
    """C2 in a FRESH process through the plain entry (ilupp_hip_ilu0_create_device, no nnz handed in, no shape remembered): the first
    construction + apply of the process (code objects loaded, pools grown, the pattern proven before anything is assumed), then the second
    and third -- what the headline's warm loop does not show."""
    try:
        r = subprocess.run([sys.executable, "-c", _COLD % (ROOT, os.path.join(ROOT, "tests")), str(g)], capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("COLD ")]
        if r.returncode != 0 or not line:
            return {"failed": (r.stderr or r.stdout)[-400:]}
        rec = json.loads(line[0][5:])
        fb, ab = algorithmic_bytes(rec["n"], rec["nnz"])
        ms = rec["ms"]
        return {"workload": "C2COLD: the headline step (ILU0Preconditioner_device without nnz + one apply, %d^3) in a fresh process: first, second, third call" % g,
                "path": rec["path"], "first_ms": ms[0], "second_ms": ms[1], "third_ms": ms[2],
                "first_fraction": (fb + ab) / (ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, "third_fraction": (fb + ab) / (ms[2] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    except Exception as e:
        return {"failed": repr(e)}


def extra_config(name, dev, steps, with_cpu=True):
    """C3 / C4 on device-resident inputs: seconds, factor bytes (read A + write the factors produced), GB/s"""
    import torch
    import matgen
    from ilupp_amd import _native
    if name == "C3":
        d, i, p = matgen.random_dd(1000000, 19, 25.0, 12345)
        make = lambda a: _native.ILUTPreconditioner_device(*a, True, 10, 1e-4)
        what = "C3: ILUTPreconditioner(fill_in=10, threshold=1e-4), random diagonally dominant CSR n=1e6"
    elif name == "ILUC":
        d, i, p = matgen.poisson3d(128)
        make = lambda a: _native.ILUCPreconditioner_device(*a, True, 8, 1e-2)
        what = "ILUC: ILUCPreconditioner(fill_in=8, threshold=1e-2), 3-D 7-point Laplacian 128^3"
    elif name == "C5M":
        # the same matrix with the I-matrix preprocessing BASELINE's default_configuration(10) names: maximum-weight matching (permutation
        # and scalings from the host's augmenting-path search, applied on the device) + the factorisation without pivoting
        import ilupp_amd as ilupp
        d, i, p = matgen.random_dd(1000000, 8, 25.0, 12345)
        prm = ilupp.iluplusplus_precond_parameter()
        prm.default_configuration(1)
        prm.PREPROCESSING.set_MAX_WEIGHTED_MATCHING_ORDERING()
        prm.threshold = 1e-3
        make = lambda a: _native.MultilevelILUCDPPreconditioner_device(*a, True, prm)
        what = "C5M: ILUppPreconditioner(MAX_WEIGHTED_MATCHING_ORDERING, preset 10 without pivoting, threshold=1e-3), random unsymmetric CSR n=1e6 (matching: initialisation on the device, log/exp on the host cores)"
    elif name == "C5P":
        # BASELINE config 5 AS NAMED: default_configuration(10) = maximum weighted matching + the factorisation WITH pivoting (partialILUCDP).
        # The factorisation is a chain of n steps (every step picks its column by the values of the step, its row by the fill so far): one
        # wave of the GPU walks it, ~11-23 us per step -- n = 1e6 takes 23.4 s (profiles/r03_c5p.txt; the reference 6.8 s on one host core),
        # so the default run measures n = 1e5 and says so.
        import ilupp_amd as ilupp
        d, i, p = matgen.random_dd(100000, 8, 25.0, 12345)
        prm = ilupp.iluplusplus_precond_parameter()
        prm.default_configuration(10)
        prm.threshold = 1e-3
        make = lambda a: _native.MultilevelILUCDPPreconditioner_device(*a, True, prm)
        what = ("C5P: ILUppPreconditioner(default_configuration(10): MAX_WEIGHTED_MATCHING_ORDERING + the factorisation WITH pivoting, threshold=1e-3), "
                "random unsymmetric CSR n=1e5 (a chain of n sequential steps on one wave; n=1e6: 23.4 s)")
    elif name == "C5":
        # BASELINE config 5's shape (unsymmetric CSR, n = 1e6) with the multilevel preconditioner this build has: default_configuration(1)
        # = normalisation + PQ ordering + the factorisation WITHOUT pivoting (preset 10): the parameter family whose rows and columns are
        # fixed beforehand and that runs as a dataflow computation on all CUs.  BASELINE names default_configuration(10), whose factorisation
        # pivots (partialILUCDP, a sequential chain): that one is "C5P" -- every line says which one ran.
        import ilupp_amd as ilupp
        d, i, p = matgen.random_dd(1000000, 8, 25.0, 12345)
        prm = ilupp.iluplusplus_precond_parameter()
        prm.default_configuration(1)
        prm.threshold = 1e-3
        make = lambda a: _native.MultilevelILUCDPPreconditioner_device(*a, True, prm)
        what = "C5: ILUppPreconditioner(default_configuration(1): NORMALIZE_COLUMNS+NORMALIZE_ROWS+PQ_ORDERING, preset 10 without pivoting, threshold=1e-3), random unsymmetric CSR n=1e6"
    elif name == "C5L":
        # a matrix whose diagonal is too weak for one level: the PQ ordering keeps a part of the rows per level and the rest goes on as a
        # Schur complement -- the object the driver sees has several levels (construct = every level's ordering + factorisation + Schur rows;
        # apply = the recursion of preconditioner_implementation.h:433-488 over all of them)
        import ilupp_amd as ilupp
        d, i, p = matgen.random_dd(1000000, 3, 0.6, 12345)
        prm = ilupp.iluplusplus_precond_parameter()
        prm.default_configuration(1)
        prm.threshold = 0.3
        make = lambda a: _native.MultilevelILUCDPPreconditioner_device(*a, True, prm)
        what = "C5L: ILUppPreconditioner(default_configuration(1), threshold=0.3), random unsymmetric CSR n=1e6 with a weak diagonal (0.6): a multi-level object"
    elif name in ("C2H", "C2HP"):
        # what a matrix that is NOT the benchmark's box gets: the 256^3 7-point mesh with 3 % of its points removed (x-lines of irregular
        # length: the box-grid templates do not hold, the general analysis decides), and the same matrix under a random symmetric
        # permutation (no line structure left at all)
        d, i, p = matgen.mesh_with_holes(256, 5, 0.03, permute=(name == "C2HP"))
        make = lambda a: _native.ILU0Preconditioner_device(*a, True)
        what = ("%s: ILU0Preconditioner, 7-point mesh 256^3 with 3 %% of its points removed%s" %
                (name, ", rows and columns in a random symmetric permutation" if name == "C2HP" else ""))
    elif name in ("S27", "S9"):
        # ILU(0) beyond the 7-point rows of the headline: box stencils (eliminations meet off-diagonal entries)
        dims = (128, 128, 128) if name == "S27" else (2048, 2048)
        d, i, p = matgen.box_stencil(dims)
        make = lambda a: _native.ILU0Preconditioner_device(*a, True)
        what = "%s: ILU0Preconditioner, %d-point box stencil %s CSR" % (name, 3 ** len(dims), "x".join(map(str, dims)))
    else:
        d, i, p = matgen.poisson3d(256)
        make = lambda a: _native.ICholTPreconditioner_device(*a, True, 0, 0.0)
        what = "C4: ICholTPreconditioner(add_fill_in=0, threshold=0), 3-D 7-point Laplacian 256^3"
    n, nnz = p.shape[0] - 1, int(p[-1])
    td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
    x = torch.ones(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    args = (td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n)
    walls, kms, apps, firsts, first_walls = [], [], [], [], []
    nnz_out = 0
    for rep in range(max(2, steps)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        P = make(args)
        t1 = time.perf_counter()
        tf0 = time.perf_counter()
        P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
        first_wall = (time.perf_counter() - tf0) * 1e3
        first = P.timings()["last_apply_ms"]
        # the first apply of a long-row factor renumbers it by dependency level (sptrsv_lvl.hip) and the first apply of an LL^T object
        # builds its static sweep records; both are reported: "first_apply_ms" and the steady-state "apply_ms" of a Krylov iteration
        x.fill_(1.0)
        P.apply_device(x.data_ptr(), n, transpose=False, sync=True)
        t = P.timings()
        path = P.path() if hasattr(P, "path") else None          # (the multilevel object has no such word: its levels differ)
        knames = P.kernel_names() if hasattr(P, "kernel_names") else ()
        nnz_out = P.total_nnz + (n if name == "C3" else 0)       # stored entries of the factors (ILUT's total_nnz leaves the unit diagonal out)
        if name in ("C5", "C5M", "C5P", "C5L"):
            t["numeric_kernel_ms"] = t["kernel_ms"]
            levels = P.levels()
            nnz_out = sum(sum(P.level_sizes(k)[1:]) for k in range(levels))      # both unit diagonals stored, per level
        if rep:
            walls.append(t1 - t0); kms.append(t["numeric_kernel_ms"]); apps.append(t["last_apply_ms"]); firsts.append(first); first_walls.append(first_wall)
        x.fill_(1.0)
        P = None
    nf = 1 if name == "C4" else 2
    fbytes = (12 * nnz + 4 * (n + 1)) + (12 * nnz_out + 4 * (n + 1) * nf)
    sec = float(np.median(walls))
    kms_med = float(np.median(kms))
    more = {"levels": int(levels)} if name in ("C5", "C5M", "C5P", "C5L") else {}
    del td, ti, tp, x
    torch.cuda.empty_cache()
    cpu = cpu_extra(name, d, i, p) if with_cpu else None
    kernel = EXTRA_KERNEL.get(name)
    if name in ("C2H", "C2HP", "S27", "S9") and knames:
        kernel = knames[0]                     # (the library says which kernel factored: ilupp_hip_kernel_names)
    if path == "icholt:grid-static":
        # ICholT(0, 0) of a box grid: the speculative static kernel (icholt_grid.hip) -- A's pattern assumed for every column, verified
        kernel = "k_icholt_grid"
    ktraffic = measured_traffic(kernel, 256) if kernel in ("k_icholt_grid", "k_ilut_rows_wp") else None
    return {**more, "workload": what, "path": path, "n": n, "nnz": nnz, "factor_nnz": int(nnz_out), "construct_s": sec,
            "numeric_kernel_ms": kms_med, "first_apply_ms": float(np.median(firsts)), "first_apply_wall_ms": float(np.median(first_walls)),
            # what a caller waits for before the first preconditioned iteration can start, and its share of the roofline (factor bytes +
            # one apply's: read the factors, read and write x per solve)
            "construct_plus_first_apply_s": sec + 1e-3 * float(np.median(first_walls)),
            "construct_plus_first_apply_fraction": (fbytes + 12 * nnz_out + 8 * (n + 1) + 32 * n) / (sec + 1e-3 * float(np.median(first_walls))) / 1e9 / HBM_PEAK_GBS,
            "first_apply_note": "first_apply_ms / apply_ms are the sweeps' kernel time (GPU events); first_apply_wall_ms is the wall clock of the FIRST apply call, which also builds what the sweeps need (level order of long-row factors, the static records of an LL^T object's factor pair)",
            "apply_ms": float(np.median(apps)),
            "nnz_per_s": nnz / sec, "factor_bytes": fbytes, "achieved_GBs": fbytes / sec / 1e9,
            "hbm_fraction": fbytes / sec / 1e9 / HBM_PEAK_GBS,
            # the dominant kernel of the construction against the HBM roofline: SURVEY 8(d)'s bytes for these configs (read A + write the
            # factors actually produced) over the kernel's own time; no counter traffic was collected for these kernels
            "roofline": {"bound": "hbm", "kernel": kernel, "algorithmic_bytes": fbytes, "avg_launch_ms": kms_med,
                         "achieved": (fbytes / (kms_med * 1e-3) / 1e9) if kms_med > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (fbytes / (kms_med * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms_med > 0 else None,
                         "traffic": ktraffic,
                         "traffic_over_algorithmic": (ktraffic / fbytes) if ktraffic else None,
                         "traffic_source": (", ".join(sorted(TRAFFIC_SOURCE)) + " (committed counter passes, not this run)") if ktraffic else None},
            "cpu_baseline": cpu}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=256, help="grid points per dimension (256 = BASELINE config C2)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--nrhs", type=int, default=0, help="right-hand sides kept resident (default: one per step, at most 16)")
    ap.add_argument("--cpu-grid", type=int, default=0, help="grid of the CPU baseline sample (default: same as --grid)")
    ap.add_argument("--config", action="append", default=[], choices=["C2", "C2COLD", "C2H", "C2HP", "C3", "C4", "C5", "C5M", "C5P", "C5PB", "C5L", "ILUC", "S27", "S9"],
                    help="extra configs measured after the headline one (C2 is always the bench line; default: C3, C4, C5, C5M, C5L, C5P, C5PB, ILUC)")
    ap.add_argument("--no-extra", action="store_true", help="skip the default extra configs (C3, C4, C5, ILUC) and the refactor loop")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import matgen
    from ilupp_amd import _native
    from ilupp_amd.batched import run_batch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BENCH_SHARE_GPU=1 (tests only): all ranks on GPU 0 with the gloo backend -- the N > 1 code path of this file on a one-GPU box
        # (RCCL refuses two ranks on one device); the measured path is one rank per GPU over RCCL
        share = os.environ.get("BENCH_SHARE_GPU") == "1"
        if share:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if share else "nccl")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    rc = _native.lib().ilupp_hip_set_device(dev.index)
    assert rc == 0

    g = args.grid
    # batched case: rank r factors matrix r of the batch
    d, i, p = batch_matrix(g, rank if world > 1 else 0)
    n, nnz = p.shape[0] - 1, int(p[-1])
    td = torch.from_numpy(d).to(dev)
    ti = torch.from_numpy(i).to(dev)
    tp = torch.from_numpy(p).to(dev)
    # right-hand sides resident in HBM before the timed region starts: one vector of ones per step (apply works in place)
    nrhs = args.nrhs if args.nrhs > 0 else max(1, min(args.steps + args.warmup, 16))
    txs = [torch.ones(n, dtype=torch.float64, device=dev) for _ in range(nrhs)]
    del d, i
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    P = None
    fac_ms, num_ms, ana_ms, app_ms, ls_ms, us_ms, knum_ms = [], [], [], [], [], [], []
    nstep = 0

    def step(record):
        nonlocal P, nstep
        P = None      # release the previous factorisation first (its buffers go back to the allocator)
        P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True, nnz=nnz)
        tx = txs[nstep % nrhs]
        nstep += 1
        P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
        if record:
            t = P.timings()
            ana_ms.append(t["analysis_ms"]); num_ms.append(t["numeric_ms"]); fac_ms.append(t["analysis_ms"] + t["numeric_ms"])
            app_ms.append(t["last_apply_ms"]); ls_ms.append(t["lsolve_kernel_ms"]); us_ms.append(t["usolve_kernel_ms"])
            knum_ms.append(t["numeric_kernel_ms"])

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    wall = time.perf_counter() - t0
    path = P.path()
    analysis_path = P.analysis_path() if hasattr(P, "analysis_path") else "general"
    knames = P.kernel_names() if hasattr(P, "kernel_names") else ()
    # numeric re-factorisation with NEW values on the analysed pattern + one apply (what a time-stepping caller pays per step),
    # timed the same way: ilupp_hip_ilu0_refactor_device reads the new value array where it lies
    refac = None
    if not args.no_extra and world == 1:
        td2 = td * 1.0009765625
        torch.cuda.synchronize()
        rn, ra = [], []
        for _ in range(2):
            P.refactor_device(td2.data_ptr(), ti.data_ptr(), tp.data_ptr())
        k = max(3, min(args.steps, 10))
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        for it in range(k):
            P.refactor_device((td2 if it % 2 == 0 else td).data_ptr(), ti.data_ptr(), tp.data_ptr())
            tx = txs[it % nrhs]
            P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
            t = P.timings()
            rn.append(t["numeric_ms"]); ra.append(t["last_apply_ms"])
        torch.cuda.synchronize()
        rwall = (time.perf_counter() - r0) / k
        refac = {"ms_per_step": 1e3 * rwall, "steps": k, "numeric_ms": float(np.median(rn)), "apply_ms": float(np.median(ra)),
                 "what": "ilupp_hip_ilu0_refactor_device (new values, analysed pattern) + one apply, wall clock per step"}
        # leave the object with the original values (the checksum below is of A itself)
        P.refactor_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr())
        del td2
    # the same step through the entry the reference's constructor maps to (ilupp_hip_ilu0_create_device: no nnz handed in, so the
    # construction reads indptr[n] back before it starts), beside the headline's
    plain = None
    if world == 1:
        k = max(3, min(args.steps, 10))
        for it in range(k + 2):
            if it == 2:
                torch.cuda.synchronize()
                q0 = time.perf_counter()
            Pp = None
            Pp = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
            Pp.apply_device(txs[it % nrhs].data_ptr(), n, transpose=False, sync=True)
        torch.cuda.synchronize()
        pwall = (time.perf_counter() - q0) / k
        Pp = None
        plain = {"ms_per_step": 1e3 * pwall, "steps": k,
                 "what": "ILU0Preconditioner_device without nnz (ilupp_hip_ilu0_create_device, what ILU0Preconditioner(A) on device arrays maps to) + one apply, wall clock per step"}
    # apply(ones) with the last factorisation (untimed): compared, as an array, with the CPU leg's
    tx = txs[0]
    tx.fill_(1.0)
    torch.cuda.synchronize()
    P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
    x_gpu = tx.cpu().numpy()
    checksum = float(x_gpu.sum())

    # what this box's HBM gives a plain device-to-device copy (read + written bytes), next to the 8 TB/s spec peak
    copy_gbs = None
    if rank == 0:
        try:
            src = torch.empty(1 << 27, dtype=torch.float64, device=dev)      # 1 GiB
            dst = torch.empty_like(src)
            src.fill_(1.0); dst.copy_(src); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record(); torch.cuda.synchronize()
            copy_gbs = 5 * 2.0 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except Exception:
            copy_gbs = None

    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())
    ms_per_step = 1e3 * wall / args.steps

    # the batch through the driver of the N > 1 path (ilupp_amd/batched.py): one record per matrix, gathered on
    # every rank; rank 0 then factors every matrix itself and demands byte-identical outputs
    batch = None
    if world > 1:
        def work(member):
            dm, im, pm = batch_matrix(g, member)
            a = [torch.from_numpy(v).to(dev) for v in (dm, im, pm)]
            xb = torch.ones(n, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Pm = _native.ILU0Preconditioner_device(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), n, True)
            Pm.apply_device(xb.data_ptr(), n, transpose=False, sync=True)
            ms = 1e3 * (time.perf_counter() - t0)
            nl, nu = (f[0].shape[0] for f in Pm.factors_info()) if g <= 64 else (0, 0)
            return (member, int(nl), int(nu), int(Pm.total_nnz), hashlib.sha256(xb.cpu().numpy().tobytes()).hexdigest(), ms)
        recs = run_batch(world, work)
        if rank == 0:
            solo = [work(m) for m in range(world)]
            same = all(a[:5] == b[:5] for a, b in zip(recs, solo))
            batch = {"records": [{"matrix": r[0], "total_nnz": r[3], "sha256_apply": r[4][:16], "ms": r[5]} for r in recs],
                     "identical_to_single_rank": bool(same)}
            assert same, "batched outputs differ from the single-rank run"

    # BASELINE config 5 the same way: `world` unsymmetric matrices of n = 1e6 (seeds 12345 + member), one multilevel ILU++ preconditioner
    # (default_configuration(1), threshold 1e-3: the family without pivoting) + one apply each, sharded over the ranks
    # ... first with the preset BASELINE names, default_configuration(10) (the factorisation WITH pivoting: a chain of n steps, so n = 1e5
    # here -- at n = 1e6 one matrix takes tens of seconds, profiles/r03_c5p.txt), then with default_configuration(1) at n = 1e6
    batch_ml = None
    batch_ml10 = None
    if world > 1:
        for preset, nml in ((10, int(os.environ.get("BENCH_C5_N", "1000000"))), (1, int(os.environ.get("BENCH_C5_N", "1000000")))):
            def work_ml(member, preset=preset, nml=nml):
                return ml_batch_member(dev, member, nml, preset)
            barrier()
            b0 = time.perf_counter()
            recs = run_batch(world, work_ml)
            barrier()
            bwall = time.perf_counter() - b0
            if rank == 0:
                solo = [work_ml(m) for m in range(world)]
                same = all(a[:4] == b[:4] for a, b in zip(recs, solo))
                rec = {"what": "C5 batch: ILUppPreconditioner(default_configuration(%d), threshold=1e-3) + apply on %d unsymmetric matrices n=%d, one per rank" % (preset, world, nml),
                       "n": nml, "preset": preset,
                       "cpu_baseline": cpu_batch(world, nml, preset) if not args.no_cpu else None,
                       "records": [{"matrix": r[0], "levels": r[1], "total_nnz": r[2], "sha256_apply": r[3][:16], "ms": r[4]} for r in recs],
                       "wall_s_incl_matrix_generation": bwall, "identical_to_single_rank": bool(same)}
                assert same, "batched multilevel outputs (default_configuration(%d)) differ from the single-rank run" % preset
                if preset == 10:
                    batch_ml10 = rec
                else:
                    batch_ml = rec

    if rank == 0:
        fb, ab = algorithmic_bytes(n, nnz)
        med = lambda v: float(np.median(v)) if v else 0.0
        k_num = med(knum_ms)         # numeric factor kernel alone (HIP events on the library's stream)
        gpu_ms = med(fac_ms) + med(app_ms)
        kernel = FACTOR_KERNEL.get(path, path)
        k_fwd, k_bwd = "k_sptrsv_st<1, false>", "k_sptrsv_st<-1, false>"
        if knames:                              # (the library says which kernels the object runs: ilupp_hip_kernel_names)
            kernel, k_fwd, k_bwd = knames
        # dominant kernel = the numeric factorisation sweep (the longest kernel of the step); its algorithmic bytes = the
        # factor bytes of SURVEY.md section 8(d): read A once + write L and U once
        step_s = wall / args.steps
        k_traffic = measured_traffic(kernel, g)
        phases = [
            {"name": kernel, "what": "numeric factorisation kernel", "ms": k_num, "algorithmic_bytes": fb, "traffic": k_traffic},
            {"name": ("analysis (k_grid_check next to k_grid_lanes + the lane-table kernels)" if analysis_path == "grid"
                      else "analysis (k_row_cuts_counts + lane tables)"), "what": "pattern analysis and schedule, whole phase",
             "ms": med(ana_ms), "algorithmic_bytes": None,
             "traffic": measured_traffic("k_grid_check" if analysis_path == "grid" else "k_row_cuts_counts", g)},
            {"name": k_fwd, "what": "L solve, whole phase", "ms": med(ls_ms),
             "algorithmic_bytes": ab // 2, "traffic": measured_traffic(k_fwd, g)},
            {"name": k_bwd, "what": "U solve, whole phase", "ms": med(us_ms),
             "algorithmic_bytes": ab - ab // 2, "traffic": measured_traffic(k_bwd, g)},
        ]
        step_frac = ((fb + ab) / step_s / 1e9) / HBM_PEAK_GBS
        out = {
            "metric": "ILU(0) factor+apply nnz/s, 3-D 7-pt Poisson fp64",
            "value": world * nnz / (wall / args.steps),
            "unit": "nnz/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: ILU(0) factor + one L/U apply, 3-D 7-point Poisson %d^3 CSR (n=%d, nnz=%d), fp64/int32" % (g, n, nnz),
                       "parallelism": "1 matrix per GPU, no data-path collective" if world > 1 else "single GPU",
                       "step": "full ILU0Preconditioner construction (pattern analysis + row scheduling + numeric factorisation) "
                               "+ one apply, A and x resident in HBM, the caller hands in len(indices) as nnz (ilupp_hip_ilu0_create_device_nnz: "
                               "no read-back before the construction starts); wall clock over all steps",
                       "path": path, "analysis": analysis_path},
            # THE fraction of this line: the whole step (factor + apply bytes of SURVEY 8d) over the wall clock of a step, against the
            # 8 TB/s HBM peak -- the number BASELINE.json's 0.40 target is about.  roofline.frac below is the dominant KERNEL's.
            "headline_fraction": step_frac,
            "headline_fraction_what": "(factor + apply algorithmic bytes of SURVEY 8d) / (wall clock of a step) / 8 TB/s: the whole step, host gaps and analysis included",
            "gpu_ms": {"analysis": med(ana_ms), "numeric": med(num_ms), "factor": med(fac_ms), "apply": med(app_ms),
                       "lsolve": med(ls_ms), "usolve": med(us_ms), "factor_plus_apply": gpu_ms,
                       "numeric_kernel": k_num},
            "gpu_event_value_nnz_per_s": nnz / (gpu_ms * 1e-3) if gpu_ms > 0 else None,
            "hbm_fraction_factor_plus_apply": step_frac,      # (wall clock, = headline_fraction; the sum of the GPU events is gpu_ms.factor_plus_apply)
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": (fb / (k_num * 1e-3) / 1e9) if k_num > 0 else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (fb / (k_num * 1e-3) / 1e9) / HBM_PEAK_GBS if k_num > 0 else None,
                         "frac_note": "achieved / frac use SURVEY 8(d)'s ALGORITHMIC factor bytes (A with its index arrays in, L and U with theirs out) over "
                                      "the numeric kernel's time; the kernel itself reads A's values only and writes value records (the pattern is read by "
                                      "the analysis, the factors' CSR index arrays are made on factors()): frac_bytes_moved is the fraction on the bytes it "
                                      "moves, factor_phase_frac the whole factor phase (analysis + numeric) against the same algorithmic bytes, step_frac the "
                                      "whole step",
                         "frac_bytes_moved": ((k_traffic / (k_num * 1e-3) / 1e9) / HBM_PEAK_GBS) if (k_num > 0 and k_traffic) else None,
                         "factor_phase_frac": ((fb / (med(fac_ms) * 1e-3) / 1e9) / HBM_PEAK_GBS) if med(fac_ms) > 0 else None,
                         "traffic": k_traffic,
                         "traffic_source": (", ".join(sorted(TRAFFIC_SOURCE)) + " (committed PMC passes of this command, not collected in this run)") if TRAFFIC_SOURCE else None,
                         "algorithmic_bytes_per_launch": fb,
                         "avg_launch_ms": k_num,
                         # the whole step against the roofline: factor + apply bytes of section 8(d) over the wall clock of a step
                         "step_frac": step_frac,
                         "step_algorithmic_bytes": fb + ab,
                         "phases": phases,
                         "copy_GBs_measured": copy_gbs,
                         "frac_of_measured_copy": ((fb / (k_num * 1e-3) / 1e9) / copy_gbs) if (k_num > 0 and copy_gbs) else None},
            "checksum": checksum,
        }
        if plain is not None:
            plain["hbm_fraction"] = ((fb + ab) / (plain["ms_per_step"] * 1e-3) / 1e9) / HBM_PEAK_GBS
            plain["value_nnz_per_s"] = nnz / (plain["ms_per_step"] * 1e-3)
            out["plain_entry"] = plain
        if refac is not None:
            refac["hbm_fraction"] = ((fb + ab) / (refac["ms_per_step"] * 1e-3) / 1e9) / HBM_PEAK_GBS
            refac["value_nnz_per_s"] = nnz / (refac["ms_per_step"] * 1e-3)
            out["refactor"] = refac
        if batch is not None:
            out["batch"] = batch
        if batch_ml10 is not None:
            out["batch_ml_config10"] = batch_ml10          # BASELINE config 5 as named
        if batch_ml is not None:
            out["batch_ml"] = batch_ml
        if not args.no_cpu and world == 1:
            cg = args.cpu_grid or g
            rec, x_cpu = cpu_baseline(cg)
            out["cpu_baseline"] = rec
            if cg == g:
                # bit-for-bit: the kernels follow the reference's operation order without FMA contraction
                out["parity"] = {"apply_ones_equal": bool(np.array_equal(x_gpu, x_cpu, equal_nan=True)),
                                 "max_rel_diff": float(np.max(np.abs(x_gpu - x_cpu) / np.maximum(np.abs(x_cpu), 1e-300)))}
                assert out["parity"]["max_rel_diff"] <= 1e-12, "GPU apply(ones) differs from the CPU baseline"
        else:
            out["cpu_baseline"] = None
        extra = {}
        cfgs = list(args.config)
        if not args.no_extra and world == 1:
            cfgs = [c for c in ("C2COLD", "C2H", "C2HP", "C3", "C4", "C5", "C5M", "C5L", "C5P", "C5PB", "ILUC") if c not in cfgs] + cfgs
        for cfg in cfgs:
            if cfg == "C2COLD" and world == 1:
                extra[cfg] = cold_config(g)
            elif cfg in ("C2H", "C2HP", "C3", "C4", "C5", "C5M", "C5P", "C5L", "ILUC", "S27", "S9") and world == 1:
                del_txs = txs[:]        # free the headline workload first
                txs.clear(); del del_txs
                extra[cfg] = extra_config(cfg, dev, max(2, min(args.steps, 3)), with_cpu=not args.no_cpu)
            elif cfg == "C5PB" and world == 1:
                del_txs = txs[:]
                txs.clear(); del del_txs
                extra[cfg] = batch_config(args.steps)
        if extra:
            out["extra"] = extra
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
