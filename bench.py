#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Metric: ILU(0) factor + one L/U apply on the 3-D 7-point Poisson 256^3 CSR matrix (fp64 values,
int32 indices), reported as nnz(A)/s, with the achieved fraction of the HBM roofline for the
dominant kernel and the reference's own CPU path timed beside it.

A "step" = one complete ILU(0) factorisation of the device-resident CSR matrix (symbolic pattern
split + row scheduling + numeric factorisation) followed by one apply() on a device-resident
vector, i.e. exactly what `P = ilupp.ILU0Preconditioner(A); P.apply(x)` does, with A and x already
in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid G] [--no-cpu]

N > 1: launched by torch.distributed.run, one rank per GPU; the path does not shard (a single
factorisation is one dependency chain), so every rank factors its own matrix of the batch
(weak scaling, no data-path collective; RCCL only for the barrier / max-over-ranks timing).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def algorithmic_bytes(n, nnz):
    """factor = read A once + write L,U once; apply = read L,U once + read/write x per solve."""
    factor = (12 * nnz + 4 * (n + 1)) + (12 * (nnz + n) + 8 * (n + 1))
    apply_ = 12 * (nnz + n) + 8 * (n + 1) + 32 * n
    return factor, apply_


def measured_traffic(kernel, g):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r*_pmc_hbm.json, collected with
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs of this same command).  gfx950
    correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports one half of the loaded bytes, WRITE_SIZE
    is taken as reported; both counters are in KiB.  Only valid for the 256^3 workload it was measured on."""
    import glob
    if g != 256:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"]["ilupp::" + kernel]
        return (2.0 * k["FETCH_SIZE_KiB_avg_per_launch"] + k["WRITE_SIZE_KiB_avg_per_launch"]) * 1024.0
    except Exception:
        return None


def cpu_baseline(g, want_ref=True):
    """The reference's own C++ path (oracle/_ref, kind "reference") or the plain-C restatement
    (kind "port") on ONE host core (the reference is single-threaded): ILU(0) factor + one apply on the
    same workload shape.  Bounded: one pass over the workload (about 10-15 s at 256^3)."""
    import matgen
    from oracle import oracle as O
    kind = "reference" if (want_ref and O.ref_available()) else "port"
    lib = O.ref() if kind == "reference" else O.orc()
    d, i, p = matgen.poisson3d(g)
    n, nnz = p.shape[0] - 1, int(p[-1])
    t0 = time.perf_counter()
    L, U = lib.ilu0((d, i, p, True))
    t1 = time.perf_counter()
    x = lib.apply_lu(L, U, np.ones(n), O.ID)
    t2 = time.perf_counter()
    try:
        cpu = open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip()
    except Exception:
        cpu = "unknown"
    return {"value": nnz / (t2 - t0), "unit": "nnz/s", "cores": 1, "kind": kind,
            "sample": "the full workload once: 3-D 7-pt Poisson %d^3 (n=%d, nnz=%d); factor %.3f s + apply %.3f s "
                      "through the C-ABI wrapper (includes copying L/U out and in); host CPU: %s, %d cores present"
                      % (g, n, nnz, t1 - t0, t2 - t1, cpu, os.cpu_count() or 0),
            "factor_s": t1 - t0, "apply_s": t2 - t1, "checksum": float(np.sum(x))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=256, help="grid points per dimension (256 = BASELINE config C2)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--nrhs", type=int, default=0, help="right-hand sides kept resident (default: one per step, at most 16)")
    ap.add_argument("--cpu-grid", type=int, default=0, help="grid of the CPU baseline sample (default: same as --grid)")
    args = ap.parse_args()

    import torch
    import matgen
    from ilupp_amd import _native

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    rc = _native.lib().ilupp_hip_set_device(dev.index)
    assert rc == 0

    g = args.grid
    d, i, p = matgen.poisson3d(g)
    # batched case: rank r factors its own matrix of the batch (distinct diagonal shift)
    if world > 1:
        d = d + np.where(d > 0, 0.01 * rank, 0.0)
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
        P = None      # release the previous factorisation first (its buffers go back to the HIP allocator)
        P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
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
    # checksum of apply(ones) with the last factorisation (untimed; compared with the CPU baseline's)
    tx = txs[0]
    tx.fill_(1.0)
    torch.cuda.synchronize()
    P.apply_device(tx.data_ptr(), n, transpose=False, sync=True)
    checksum = float(tx.sum().item())

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

    if rank == 0:
        fb, ab = algorithmic_bytes(n, nnz)
        med = lambda v: float(np.median(v)) if v else 0.0
        k_num = med(knum_ms)         # numeric factor kernel alone (HIP events on the library's stream)
        gpu_ms = med(fac_ms) + med(app_ms)
        reuse_ms = med(num_ms) + med(app_ms)
        # dominant kernel = the numeric factorisation sweep; its algorithmic bytes = the factor bytes of
        # read A once + write L and U once
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
                               "+ one apply, A and x resident in HBM; wall clock over all steps"},
            "gpu_ms": {"analysis": med(ana_ms), "numeric": med(num_ms), "factor": med(fac_ms), "apply": med(app_ms),
                       "lsolve": med(ls_ms), "usolve": med(us_ms), "factor_plus_apply": gpu_ms,
                       "numeric_kernel": k_num},
            "gpu_event_value_nnz_per_s": nnz / (gpu_ms * 1e-3) if gpu_ms > 0 else None,
            "pattern_reuse_value_nnz_per_s": nnz / (reuse_ms * 1e-3) if reuse_ms > 0 else None,
            "hbm_fraction_factor_plus_apply": ((fb + ab) / (gpu_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if gpu_ms > 0 else None,
            "hbm_fraction_pattern_reuse": ((fb + ab) / (reuse_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if reuse_ms > 0 else None,
            "roofline": {"bound": "hbm", "kernel": "k_ilu0_lm",
                         "achieved": (fb / (k_num * 1e-3) / 1e9) if k_num > 0 else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (fb / (k_num * 1e-3) / 1e9) / HBM_PEAK_GBS if k_num > 0 else None,
                         "traffic": measured_traffic("k_ilu0_lm", g),
                         "algorithmic_bytes_per_launch": fb,
                         "avg_launch_ms": k_num,
                         "copy_GBs_measured": copy_gbs,
                         "frac_of_measured_copy": ((fb / (k_num * 1e-3) / 1e9) / copy_gbs) if (k_num > 0 and copy_gbs) else None},
            "checksum": checksum,
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_grid or g)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
