#!/usr/bin/env python3
"""Summarise the two HBM-traffic counter passes of bench.py into profiles/rNN_pmc_hbm.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -- python3 bench.py --steps 2 --warmup 1 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR_W -- python3 bench.py --steps 2 --warmup 1 --no-cpu
    python3 profiles/tools/pmc_summary.py DIR_F DIR_W profiles/r01_pmc_hbm.json

Per kernel: the average counter value per launch (both counters are in KiB; the gfx950 correction --
FETCH_SIZE reports one half of the loaded bytes -- is applied by bench.py when it reads this file).
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    # one row per (dispatch, counter instance): sum the instances of a dispatch, then average over dispatches
    disp = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")   # (template arguments kept: the L and the U sweep are two kernels)
        key = (k, r["Dispatch_Id"])
        disp[key] += float(r["Counter_Value"])
        name[key] = k
    agg = collections.defaultdict(list)
    for key, v in disp.items():
        agg[name[key]].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("ilupp::"):
            continue
        kernels[k] = {"FETCH_SIZE_KiB_avg_per_launch": fe.get(k, (0.0, 0))[0], "WRITE_SIZE_KiB_avg_per_launch": wr.get(k, (0.0, 0))[0],
                      "launches_fetch": fe.get(k, (0.0, 0))[1], "launches_write": wr.get(k, (0.0, 0))[1]}
    json.dump({"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extra --config C4 --config C3",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-52s fetch %10.1f MiB  write %10.1f MiB  (hbm bytes %.3f GB)" % (
            k, v["FETCH_SIZE_KiB_avg_per_launch"] / 1024, v["WRITE_SIZE_KiB_avg_per_launch"] / 1024,
            (2 * v["FETCH_SIZE_KiB_avg_per_launch"] + v["WRITE_SIZE_KiB_avg_per_launch"]) * 1024 / 1e9))


if __name__ == "__main__":
    main()
