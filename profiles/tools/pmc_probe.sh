#!/bin/bash
# usage (on the GPU box): profiles/tools/pmc_probe.sh TAG "COUNTER1 COUNTER2 ..." [GRID]
# one counter pass (--pmc only with --kernel-trace) over a device-resident ILU(0) construct + apply; per kernel: counter averages
export TMPDIR=/tmp
G=${3:-256}
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pp_$1 -- python3 profiles/tools/st_time.py $G > gpurun_out/pp_$1.log 2>&1
python3 - "gpurun_out/pp_$1" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
disp = collections.defaultdict(float); nm = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-28:]
    key = (k, r["Dispatch_Id"], r["Counter_Name"]); disp[key] += float(r["Counter_Value"])
agg = collections.defaultdict(list)
for (k, d, c), v in disp.items(): agg[(k, c)].append(v)
ks = sorted({k for k, c in agg}); cs = sorted({c for k, c in agg})
print("%-28s" % "kernel" + "".join("%22s" % c[-21:] for c in cs))
for k in ks:
    if any(sum(agg[(k, c)]) / len(agg[(k, c)]) > 1e6 for c in cs if (k, c) in agg):
        print("%-28s" % k + "".join("%22.4g" % (sum(agg[(k, c)]) / len(agg[(k, c)])) if (k, c) in agg else "%22s" % "-" for c in cs))
PY
