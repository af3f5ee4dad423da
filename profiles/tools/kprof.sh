#!/bin/bash
# usage (on the GPU box): profiles/tools/kprof.sh TAG [lib.so] -> top kernels of one bench.py run
export TMPDIR=/tmp
[ -n "$2" ] && export ILUPP_HIP_LIBRARY=$PWD/$2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp_$1 -- python3 bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/kp_$1.log 2>&1
f=$(ls gpurun_out/kp_$1/*/*kernel_stats.csv | head -1)
python3 - "$f" "$1" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("==", sys.argv[2])
for r in rows[:int(16)]:
    print("%-44s calls %4s avg %10.1f us" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"])/1e3))
PY
