#!/bin/bash
# Runs on the GPU box (through gpurun): the round's evidence for bench.py's numbers.
#   bash profiles/tools/collect_evidence.sh OUTDIR
# 1. python bench.py (the default command: headline + refactor + C3, C4, ILUC; with the reference CPU baseline) + S27, S9 -> OUTDIR/bench_full.json
# 2. rocprofv3 --kernel-trace --stats of the headline part of the same command + C4 (no CPU leg, no other extras)   -> OUTDIR/kernel_stats.csv
# 3. two counter passes (FETCH_SIZE, WRITE_SIZE; --pmc only with --kernel-trace) of the headline part + C4 + C3  -> OUTDIR/pmc_hbm.json
set -u
OUT=${1:-gpurun_out/evidence}
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 3 --config S27 --config S9 > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
tail -1 "$OUT/bench_full.json" | cut -c1-600
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-extra --config C4 > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_f" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extra --config C4 --config C3 > "$OUT/pmc_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_w" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extra --config C4 --config C3 > "$OUT/pmc_w.log" 2>&1
python3 profiles/tools/pmc_summary.py "$OUT/pmc_f" "$OUT/pmc_w" "$OUT/pmc_hbm.json" > "$OUT/pmc_summary.txt"
cp "$(ls "$OUT"/stats/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
head -12 "$OUT/kernel_stats.csv" | cut -c1-160
