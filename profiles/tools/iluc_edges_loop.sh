#!/bin/bash
# The ILUC error-path tests in a loop with a small cache limit (blocks go back to the driver all the time) and the strict pool:
# a double release aborts the process.
export ILUPP_POOL_STRICT=1 ILUPP_CACHE_LIMIT_MB=${1:-8}
N=${2:-100}
for i in $(seq 1 $N); do
  python -m pytest tests/test_gpu_iluc.py -q -x -k "edges" > /tmp/iluc_edges.log 2>&1 || { echo "FAILED at iteration $i"; tail -30 /tmp/iluc_edges.log; exit 1; }
done
echo "iluc edges x$N with ILUPP_CACHE_LIMIT_MB=$ILUPP_CACHE_LIMIT_MB: clean"
