#!/bin/bash
# A/B of the grid cap of the analysis' first pass (k_row_cuts_counts): ILUPP_RCC_BLOCKS=... python bench.py (headline only)
for b in 4096 8192 16384 32768 65536; do
  ILUPP_RCC_BLOCKS=$b python3 bench.py --steps 10 --warmup 3 --no-cpu --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('blocks', $b, 'ms_per_step %.4f' % d['ms_per_step'], 'analysis %.4f' % d['gpu_ms']['analysis'], 'numeric %.4f' % d['gpu_ms']['numeric'], 'apply %.4f' % d['gpu_ms']['apply'])
"
done
