#!/bin/bash
# serial and in-flight step of several library builds, alternating (fresh process per run)
for r in 1 2 3; do
for l in "$@"; do
  AGX_LIBRARY=$PWD/$l python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', 'serial', d['ms_per_step'], 'in flight', d['pipelined']['ms_per_step'], 'K1', d['roofline']['avg_launch_ms'])"
done; done
