for r in 1 2 3; do
for l in cur k1w4; do
  AGX_LIBRARY=$PWD/ab/libagx_$l.so python bench.py --steps 40 --warmup 3 --no-extra --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', 'serial', d['ms_per_step'], 'pipelined', d['pipelined']['ms_per_step'], 'K1', d['roofline']['avg_launch_ms'])"
done; done
