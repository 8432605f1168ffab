#!/bin/bash
# round 5: the result gather beside the chain by steps per collective, one rank through nccl (profiles/r5_gather_steps.txt)
mkdir -p gpurun_out/r5_gather_steps
export MASTER_ADDR=127.0.0.1
C="--gpus 1 --steps 60 --warmup 3 --no-extra --no-cpu-baseline --extra-pipeline 0 --no-verify"
for rep in 1 2 3; do
for k in 0 1 2 4 8; do
  if [ $k = 0 ]; then python bench.py $C > gpurun_out/r5_gather_steps/o.json 2>/dev/null; else MASTER_PORT=$((29600 + RANDOM % 200)) python bench.py $C --collective-world-1 --gather-steps $k > gpurun_out/r5_gather_steps/o.json 2> gpurun_out/r5_gather_steps/o.err; fi
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_gather_steps/o.json') if l.startswith('{')][-1]); print('steps per gather $k (0 = no collective): ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), d['gather_check']['through_collective'])"
done
done
