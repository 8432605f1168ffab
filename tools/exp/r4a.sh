#!/bin/bash
# round 4, experiment A: K1 capped at four workgroups per CU by LDS, K1 in frame groups; serial and three batches in flight
set -e
mkdir -p gpurun_out/r4a
python tools/env_sweep.py '{}' '{"AGX_K1_LDS_KB":"33"}' '{"AGX_K1_GROUP":"64"}' '{"AGX_K1_GROUP":"32"}' '{"AGX_K1_GROUP":"16"}' '{"AGX_K1_GROUP":"32","AGX_K1_LDS_KB":"33"}' > gpurun_out/r4a/sweep.txt 2>&1
cat gpurun_out/r4a/sweep.txt
AGX_K1_GROUP=32 python -m pytest tests/test_gpu_bench_geometry.py -x -q -k "every_frame" > gpurun_out/r4a/parity_group32.txt 2>&1; tail -3 gpurun_out/r4a/parity_group32.txt
for e in "" "AGX_K1_LDS_KB=33" "AGX_K1_GROUP=32" "AGX_K1_LDS_KB=33 AGX_K1_GROUP=32"; do
  tag=$(echo "$e" | tr ' =' '__'); [ -z "$tag" ] && tag=default
  env $e python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline --no-verify > gpurun_out/r4a/bench_$tag.json 2> gpurun_out/r4a/bench_$tag.err
  python - "$tag" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4a/bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "serial", d["ms_per_step"], "median", d.get("ms_per_step_median"), "pipelined", d.get("pipelined",{}).get("ms_per_step"), d["chain"]["kernel_ms_per_step"])
PY
done
