#!/bin/bash
mkdir -p gpurun_out/r4f
python tools/env_sweep.py '{}' '{"AGX_BLUR_SEL":"1"}' > gpurun_out/r4f/sweep.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4f/sweep.txt
AGX_BLUR_SEL=1 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r4f/bench_sel.json 2> gpurun_out/r4f/bench_sel.err; tail -3 gpurun_out/r4f/bench_sel.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4f/bench_sel.json").read().strip().splitlines()[-1])
print("SEL serial", d["ms_per_step"], "median", d.get("ms_per_step_median"), "pipelined", d.get("pipelined",{}).get("ms_per_step"), "verified", d.get("verified_frames"), {k:v for k,v in d["chain"]["kernel_ms_per_step"].items() if v})
PY
