#!/bin/bash
# serial and three-in-flight step of the three sparse paths through bench.py, alternating
mkdir -p gpurun_out/r4d
for r in 1 2; do for p in 1 2 3; do
  AGX_SPARSE_PATH=$p python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline --no-verify > gpurun_out/r4d/bench_p${p}_r$r.json 2> gpurun_out/r4d/bench_p${p}_r$r.err
  python - $p $r <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4d/bench_p%s_r%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print("path", sys.argv[1], "serial", d["ms_per_step"], "median", d.get("ms_per_step_median"), "pipelined", d.get("pipelined",{}).get("ms_per_step"), {k:v for k,v in d["chain"]["kernel_ms_per_step"].items() if v})
PY
done; done
python tools/sparse_frame_phases.py 2>&1 | grep -v amdgpu.ids
