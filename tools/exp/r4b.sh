#!/bin/bash
# round 4, experiment B: the fused per-frame sparse kernel (AGX_SPARSE_PATH=2) against the three launches (=1)
mkdir -p gpurun_out/r4b
AGX_SPARSE_PATH=2 timeout -k 10 600 python -m pytest tests/test_gpu_bench_geometry.py tests/test_gpu_parity.py -x -q > gpurun_out/r4b/parity_fused.txt 2>&1; tail -5 gpurun_out/r4b/parity_fused.txt
timeout -k 10 300 python tools/env_sweep.py '{"AGX_SPARSE_PATH":"1"}' '{"AGX_SPARSE_PATH":"2"}' > gpurun_out/r4b/sweep.txt 2>&1
cat gpurun_out/r4b/sweep.txt
