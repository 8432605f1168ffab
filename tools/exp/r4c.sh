#!/bin/bash
mkdir -p gpurun_out/r4c
AGX_SPARSE_PATH=2 timeout -k 10 600 python -m pytest tests/test_gpu_bench_geometry.py tests/test_gpu_parity.py tests/test_gpu_memory_safety.py -x -q > gpurun_out/r4c/parity_fused.txt 2>&1; tail -5 gpurun_out/r4c/parity_fused.txt
timeout -k 10 300 python tools/sparse_frame_phases.py > gpurun_out/r4c/phases.txt 2>&1; cat gpurun_out/r4c/phases.txt
timeout -k 10 300 python tools/env_sweep.py '{"AGX_SPARSE_PATH":"1"}' '{"AGX_SPARSE_PATH":"2"}' > gpurun_out/r4c/sweep.txt 2>&1
cat gpurun_out/r4c/sweep.txt
