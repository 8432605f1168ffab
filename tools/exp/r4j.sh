#!/bin/bash
mkdir -p gpurun_out/r4j
AGX_SPARSE_PATH=4 timeout -k 10 300 python -m pytest tests/test_gpu_bench_geometry.py -x -q -k "every_frame or top_shard or other_formats" > gpurun_out/r4j/parity_flow_bench.txt 2>&1; tail -3 gpurun_out/r4j/parity_flow_bench.txt
AGX_SPARSE_PATH=4 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_memory_safety.py -x -q > gpurun_out/r4j/parity_flow.txt 2>&1; tail -3 gpurun_out/r4j/parity_flow.txt
timeout -k 10 300 python tools/env_sweep.py '{"AGX_SPARSE_PATH":"1"}' '{"AGX_SPARSE_PATH":"3"}' '{"AGX_SPARSE_PATH":"2"}' '{"AGX_SPARSE_PATH":"4"}' > gpurun_out/r4j/sweep.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4j/sweep.txt
