#!/bin/bash
# round 4 stress set: randomised parity with each sparse path forced, concurrency (three detectors in flight) with the paths forced and
# on the benchmarked geometry, detect_batch, threads
mkdir -p gpurun_out/r4_stress
for p in 2 3 1; do
  AGX_SPARSE_PATH=$p timeout -k 10 400 python tools/stress_parity.py 300 $((40 + p)) > gpurun_out/r4_stress/parity_path$p.txt 2>&1; echo "parity path $p:"; tail -1 gpurun_out/r4_stress/parity_path$p.txt
done
for p in 2 3; do
  AGX_SPARSE_PATH=$p timeout -k 10 300 python tools/stress_concurrency.py 300 3 > gpurun_out/r4_stress/concurrency_path$p.txt 2>&1; echo "concurrency path $p:"; tail -2 gpurun_out/r4_stress/concurrency_path$p.txt
done
timeout -k 10 300 python tools/stress_concurrency.py 150 3 big > gpurun_out/r4_stress/concurrency_big.txt 2>&1; echo "concurrency big:"; tail -2 gpurun_out/r4_stress/concurrency_big.txt
timeout -k 10 200 python tools/stress_threads.py > gpurun_out/r4_stress/threads.txt 2>&1; echo "threads:"; tail -2 gpurun_out/r4_stress/threads.txt
