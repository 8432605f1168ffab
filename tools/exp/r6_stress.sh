#!/bin/bash
# round 6 stress set on the round's build: randomised parity (a quarter of the cases have unaligned widths: K1's unaligned form) with the default
# paths and with the byte-gathering form forced; three detectors in flight; agx_detect_batch across thread counts against the oracle's detect; threads
mkdir -p gpurun_out/r6_stress
timeout -k 10 500 python tools/stress_parity.py 600 51 > gpurun_out/r6_stress/parity_default.txt 2>&1; echo "parity, default forms:"; tail -1 gpurun_out/r6_stress/parity_default.txt
AGX_K1_UNALIGNED_FAST=0 timeout -k 10 400 python tools/stress_parity.py 300 52 > gpurun_out/r6_stress/parity_generic.txt 2>&1; echo "parity, byte-gathering form forced:"; tail -1 gpurun_out/r6_stress/parity_generic.txt
AGX_SPARSE_PATH=2 timeout -k 10 400 python tools/stress_parity.py 300 53 > gpurun_out/r6_stress/parity_path2.txt 2>&1; echo "parity, sparse path 2:"; tail -1 gpurun_out/r6_stress/parity_path2.txt
timeout -k 10 300 python tools/stress_concurrency.py 300 3 > gpurun_out/r6_stress/concurrency.txt 2>&1; echo "concurrency:"; tail -2 gpurun_out/r6_stress/concurrency.txt
timeout -k 10 300 python tools/stress_detect_batch.py > gpurun_out/r6_stress/detect_batch.txt 2>&1; echo "detect_batch:"; tail -3 gpurun_out/r6_stress/detect_batch.txt
timeout -k 10 200 python tools/stress_threads.py > gpurun_out/r6_stress/threads.txt 2>&1; echo "threads:"; tail -2 gpurun_out/r6_stress/threads.txt
# round 6: the device tail against the host tail on a larger synthetic set (its upload scheduling and buffers changed this round)
timeout -k 10 500 python tools/device_tail_stress.py 2048 > gpurun_out/r6_stress/device_tail.txt 2>&1; echo "device tail stress:"; tail -3 gpurun_out/r6_stress/device_tail.txt
