#!/bin/bash
# round 5: what the box's host side can do (cores, quota, thread scaling of the tail alone) + detect_batch baseline
set -x
mkdir -p gpurun_out/r5_host_capability
exec > gpurun_out/r5_host_capability/log.txt 2>&1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
cat /proc/self/status | grep -i "cpus_allowed_list\|Threads"
lscpu | head -25
free -g | head -3
which perf || true
g++ -O3 -std=c++17 -pthread -ffp-contract=off -I. tools/tail_scaling/tail_scaling.cpp aprilgrid-rs_amd/csrc/host_tail.cpp -o /tmp/tail_scaling || exit 1
python tools/tail_scaling/dump_cases.py /tmp/cases.bin 256 --gpu || exit 1
/tmp/tail_scaling /tmp/cases.bin 1,2,4,8,16,32,64,128,192,256 4096
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
python - <<'PY'
import sys, time, json
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import aprilgrid_rs_amd as A
import bench_images
det = A.TagDetector("t36h11", None, device=0)
for rep in range(2):
    print(json.dumps(bench_images.detect_batch_table(det, thread_counts=(1, 8, 16, 32, 64, 128))), flush=True)
PY
