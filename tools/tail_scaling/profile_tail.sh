#!/bin/bash
# The host tail per thread on this box: tails per second (tools/tail_scaling, one thread) and an rdtsc-section profile of
# host_tail.cpp (tsc_patch.py inserts the counters into a copy; tp_tsc.cpp prints them) -- the numbers of profiles/r5_host_tail_speed.txt.
# An optional second argument is another version of host_tail.cpp to measure beside it (e.g. `git show <commit>:aprilgrid-rs_amd/csrc/host_tail.cpp`).
# usage (GPU box, repo root): tools/tail_scaling/profile_tail.sh [frames=64] [other_host_tail.cpp]
N=${1:-64}; OTHER=$2; R=$PWD; CL=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
python tools/tail_scaling/dump_cases.py /tmp/cases.bin $N --gpu || exit 1
python - <<'PY'
import struct
f=open('/tmp/cases.bin','rb'); w,h,n=struct.unpack('iii',f.read(12)); ns,=struct.unpack('i',f.read(4)); s=f.read(20*ns); g=f.read(w*h)
open('/tmp/case.bin','wb').write(struct.pack('iii',w,h,ns)+s+g)
PY
rm -f /tmp/ts_other /tmp/tp_other
for v in cur other; do
  SRC=aprilgrid-rs_amd/csrc/host_tail.cpp; if [ $v = other ]; then SRC=$OTHER; fi; if [ -z "$SRC" ]; then continue; fi
  mkdir -p /tmp/v_$v && python tools/tail_scaling/tsc_patch.py $SRC /tmp/v_$v/host_tail_tsc.cpp
  $CL -O3 -std=c++17 -ffp-contract=off -I$R -I$R/aprilgrid-rs_amd/csrc tools/tail_scaling/tp_tsc.cpp /tmp/v_$v/host_tail_tsc.cpp -o /tmp/tp_$v || exit 1
  $CL -O3 -std=c++17 -pthread -ffp-contract=off -I$R -I$R/aprilgrid-rs_amd/csrc tools/tail_scaling/tail_scaling.cpp $SRC -o /tmp/ts_$v || exit 1
done
for i in 1 2 3; do for v in cur other; do if [ -x /tmp/ts_$v ]; then echo "== $v $(NO_SPIN=1 /tmp/ts_$v /tmp/cases.bin 1 1024 | tail -1)"; fi; done; done
for v in cur other; do if [ -x /tmp/tp_$v ]; then echo "== $v"; /tmp/tp_$v /tmp/case.bin; fi; done
exit 0
