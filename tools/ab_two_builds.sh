#!/bin/bash
# A/B of two builds of libaprilgrid_amd.so on ONE box: the two libraries alternate, three rounds, the same command each time
# (boxes of the pool differ by several per cent, and so do two runs minutes apart: only an alternation on one box separates a
# 1 % change from that).  Round 5's device-tail steps were kept or rejected by this (profiles/README.md).
#   cp aprilgrid-rs_amd/libaprilgrid_amd.so scratch/lib_old.so      # build A (scratch/ travels to the GPU box, git ignores it)
#   ... change, make ...; cp aprilgrid-rs_amd/libaprilgrid_amd.so scratch/lib_new.so
#   gpurun -- 'tools/ab_two_builds.sh scratch/lib_old.so scratch/lib_new.so'
# default command: the device tail on the bench's 256 frames, per-frame ticks and the call's time
A=${1:-scratch/lib_old.so}; B=${2:-scratch/lib_new.so}
shift 2 2>/dev/null
CMD=${*:-env AGX_TAIL_DEBUG=1 python tools/device_tail_check.py 256 0 L8}
cd ${GRAFT_REPO_ROOT:-.}
cp aprilgrid-rs_amd/libaprilgrid_amd.so /tmp/lib_keep.so
for r in 1 2 3; do
  for v in "$A" "$B"; do
    cp "$v" aprilgrid-rs_amd/libaprilgrid_amd.so
    echo "== $v"; $CMD 2>&1 | grep -E "ticks per frame|device tail:" | tail -2 | cut -c1-140
  done
done
cp /tmp/lib_keep.so aprilgrid-rs_amd/libaprilgrid_amd.so
