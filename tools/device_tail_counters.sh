#!/bin/bash
# SQ counters of k_board_tail (what a frame's workgroup does with its time): separate --pmc passes over tools/device_tail_check.py
# usage (on the GPU box, from the repo root): tools/device_tail_counters.sh > gpurun_out/tail_counters.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"; do
  D=/tmp/tailpmc; rm -rf $D
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $D -- python3 tools/device_tail_check.py 256 0 L8 > /dev/null 2>&1
  python3 - $D <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_board_tail" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print("k_board_tail %-22s mean %.4g per launch of 256 frames (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
