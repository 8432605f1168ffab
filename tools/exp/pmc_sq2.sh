#!/bin/bash
# SQ instruction counters of the chain's kernels (two --pmc passes on tools/sweep.py, 256 frames)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_sq2; mkdir -p $OUT
i=0
for SET in "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM" "SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  UNIQUE=256 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools/sweep.py 0 > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "agx::" not in k: continue
        k = k.split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v)//2:]
        print("   %-24s mean %.5g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
rm -rf $OUT/p*/
