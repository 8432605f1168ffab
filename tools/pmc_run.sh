#!/bin/bash
# Collect PMC counters for the agx kernels, one rocprofv3 pass per counter set (pmc only, no
# other trace domains besides the kernel trace), on a short run of tools/sweep.py.
# usage: tools/pmc_run.sh <outdir> <rows-per-seg>
set -e
OUT=$1; ROWS=${2:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  FRAMES=${FRAMES:-256} rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools/sweep.py $ROWS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum
TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
SETS
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
        v = v[len(v)//2:]  # later dispatches (steady state)
        print("   %-24s mean %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
rm -rf $OUT/p*/
