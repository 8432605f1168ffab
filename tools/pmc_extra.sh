#!/bin/bash
# Extra PMC passes for K1 (instruction fetch, LDS waits, scalar pipe); same conventions as pmc_run.sh.
# usage: tools/pmc_extra.sh <outdir>
set -e
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools/sweep.py 0 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done <<'SETS'
SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU
SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_blur_hessian" not in k: continue
        k = k.split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v)//2:]
        print("   %-24s mean %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
rm -rf $OUT/p*/
