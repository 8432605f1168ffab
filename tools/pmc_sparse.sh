#!/bin/bash
# HBM / L2 traffic counters of the chain's kernels (separate rocprofv3 --pmc passes on tools/sweep.py).
# usage: tools/pmc_sparse.sh <outdir>      (env AGX_LIBRARY selects another build)
set -e
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  UNIQUE=256 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools/sweep.py 0 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done <<'SETS'
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum
TCC_MISS_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
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
        print("   %-28s mean %.5g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
rm -rf $OUT/p*/
