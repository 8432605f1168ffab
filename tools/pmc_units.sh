#!/bin/bash
# Which unit holds the sparse kernels back?  Texture-addresser / L1 / TLB / fabric counters of the chain's kernels
# (separate rocprofv3 --pmc passes on tools/sweep.py, as tools/pmc_sparse.sh).
# usage: tools/pmc_units.sh <outdir>      (env AGX_LIBRARY selects another build)
set -e
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  UNIQUE=256 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools/sweep.py 0 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
  echo "pass $i done"
done <<'SETS'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_max
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_SERIALIZATION_STALL_sum
TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_32B_sum
TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
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
        print("   %-40s mean %.5g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
rm -rf $OUT/p*/
