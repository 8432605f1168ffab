#!/bin/bash
# Evidence for profiles/ (round 2): (1) FETCH/WRITE calibration, (2) rocprofv3 --kernel-trace --stats
# of bench.py's main leg, (3) FETCH_SIZE / WRITE_SIZE of the agx kernels (separate --pmc passes) for
# the L8, RGB8 and 4K workloads, (4) the full bench line.
# usage: tools/final_profile_r2.sh <tag>        (on the GPU box, from the repo root)
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_$TAG; mkdir -p $OUT; rm -f $OUT/*.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/calib_$C -- tools/ubench/fetch_calib > /dev/null 2>&1
  python3 - /tmp/calib_$C $C >> $OUT/calibration.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == sys.argv[2]: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print("%-12s %-10s mean %.1f KB per launch of 1048576 KB moved  -> factor %.4f" % (sys.argv[2], k, sum(v)/len(v), sum(v)/len(v)/1048576.0))
PY
done
cat $OUT/calibration.txt
COMMON="--no-cpu-baseline --extra-pipeline 0 --no-extra --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$TAG -- python3 bench.py --steps 50 --warmup 3 $COMMON > $OUT/bench_under_rocprof.json 2> /dev/null
S=$(find /tmp/stats_$TAG -name "*kernel_stats.csv" | head -1); (head -1 $S; grep "agx::" $S) > $OUT/kernel_stats.csv; cut -c1-130 $OUT/kernel_stats.csv
pmc() {  # name, bench args
  NAME=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${TAG}_${NAME}_$C -- python3 bench.py --steps 5 --warmup 2 --settle-ms 0 $COMMON "$@" > /dev/null 2>&1
    python3 - /tmp/pmc_${TAG}_${NAME}_$C $C $NAME >> $OUT/traffic.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "agx::" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for k, v in agg.items():
    v = v[len(v)//2:]
    print("%-6s %-12s %-40s mean %.1f KB per launch (n=%d)" % (sys.argv[3], sys.argv[2], k, sum(v)/len(v), len(v)))
PY
  done
}
pmc L8
pmc RGB8 --format RGB8 --unique 64
pmc 4K --width 3840 --height 2160 --frames 32 --unique 8
cat $OUT/traffic.txt
python3 bench.py --steps 50 --warmup 3 > $OUT/bench.json 2>/dev/null; cut -c1-300 $OUT/bench.json
