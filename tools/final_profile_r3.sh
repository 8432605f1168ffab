#!/bin/bash
# Evidence for profiles/ (round 3), all on ONE box in ONE call:
#  (1) FETCH / WRITE calibration, (2) rocprofv3 --kernel-trace --stats of bench.py's main leg, (3) FETCH_SIZE / WRITE_SIZE /
#  TCC counters of the chain's kernels (separate --pmc passes), (4) the wave timeline and K2 phase breakdown of the
#  sparse kernels, (5) kernel trace of the pass with three batches in flight (what overlaps with what), (6) the full
#  bench line.
# usage: tools/final_profile_r3.sh <tag>        (on the GPU box, from the repo root)
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
echo "== pmc traffic"
tools/pmc_sparse.sh $OUT/pmc > $OUT/pmc_traffic.txt 2>&1; cat $OUT/pmc_traffic.txt
echo "== wave timeline"
python3 tools/wave_timeline.py 2>/dev/null > $OUT/wave_timeline.txt; cut -c1-400 $OUT/wave_timeline.txt
python3 tools/verify_phases.py 2>/dev/null > $OUT/verify_phases.txt; cat $OUT/verify_phases.txt
echo "== three batches in flight"
D=/tmp/pipe_$TAG; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 40 --warmup 3 --pipeline 3 $COMMON > $OUT/bench_pipeline3_under_rocprof.json 2> /dev/null
python3 - $D > $OUT/pipeline3_overlap.txt <<'PY'
import csv, glob, sys, statistics, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "agx::" in r["Kernel_Name"]]
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("agx::", "").split("<")[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r)) for r in rows)
ev = ev[len(ev) // 3:]  # steady state
t0, t1 = ev[0][0], max(e[1] for e in ev)
dur = collections.defaultdict(list)
for s, e, n in ev: dur[n].append((e - s) / 1e3)
print("kernels of the pass with three batches in flight (rocprofv3 --kernel-trace), steady state, %d launches" % len(ev))
for n, v in dur.items(): print("  %-18s median %.1f us (min %.1f, max %.1f)" % (n, statistics.median(v), min(v), max(v)))
# time with k_blur_hessian running / with any sparse kernel running / with both
pts = sorted(set([s for s, e, n in ev] + [e for s, e, n in ev]))
k1 = sp = both = none = 0
for a, b in zip(pts[:-1], pts[1:]):
    m = (a + b) // 2
    has1 = any(s <= m < e and n == "k_blur_hessian" for s, e, n in ev)
    hass = any(s <= m < e and n != "k_blur_hessian" for s, e, n in ev)
    d = b - a
    if has1 and hass: both += d
    elif has1: k1 += d
    elif hass: sp += d
    else: none += d
tot = (t1 - t0)
print("  wall %.1f us: blur kernel alone %.1f %%, blur + a sparse kernel %.1f %%, sparse kernels alone %.1f %%, nothing %.1f %%" % (tot / 1e3, 100 * k1 / tot, 100 * both / tot, 100 * sp / tot, 100 * none / tot))
n1 = len(dur["k_blur_hessian"])
print("  per batch: %.1f us of wall time (%d batches)" % (tot / 1e3 / n1, n1))
PY
cat $OUT/pipeline3_overlap.txt
rm -rf $D
python3 bench.py --steps 50 --warmup 3 > $OUT/bench.json 2>/dev/null; cut -c1-300 $OUT/bench.json
python3 tools/kernel_resources.py > $OUT/kernel_resources.txt 2>/dev/null
