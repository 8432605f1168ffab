#!/bin/bash
# Evidence for profiles/ (round 6: the chain's kernels are round 5's; the bench, the launch and the tests around them changed), all on ONE box in ONE call:
#  (1) FETCH / WRITE calibration incl. random 64-byte gathers (the sparse kernels' pattern: tools/ubench/fetch_calib.hip gather16 / gather36), (2) rocprofv3 --kernel-trace --stats of bench.py's main leg (K1 + k_verify_seeds + k_sparse_frame),
#  (3) FETCH_SIZE / WRITE_SIZE of the chain's kernels for the L8 / RGB8 / L16 / 4K / noise workloads (separate --pmc passes) and
#  profiles/k1_traffic.json from them, (4) stage times of k_sparse_frame's workgroups and its flood phases, (5) kernel trace of the
#  pass with three batches in flight, (6) the full bench line, (7) kernel resources.
# usage: tools/final_profile_r6.sh <tag>        (on the GPU box, from the repo root)
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
moved = {"gather36": 1572864.0}  # KB of 64-byte line requests (1.5 lines per lane); everything else moves 1 GiB
for k, v in agg.items():
    m = moved.get(k, 1048576.0)
    print("%-12s %-10s mean %.1f KB per launch of %.0f KB %s  -> factor %.4f" % (sys.argv[2], k, sum(v)/len(v), m, "of 64-byte lines asked for" if k.startswith("gather") else "moved", sum(v)/len(v)/m))
PY
done
cat $OUT/calibration.txt
COMMON="--no-cpu-baseline --extra-pipeline 0 --no-extra --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$TAG -- python3 bench.py --steps 50 --warmup 3 $COMMON > $OUT/bench_under_rocprof.json 2> /dev/null
S=$(find /tmp/stats_$TAG -name "*kernel_stats.csv" | head -1); (head -1 $S; grep "agx::" $S) > $OUT/kernel_stats.csv; cut -c1-130 $OUT/kernel_stats.csv
echo "== pmc traffic"
: > $OUT/pmc_traffic.txt
for CFG in "256 1280 800 L8 0" "256 1280 800 RGB8 0" "256 1280 800 L16 0" "32 3840 2160 L8 0" "64 1280 800 L8 1"; do
  set -- $CFG
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${TAG}_$C; rm -rf $D
    FRAMES=$1 WIDTH=$2 HEIGHT=$3 FORMAT=$4 NOISE=$5 UNIQUE=$([ $2 = 3840 ] && echo 8 || echo 64) rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 tools/sweep.py 0 > /dev/null 2>&1
    python3 - $D $C "$1x$2x$3_$4$([ $5 = 1 ] && echo _noise)" >> $OUT/pmc_traffic.txt <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "agx::" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for k, v in agg.items():
    v = v[len(v)//2:]
    print("%s %s %s mean %.6g KB (n=%d)" % (sys.argv[3], k, sys.argv[2], sum(v)/len(v), len(v)))
PY
    rm -rf $D
  done
done
cat $OUT/pmc_traffic.txt
python3 - $OUT/pmc_traffic.txt $TAG > $OUT/k1_traffic.json <<'PY'
import json, sys, collections, re
# bytes per K1 launch = FETCH_SIZE (KB; 64-byte requests: x2 for K1's 128-byte streaming reads, per the calibration) + WRITE_SIZE (KB)
d = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) (agx::\S+?(?:<[^>]*>)?) (FETCH_SIZE|WRITE_SIZE) mean (\S+) KB", line)
    if m and "k_blur_hessian" in m.group(2): d[m.group(1)][m.group(3)] = float(m.group(4))
out = {}
for cfg, c in d.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out[cfg] = {"bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                    "source": "profiles/r6_%s_pmc_traffic.txt (FETCH_SIZE x2 for streaming reads per profiles/r6_%s_calibration.txt, + WRITE_SIZE); a PMC pass of its own on another run -- not a counter of the bench run that quotes it" % (sys.argv[2], sys.argv[2])}
print(json.dumps(out, indent=1))
PY
cat $OUT/k1_traffic.json
python3 - $OUT/pmc_traffic.txt $OUT/calibration.txt > $OUT/sparse_traffic.txt <<'PY'
import re, sys, collections
# the sparse kernels' FETCH_SIZE corrected by the factor the random-gather calibration kernels show on this box
fac = {}
for line in open(sys.argv[2]):
    m = re.match(r"FETCH_SIZE\s+(\S+)\s+mean .* factor (\S+)", line)
    if m: fac[m.group(1)] = float(m.group(2))
g = [fac[k] for k in ("gather16", "gather36") if k in fac and fac[k] > 0]
print("FETCH_SIZE / (64-byte lines asked for) on random gathers: %s -> correction x %.3f (streaming reads: x %.3f)" % (
    {k: fac[k] for k in ("gather16", "gather36") if k in fac}, 1.0 / (sum(g) / len(g)) if g else float("nan"), 1.0 / fac.get("read16", float("nan"))))
d = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"256x1280x800_L8 (agx::\S+?(?:<[^>]*>)?) (FETCH_SIZE|WRITE_SIZE) mean (\S+) KB", line)
    if m and ("k_verify_seeds" in m.group(1) or "k_sparse_frame" in m.group(1)): d[m.group(1)][m.group(2)] = float(m.group(3))
corr = 1.0 / (sum(g) / len(g)) if g else 1.0
for k, c in d.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        print("%-40s raw FETCH %.1f MB + WRITE %.1f MB; lines fetched (gather-corrected) %.1f MB + %.1f MB written = %.1f MB per launch" % (
            k, c["FETCH_SIZE"] / 1024, c["WRITE_SIZE"] / 1024, c["FETCH_SIZE"] * corr / 1024, c["WRITE_SIZE"] / 1024, (c["FETCH_SIZE"] * corr + c["WRITE_SIZE"]) / 1024))
PY
cat $OUT/sparse_traffic.txt
echo "== the device tail beside the chain"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tail_$TAG -- python3 tools/device_tail_stream.py 2048 > $OUT/device_tail_stream.txt 2> /dev/null
S=$(find /tmp/tail_$TAG -name "*kernel_stats.csv" | head -1); (head -1 $S; grep "agx::" $S) > $OUT/device_tail_kernel_stats.csv; cut -c1-130 $OUT/device_tail_kernel_stats.csv
echo "== the driver's command"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2>/dev/null; cut -c1-200 $OUT/bench_driver_command.json
python3 bench.py --steps 50 --warmup 3 > $OUT/bench.json 2>/dev/null; cut -c1-300 $OUT/bench.json
python3 tools/kernel_resources.py > $OUT/kernel_resources.txt 2>/dev/null
echo "== one rank through RCCL's gather, a collective per step (stdout must be the JSON line alone)"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --collective-world-1 --no-extra --no-cpu-baseline > $OUT/bench_collective_world_1.json 2>/dev/null; wc -l $OUT/bench_collective_world_1.json
echo "== plain bench.py --gpus 2 on this box (one GPU: refused early, rc 2)"
python3 bench.py --gpus 2 --steps 2 > $OUT/plain_gpus_2.txt 2>&1; echo "rc=$?" >> $OUT/plain_gpus_2.txt; tail -2 $OUT/plain_gpus_2.txt
