#!/bin/bash
# Kernel trace of the chain without event pairs on the stream: per-kernel durations AND the gaps between the
# kernels of a batch (rocprofv3 --kernel-trace on tools/sweep.py's untimed loop).
# usage: tools/trace_chain.sh <lib.so> <outfile>
LIB=$1; OUT=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=/tmp/trace_$$; rm -rf $D
AGX_LIBRARY=$PWD/$LIB UNIQUE=256 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 tools/sweep.py 0 > /dev/null 2>&1
python3 - $D $LIB > $OUT <<'PY'
import csv, glob, sys, statistics, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "agx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("agx::", "")
# batches = runs starting with k_blur_hessian
batches, cur = [], []
for r in rows:
    if name(r).startswith("k_blur_hessian") and cur:
        batches.append(cur); cur = []
    cur.append(r)
batches.append(cur)
n = len(batches[0])
batches = [b for b in batches if len(b) == n][5:]  # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list); span = []
for i, b in enumerate(batches):
    for j, r in enumerate(b):
        dur[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if j: gap["%s -> %s" % (name(b[j - 1]), name(r))].append((int(r["Start_Timestamp"]) - int(b[j - 1]["End_Timestamp"])) / 1e3)
    span.append((int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])) / 1e3)
    if i: gap["batch -> batch"].append((int(b[0]["Start_Timestamp"]) - int(batches[i - 1][-1]["End_Timestamp"])) / 1e3)
print(sys.argv[2], "batches", len(batches))
for k, v in dur.items(): print("  %-40s median %.1f us  (min %.1f max %.1f)" % (k, statistics.median(v), min(v), max(v)))
for k, v in gap.items(): print("  gap %-50s median %.1f us" % (k, statistics.median(v)))
print("  first start -> last end of a batch: median %.1f us; sum of kernel medians %.1f us" % (statistics.median(span), sum(statistics.median(v) for v in dur.values())))
PY
rm -rf $D
