"""Summarise a rocprofv3 kernel-trace CSV: per-kernel busy time, overlap, gaps (last N ms)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "agx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) * 2 // 3:]  # steady state
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
span = (t1 - t0) / 1e6
# union busy time and concurrency histogram
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
conc = collections.Counter(); cur = 0; last = ev[0][0]
for t, d in ev:
    conc[cur] += t - last; last = t; cur += d
print("span %.3f ms, kernels %d, streams %s" % (span, len(rows), sorted(set(r.get("Stream_Id", r.get("Queue_Id", "?")) for r in rows))))
print("time by #concurrent kernels (ms):", {k: round(v / 1e6, 3) for k, v in sorted(conc.items())})
per = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", ""); per[k][0] += 1; per[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, us) in per.items(): print("  %-40s n=%4d avg %.1f us" % (k, n, us / n))
