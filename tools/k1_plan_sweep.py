"""The blur kernel's tiling plan (plan_k1: rows per segment x columns per strip) swept on one workload, in ONE process, the
candidates alternating over 6 rounds (order reversed every other round), K1 alone timed by hipEvents (profile level 1).
For every candidate: waves of the launch, waves / resident waves of the chip ("rounds": 256 CUs x 4 SIMDs x the waves per SIMD
the kernel's registers allow), K1 ms (median of the rounds) and the HBM-roofline fraction of its algorithmic bytes.

    WIDTH=3840 HEIGHT=2160 FRAMES=32 UNIQUE=8 FORMAT=L8 python tools/k1_plan_sweep.py [rows,rows,...] [cols,cols,...]

Prints a table and one JSON line (profiles/r6_k1_plan_sweep_*.txt keep them).  VERDICT r5 next #5."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import aprilgrid_rs_amd as A  # noqa: E402
from aprilgrid_rs_amd import synth  # noqa: E402

W_, H_, F_ = int(os.environ.get("WIDTH", "1280")), int(os.environ.get("HEIGHT", "800")), int(os.environ.get("FRAMES", "256"))
U_ = int(os.environ.get("UNIQUE", str(min(F_, 64))))
FMT = os.environ.get("FORMAT", "L8")
IN_B = {"L8": 1, "L16": 2, "RGB8": 3}[FMT]
WAVES_PER_SIMD = int(os.environ.get("WAVES_PER_SIMD", "5"))  # 92 .. 99 VGPRs (tools/kernel_resources.py): 5 of 512 / 96
RESIDENT = 256 * 4 * WAVES_PER_SIMD

rows_list = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 32, 64, 96, 128]
cols_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]

base, _ = synth.render_batch(0, U_, W_, H_, device="cuda", fmt=FMT)
frames = base.repeat((F_ // U_ + 1,) + (1,) * (base.dim() - 1))[:F_].contiguous()
det = A.TagDetector("t36h11")


def run(rows, cols):
    for k in ("AGX_K1_ROWS", "AGX_K1_STRIP_COLS"):
        os.environ.pop(k, None)
    if rows:
        os.environ["AGX_K1_ROWS"] = str(rows)
    if cols:
        os.environ["AGX_K1_STRIP_COLS"] = str(cols)
    det.set_option("reload_tuning_env", 1)
    for _ in range(4):
        det.saddles_batch_enqueue(frames)
    det.sync()
    torch.cuda.synchronize()
    det.set_option("profile_stride", 1)
    det.profile_enable(1)
    det.profile_reset()
    for _ in range(12):
        det.saddles_batch_enqueue(frames)
    det.sync()
    ms, n = det.profile_read()["k_blur_hessian"]
    det.profile_enable(0)
    plan = (det.get_option("k1_rows_per_segment"), det.get_option("k1_segments"), det.get_option("k1_strips"), det.get_option("k1_strip_columns"))
    return ms / max(n, 1), plan


cands = [(r, c) for r in rows_list for c in cols_list]
res = {c: [] for c in cands}
plans = {}
for rnd in range(6):
    for c in (cands if rnd % 2 == 0 else list(reversed(cands))):
        ms, plans[c] = run(*c)
        res[c].append(ms)
px = F_ * W_ * H_
table = []
for c in cands:
    rps, segs, strips, scols = plans[c]
    waves = strips * segs * F_
    ms = statistics.median(res[c])
    table.append({"rows_asked": c[0], "cols_asked": c[1], "rows_per_segment": rps, "segments": segs, "strips": strips, "strip_columns": scols,
                  "waves": waves, "rounds": round(waves / RESIDENT, 2), "k1_ms": round(ms, 4), "k1_ms_min": round(min(res[c]), 4),
                  "frac_of_8TBps": round(px * (IN_B + 4.125) / (ms * 1e-3) / 8e12, 4)})
best = min(table, key=lambda t: t["k1_ms"])
dflt = next((t for t in table if t["rows_asked"] == 0 and t["cols_asked"] == 0), None)
print("%dx%d %s x %d (%d distinct)   resident waves %d" % (W_, H_, FMT, F_, U_, RESIDENT))
print("%5s %5s | %4s %4s %6s %5s | %7s %6s | %8s %8s %7s" % ("rows", "cols", "rps", "segs", "strips", "scols", "waves", "rounds", "K1 ms", "min", "frac"))
for t in table:
    print("%5d %5d | %4d %4d %6d %5d | %7d %6.2f | %8.4f %8.4f %7.4f%s" % (
        t["rows_asked"], t["cols_asked"], t["rows_per_segment"], t["segments"], t["strips"], t["strip_columns"], t["waves"], t["rounds"],
        t["k1_ms"], t["k1_ms_min"], t["frac_of_8TBps"], "  <- default" if t is dflt else ("  <- best" if t is best else "")))
print(json.dumps({"workload": "%dx%d_%s_x%d" % (W_, H_, FMT, F_), "default": dflt, "best": best,
                  "gain_of_best_over_default": round(dflt["k1_ms"] / best["k1_ms"], 4) if dflt else None, "table": table}))
