"""GPU tuning helper: per-kernel times of the chain for a list of K1 rows-per-segment values."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "256"))
rows = [int(x) for x in (sys.argv[1:] or ["0"])]
U = int(os.environ.get("UNIQUE", "32"))
W, H = int(os.environ.get("WIDTH", "1280")), int(os.environ.get("HEIGHT", "800"))
NOISE = os.environ.get("NOISE", "0") == "1"
FMT = os.environ.get("FORMAT", "L8")
base, _ = synth.render_batch(0, U, W, H, device="cuda", fmt=FMT, pure_noise=NOISE)
frames = base.repeat((F // U + 1,) + (1,) * (base.dim() - 1))[:F].contiguous()
det = A.TagDetector("t36h11")
import time
DBG = [int(x) for x in os.environ.get("DBG", "0").split(",")]
for r, dbg in [(r, d) for r in rows for d in DBG]:
    det.set_option("k1_rows_per_segment", r)
    det.set_option("debug_ablation", dbg)
    for _ in range(3):
        det.saddles_batch_enqueue(frames)
    det.sync()
    det.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
    det.profile_enable(True); det.profile_reset()
    for _ in range(10):
        det.saddles_batch_enqueue(frames)
    det.sync()
    p = det.profile_read(); det.profile_enable(False)
    print(r, "dbg", dbg, {k: round(v[0] / v[1], 4) for k, v in p.items() if v[1]}, "sum", round(sum(v[0] / v[1] for v in p.values() if v[1]), 4), "WALL ms/step", round(wall, 4), flush=True)
