"""Per-frame re-test counts of the verify kernel (debug_ablation 128): which frames carry the most, and how
the counts are distributed (the heaviest tiles of the heaviest frame are the launch's critical path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "256"))
frames, _ = synth.render_batch(0, F, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
det.set_option("debug_ablation", 128)
for _ in range(2):
    det.saddles_batch_enqueue(frames); det.sync()
rt = np.zeros(F, np.int64); bits = np.zeros(F, np.int64)
for f in range(F):
    s = det.debug_fetch(f, "verify_stats").astype(np.int64)
    rt[f] = s[[3, 7, 11, 15]].sum(); bits[f] = s[[1, 5, 9, 13]].sum()
print("re-test bits per frame: mean %.0f median %.0f p90 %.0f max %d (frame %d)" % (rt.mean(), np.median(rt), np.percentile(rt, 90), rt.max(), rt.argmax()))
print("candidate bits per frame: mean %.0f max %d (frame %d)" % (bits.mean(), bits.max(), bits.argmax()))
order = np.argsort(-rt)[:10]
print("heaviest frames:", ", ".join("%d: %d of %d" % (f, rt[f], bits[f]) for f in order))
