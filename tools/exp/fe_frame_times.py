"""What makes a frame's flood + refine stage in k_sparse_frame slow: stage time per frame against seeds, second-tier seeds, clusters."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = 256
frames, _ = synth.render_batch(0, F, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
det.set_option("sparse_path", 3)
det.set_option("debug_ablation", 131072)
for _ in range(5):
    det.saddles_batch_enqueue(frames); det.sync()
raw = np.array([det.debug_fetch(f, "verify_stats").astype(np.int64) for f in range(F)])
st = raw[:, :8] * 0.01
c = [det.debug_fetch(f, "counters") for f in range(F)]
seeds = np.array([x["seeds"] for x in c]); big = np.array([x["big_seeds"] for x in c]); clu = np.array([x["clusters"] for x in c]); ref = np.array([x["refined"] for x in c])
t = st[:, 2] - st[:, 5]
print("floods + refine per frame: median %.1f p90 %.1f max %.1f us" % (np.median(t), np.percentile(t, 90), t.max()))
for name, v in (("seeds", seeds), ("second-tier seeds", big), ("clusters", clu), ("refined", ref)):
    print("  corr with %-18s %.2f   (median %d, max %d)" % (name, np.corrcoef(t, v)[0, 1], np.median(v), v.max()))
A_ = np.stack([seeds, big, np.ones(F)], 1)
coef, *_ = np.linalg.lstsq(A_, t, rcond=None)
print("  least squares: %.1f us + %.4f us per seed + %.2f us per second-tier seed; residual rms %.1f us" % (coef[2], coef[0], coef[1], np.sqrt(np.mean((A_ @ coef - t) ** 2))))
order = np.argsort(-t)[:8]
for f in order: print("  frame %3d: %.1f us, seeds %d, second-tier %d, clusters %d" % (f, t[f], seeds[f], big[f], clu[f]))
