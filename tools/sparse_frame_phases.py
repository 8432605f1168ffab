"""When the workgroups of k_sparse_frame pass their stages (debug_ablation 131072: 10 ns ticks in the frame's stats[0..3]):
durations of verify / flood + refine / emission per frame and the launch's span.  env FRAMES, WIDTH, HEIGHT, NOISE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "256")); W = int(os.environ.get("WIDTH", "1280")); H = int(os.environ.get("HEIGHT", "800"))
frames, _ = synth.render_batch(0, F, W, H, device="cuda", pure_noise=os.environ.get("NOISE", "0") == "1")
det = A.TagDetector("t36h11")
det.set_option("sparse_path", 2)
det.set_option("debug_ablation", 131072)
for _ in range(5):
    det.saddles_batch_enqueue(frames); det.sync()
raw = np.array([det.debug_fetch(f, "verify_stats").astype(np.int64) for f in range(F)])
st = raw[:, :8] * 0.01  # us
cnt = np.array([[det.debug_fetch(f, "counters")[k] for k in ("seeds", "clusters", "refined")] for f in range(F)])
t0 = st[:, 0].min()
print("frames %d: launch span %.1f us (first start -> last end); starts within %.1f us" % (F, st[:, 3].max() - t0, st[:, 0].max() - t0))
for name, a, b in (("verify", 0, 1), ("seeds+flood+refine", 1, 2), ("emit", 2, 3), ("whole frame", 0, 3)):
    d = st[:, b] - st[:, a]
    print("%-13s median %6.1f  p10 %6.1f  p90 %6.1f  max %6.1f (frame %d)" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max(), d.argmax()))
for name, a, b in (("verify 1+2 (re-tests)", 0, 4), ("seed pass", 1, 5), ("floods + refine", 5, 2)):
    d = st[:, b] - st[:, a]
    print("  %-22s median %6.1f  p90 %6.1f  max %6.1f (frame %d)" % (name, np.median(d), np.percentile(d, 90), d.max(), d.argmax()))
w0 = raw[:, 8:14] * 0.01
print("  wave 0 of the frame, us (median over frames): words wait %.1f, slots + list %.1f, re-tests %.1f, write-back %.1f"
      % tuple(np.median(w0[:, :4], axis=0)))
rt = raw[:, 14]
v12 = st[:, 4] - st[:, 0]
print("re-tested bits per frame: median %d p90 %d max %d; verify 1+2 against them: %s" % (np.median(rt), np.percentile(rt, 90), rt.max(),
      ", ".join("%d..%d bits: %.1f us" % (lo, hi, np.median(v12[(rt >= lo) & (rt < hi)])) for lo, hi in ((0, 2000), (2000, 3000), (3000, 4000), (4000, 6000), (6000, 10000), (10000, 10**9)) if ((rt >= lo) & (rt < hi)).any())))
print("verify done at: median %.1f max %.1f; flood done at: median %.1f max %.1f" % (np.median(st[:, 1]) - t0, st[:, 1].max() - t0, np.median(st[:, 2]) - t0, st[:, 2].max() - t0))
print("seeds per frame: median %d max %d; rounds of 1024: %s" % (np.median(cnt[:, 0]), cnt[:, 0].max(), np.bincount((cnt[:, 0] + 1023) // 1024).tolist()))
d = st[:, 2] - st[:, 1]
print("flood+refine by rounds:", {int(r): round(float(np.median(d[(cnt[:, 0] + 1023) // 1024 == r])), 1) for r in np.unique((cnt[:, 0] + 1023) // 1024)})
