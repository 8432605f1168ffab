"""Where the flood + refine kernel's wave time goes (debug_ablation 16384): ticks of s_memrealtime per phase, summed
over the working waves of all frames.  (The diagnostic waits for each batch of window rows as a whole.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "256"))
frames, _ = synth.render_batch(0, F, 1280, 800, device="cuda", pure_noise=os.environ.get("NOISE", "0") == "1")
det = A.TagDetector("t36h11")
FUSED = os.environ.get("FUSED", "0") == "1"  # the flood + refine stage of k_sparse_frame instead of k_flood_refine
det.set_option("sparse_path", 2 if FUSED else 1)
det.set_option("debug_ablation", 262144 if FUSED else 16384)
for _ in range(3):
    det.saddles_batch_enqueue(frames); det.sync()
tot = np.zeros(20, np.int64)
for f in range(F):
    tot += det.debug_fetch(f, "verify_stats").astype(np.int64)
order = [8, 9, 10, 14, 11, 12, 13, 16, 17, 15, 7]
label = {8: "counters + first seed", 9: "flood: window words arrive", 10: "flood: sweeps (slowest lane)", 14: "second flood tier",
         11: "refine: rows 0..4 arrive", 12: "refine: arithmetic rows 0..4", 13: "refine: rows 5..8 arrive", 16: "refine: arithmetic rows 5..8",
         17: "refine: the fit (div, sqrt, acos, atan2)", 15: "refine: record index (atomic)", 7: "record stores + reconvergence"}
chunks = max(int(tot[19]), 1)
print("%d chunks of 64 seeds (%.1f per frame), %d clusters refined" % (chunks, chunks / F, tot[18]))
us = {k: tot[k] * 0.01 for k in order}
total = sum(us.values())
for k in order:
    print("  %-34s %9.0f us of wave time  (%4.1f %%)   %6.2f us per chunk" % (label[k], us[k], 100 * us[k] / total, us[k] / chunks))
print("  total %.0f us of wave time = %.2f us per chunk" % (total, total / chunks))
