"""Where the verify kernel's wave time goes (debug_ablation 8192): ticks of s_memrealtime per phase, summed over
all waves of all frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "256"))
frames, _ = synth.render_batch(0, F, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
det.set_option("debug_ablation", 8192)
for _ in range(3):
    det.saddles_batch_enqueue(frames); det.sync()
tot = np.zeros(20, np.int64)
for f in range(F):
    tot += det.debug_fetch(f, "verify_stats").astype(np.int64)
names = ["first loads (mask words)", "block maxima + threshold", "work list + re-tests", "seeds (registers -> LDS)", "list append (global atomic)", "stores + rest"]
us = tot[:6] * 0.01
print("tiles: %d empty, %d with candidates (per frame %.0f / %.0f)" % (tot[6], tot[7], tot[6] / F, tot[7] / F))
for n, v in zip(names, us):
    print("  %-30s %9.0f us of wave time  (%.1f %%)   %.2f us per tile" % (n, v, 100 * v / us.sum(), v / max(tot[6] + tot[7], 1)))
print("  total %.0f us of wave time" % us.sum())
