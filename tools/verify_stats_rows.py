"""Where K2's re-tested bits sit (debug_ablation bit 2048): by pair of image rows inside the first
mask word row of a K1 segment, and by segment index (0, 1, 2, >= 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "256"))
frames, _ = synth.render_batch(0, F, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
det.set_option("debug_ablation", 2048)
for it in range(3):
    det.saddles_batch_enqueue(frames); det.sync()
tot = np.zeros(20, np.int64)
for f in range(F):
    tot += det.debug_fetch(f, "verify_stats").astype(np.int64)
print("rows/seg", det.get_option("k1_rows_per_segment"))
print("re-test bits per frame in first word rows, by row pair:", " ".join("%.0f" % (v / F) for v in tot[:16]))
print("re-test bits per frame by segment 0,1,2,>=3:", " ".join("%.0f" % (v / F) for v in tot[16:]))
