"""How much of K1's candidate superset the verify kernel has to re-test, by position of the mask
word row inside a K1 segment (debug_ablation bit 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "256"))
frames, _ = synth.render_batch(0, F, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
det.set_option("debug_ablation", 128)
det.saddles_batch_enqueue(frames); det.sync()
tot = np.zeros(20, np.int64)
for f in range(F):
    tot += det.debug_fetch(f, "verify_stats").astype(np.int64)
print("rows/seg", det.get_option("k1_rows_per_segment"))
for q in range(4):
    w, b, wr, br = tot[q * 4:q * 4 + 4] / F
    print("word row %d of segment: words %.0f bits %.0f | re-test words %.0f bits %.0f | stay %.0f   (per frame)" % (q, w, b, wr, br, tot[16 + q] / F))
print("total bits %.0f re-test %.0f stay %.0f" % (tot[[1, 5, 9, 13]].sum() / F, tot[[3, 7, 11, 15]].sum() / F, tot[16:20].sum() / F))
