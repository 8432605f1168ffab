import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
det = A.TagDetector("t36h11")
frames, _ = synth.render_batch(0, 8, 1280, 800, device="cuda")
det.saddles_batch_enqueue(frames); res, st = det.saddles_batch_fetch()
for i in range(4):
    c = det.debug_fetch(i, "counters"); print(i, c)
img = np.random.default_rng(20014).integers(0, 256, (2, 2), dtype=np.uint8)
try:
    print(det.refined_saddle_points(img, as_array=True)); print(det.debug_fetch(0, "counters"), det.debug_fetch(0, "min"))
except Exception as e: print("ERR", e)
