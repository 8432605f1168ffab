"""Per-frame workload of the sparse kernels on the bench's frames: seeds, second-tier seeds, clusters, refined
records, saddles (mean / min / max over the batch).   usage: python tools/frame_counts.py   (env FRAMES, UNIQUE)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth

F = int(os.environ.get("FRAMES", "256"))
W, H = int(os.environ.get("WIDTH", "1280")), int(os.environ.get("HEIGHT", "800"))
frames, _ = synth.render_batch(0, F, W, H, device="cuda")
det = A.TagDetector("t36h11")
det.saddles_batch_enqueue(frames)
det.sync()
rows = [det.debug_fetch(f, "counters") for f in range(F)]
for k in ("seeds", "big_seeds", "clusters", "refined", "saddles"):
    v = np.array([r[k] for r in rows])
    print("%-10s mean %8.1f  min %6d  max %6d  sum %8d   chunks of 64: %d" % (k, v.mean(), v.min(), v.max(), v.sum(), int(np.ceil(v / 64).sum())))
