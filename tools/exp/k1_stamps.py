"""Reads the intra-row timestamps of the instrumented K1 build (tools/exp/libs/stamps.so: s_memtime at fixed points of row 24 of
every wave, written into the response debug plane) and prints the mean cycles per section."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "1"))
frames, _ = synth.render_batch(0, F, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
det.set_option("store_response", 1)
for _ in range(5): det.saddles_batch_enqueue(frames)
det.sync()
resp = det.debug_fetch(0, "resp", (800, 1280))
n_waves = min(4096, det.get_option("k1_strips") * F * ((800 + det.get_option("k1_rows_per_segment") - 1) // det.get_option("k1_rows_per_segment")))
st = resp.reshape(-1).view(np.uint64)[: n_waves * 16].reshape(n_waves, 16)[:, :10].astype(np.int64)
st = st[(st[:, 0] > 0) & (st[:, 6] > 0)]
names = [("group top: copies + 7 loads issued", 8, 9), ("row: LUT reads + horizontal + vertical pass", 0, 1), ("row: blur store issued", 1, 2),
         ("row: Hessian (dv)", 2, 3), ("row: sync check + compare / mask / cmax", 3, 4), ("row: rest (word-row check, row rotate)", 4, 5),
         ("row: to the next row's start", 5, 6), ("whole row (start to next start)", 0, 6)]
print("%d waves with stamps, %d frames, %d rows per segment" % (len(st), F, det.get_option("k1_rows_per_segment")))
for name, a, b in names:
    d = st[:, b] - st[:, a]
    print("  %-46s mean %7.0f  median %7.0f  p90 %7.0f cycles" % (name, d.mean(), np.median(d), np.percentile(d, 90)))
