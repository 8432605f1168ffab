"""K1 with the 4 KB table (AGX_K1_WIDE=0) and with the sixteen-copy table (AGX_K1_WIDE=1), alternating in one process; results compared."""
import os, sys, statistics
sys.path.insert(0, ".")
import torch, numpy as np
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
fmt = os.environ.get("FORMAT", "L8")
frames, _ = synth.render_batch(0, 256, 1280, 800, device="cuda", fmt=fmt)
det = A.TagDetector("t36h11")
def run(wide):
    os.environ["AGX_K1_WIDE"] = str(wide)
    for _ in range(5): det.saddles_batch_enqueue(frames)
    det.sync()
    det.profile_enable(True); det.profile_reset()
    for _ in range(20): det.saddles_batch_enqueue(frames)
    det.sync(); p = det.profile_read(); det.profile_enable(False)
    return p["k_blur_hessian"][0] / p["k_blur_hessian"][1]
outs = {}
for w in (0, 1):
    os.environ["AGX_K1_WIDE"] = str(w)
    det.saddles_batch_enqueue(frames)
    res, st = det.saddles_batch_fetch()
    assert (st == 0).all()
    outs[w] = [np.asarray(r).tobytes() for r in res]
print("lists equal:", outs[0] == outs[1], "frames", len(outs[0]))
res = {0: [], 1: []}
for r in range(6):
    for c in ((0, 1) if r % 2 == 0 else (1, 0)):
        res[c].append(run(c))
for c in res: print("%s AGX_K1_WIDE=%d: K1 median %.4f ms" % (fmt, c, statistics.median(res[c])), ["%.4f" % x for x in res[c]])
