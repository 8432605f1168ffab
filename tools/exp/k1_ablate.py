"""K1 under its debug ablations (1 no blur stores, 4 no Hessian / mask, 16 no threshold refresh), alternating in one process;
AGX_LIBRARY selects the build."""
import os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F_ = int(os.environ.get("FRAMES", "256"))
frames, _ = synth.render_batch(0, F_, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
def run(dbg):
    det.set_option("debug_ablation", dbg)
    for _ in range(5): det.saddles_batch_enqueue(frames)
    det.sync()
    det.profile_enable(True); det.profile_reset()
    for _ in range(20): det.saddles_batch_enqueue(frames)
    det.sync(); p = det.profile_read(); det.profile_enable(False)
    return p["k_blur_hessian"][0] / p["k_blur_hessian"][1]
cfgs = [int(x) for x in sys.argv[1:]] or [0, 1, 4, 5, 16, 21]
res = {c: [] for c in cfgs}
for r in range(6):
    for c in (cfgs if r % 2 == 0 else cfgs[::-1]):
        res[c].append(run(c))
for c in cfgs: print("%s debug_ablation %2d: K1 median %.4f ms" % (os.environ.get("AGX_LIBRARY", "default"), c, statistics.median(res[c])))
