"""K1 without its blur stores (debug_ablation 1) at five and at four workgroups per CU (AGX_K1_LDS_KB=33): what an occupancy drop costs
once the kernel is bound by its arithmetic."""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
frames, _ = synth.render_batch(0, 256, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
def run(cfg):
    dbg, lds = cfg
    det.set_option("debug_ablation", dbg)
    os.environ["AGX_K1_LDS_KB"] = str(lds)
    for _ in range(5): det.saddles_batch_enqueue(frames)
    det.sync()
    det.profile_enable(True); det.profile_reset()
    for _ in range(20): det.saddles_batch_enqueue(frames)
    det.sync(); p = det.profile_read(); det.profile_enable(False)
    return p["k_blur_hessian"][0] / p["k_blur_hessian"][1]
cfgs = [(0, 0), (0, 33), (1, 0), (1, 33), (1, 41)]
res = {c: [] for c in cfgs}
for r in range(6):
    for c in (cfgs if r % 2 == 0 else cfgs[::-1]):
        res[c].append(run(c))
for c in cfgs: print("debug_ablation %d lds cap %2d KB: K1 median %.4f ms" % (c[0], c[1], statistics.median(res[c])), ["%.4f" % x for x in res[c]])
