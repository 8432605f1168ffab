import os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
frames, _ = synth.render_batch(0, 256, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
det.set_option("sparse_path", 1)
def run(dbg):
    det.set_option("debug_ablation", dbg)
    for _ in range(5): det.saddles_batch_enqueue(frames)
    det.sync()
    det.profile_enable(True); det.profile_reset()
    for _ in range(20): det.saddles_batch_enqueue(frames)
    det.sync(); p = det.profile_read(); det.profile_enable(False)
    return p["k_blur_hessian"][0] / p["k_blur_hessian"][1]
res = {0: [], 1: []}
for r in range(6):
    for d in ((0, 1) if r % 2 == 0 else (1, 0)):
        res[d].append(run(d))
for d in res: print("debug_ablation", d, "K1 median %.4f ms" % statistics.median(res[d]), ["%.4f" % x for x in res[d]])
