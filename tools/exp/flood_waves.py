"""k_sparse_frame (behind k_verify_seeds) with the floods done by 16 / 8 / 4 of a frame's 16 waves: how far is the stage from being bound by its arithmetic?"""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
frames, _ = synth.render_batch(0, 256, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
det.set_option("sparse_path", 3)
def run(dbg):
    det.set_option("debug_ablation", dbg)
    for _ in range(5): det.saddles_batch_enqueue(frames)
    det.sync()
    det.profile_enable(True); det.profile_reset()
    for _ in range(20): det.saddles_batch_enqueue(frames)
    det.sync(); p = det.profile_read(); det.profile_enable(False)
    return p["k_sparse_frame"][0] / p["k_sparse_frame"][1]
cfg = {0: "16 waves", 2097152: "8 waves", 4194304: "4 waves"}
res = {d: [] for d in cfg}
for r in range(4):
    for d in (list(cfg) if r % 2 == 0 else list(cfg)[::-1]):
        res[d].append(run(d))
for d in cfg: print("floods by %-8s: k_sparse_frame median %.4f ms" % (cfg[d], statistics.median(res[d])), ["%.4f" % x for x in res[d]])
