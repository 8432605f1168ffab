"""Serial step of the sparse paths (1 = three launches, 3 = k_verify_seeds + k_sparse_frame) by batch size."""
import os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
base, _ = synth.render_batch(0, 256, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
for F in (96, 128, 160, 192, 224, 256, 320, 384, 448, 512):
    frames = base.repeat((F // 256 + 1, 1, 1))[:F].contiguous()
    res = {1: [], 3: []}
    for r in range(4):
        for p in ((1, 3) if r % 2 == 0 else (3, 1)):
            det.set_option("sparse_path", p)
            for _ in range(3): det.saddles_batch_enqueue(frames)
            det.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(15): det.saddles_batch_enqueue(frames)
            det.sync(); torch.cuda.synchronize(); res[p].append((time.perf_counter() - t0) / 15 * 1e3)
    det.set_option("sparse_path", 0)
    det.saddles_batch_enqueue(frames); det.sync()
    print("frames %3d: three launches %.4f ms, verify + one workgroup per frame %.4f ms (%+.1f us); chosen by the library: %d" % (
        F, statistics.median(res[1]), statistics.median(res[3]), 1e3 * (statistics.median(res[3]) - statistics.median(res[1])), det.get_option("last_sparse_path")), flush=True)
