"""K1 and the whole chain with the threshold poll as an awaited scalar load (AGX_K1_ASYNC_POLL=0) and as an asynchronous vector
load (=1), by batch size; alternating in one process."""
import os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
W, H = int(os.environ.get("WIDTH", "1280")), int(os.environ.get("HEIGHT", "800"))
det = A.TagDetector("t36h11")
for F in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256]:
    frames, _ = synth.render_batch(0, F, W, H, device="cuda")
    res = {0: [], 1: []}
    for r in range(6):
        for m in ((0, 1) if r % 2 == 0 else (1, 0)):
            os.environ["AGX_K1_ASYNC_POLL"] = str(m)
            for _ in range(5): det.saddles_batch_enqueue(frames)
            det.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): det.saddles_batch_enqueue(frames)
            det.sync(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
            det.profile_enable(True); det.profile_reset()
            for _ in range(10): det.saddles_batch_enqueue(frames)
            det.sync(); p = det.profile_read(); det.profile_enable(False)
            res[m].append((wall, p["k_blur_hessian"][0] / p["k_blur_hessian"][1]))
    waves = det.get_option("k1_strips") * F * ((H + det.get_option("k1_rows_per_segment") - 1) // det.get_option("k1_rows_per_segment"))
    print("%dx%d x %3d frames (%5d waves): scalar poll chain %.4f K1 %.4f | vector poll chain %.4f K1 %.4f ms" % (
        W, H, F, waves, statistics.median(x[0] for x in res[0]), statistics.median(x[1] for x in res[0]),
        statistics.median(x[0] for x in res[1]), statistics.median(x[1] for x in res[1])), flush=True)
