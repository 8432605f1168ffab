"""Per-format chain time (A/B helper: run under different AGX_LIBRARY builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
det = A.TagDetector("t36h11")
for fmt, n, w, h in (("RGB8", 256, 1280, 800), ("L16", 256, 1280, 800), ("L8", 32, 3840, 2160)):
    b, _ = synth.render_batch(0, 8, w, h, device="cuda", fmt=fmt)
    fr = b.repeat((n // 8 + 1,) + (1,) * (b.dim() - 1))[:n].contiguous()
    for _ in range(5): det.saddles_batch_enqueue(fr)
    det.sync(); t0 = time.perf_counter()
    for _ in range(20): det.saddles_batch_enqueue(fr)
    det.sync(); dt = (time.perf_counter() - t0) / 20
    print(fmt, n, w, h, "ms/step %.4f  Gpix/s %.1f" % (dt * 1e3, n * w * h / dt / 1e9), flush=True)
