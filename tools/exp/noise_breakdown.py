import os, sys, time
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "64"))
base, _ = synth.render_batch(0, 16, 1280, 800, device="cuda", pure_noise=True)
frames = base.repeat((F // 16 + 1, 1, 1))[:F].contiguous()
det = A.TagDetector("t36h11")
for path in (1, 3):
    det.set_option("sparse_path", path)
    for _ in range(5): det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
    det.profile_enable(True); det.profile_reset()
    for _ in range(10): det.saddles_batch_enqueue(frames)
    det.sync(); p = det.profile_read(); det.profile_enable(False)
    print("frames", F, "path", path, "wall %.4f" % wall, {k: round(v[0] / max(v[1], 1), 4) for k, v in p.items() if v[1]})
    c = det.debug_fetch(0, "counters"); print("   frame 0:", c)
