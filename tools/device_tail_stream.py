"""agx_detect_batch over a long stream of frames (configs[1]'s 256 frames repeated) with the host tail and with the
device tail: frames per second of one call.   python tools/device_tail_stream.py [n_frames] [pinned]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"
dev = torch.device("cuda", 0)
base = synth.render_batch(0, 256, 1280, 800, device=dev)[0].cpu().numpy()
if pinned:
    t = torch.empty((n, 800, 1280), dtype=torch.uint8).pin_memory()
    frames = t.numpy()
else:
    frames = np.empty((n, 800, 1280), np.uint8)
for i in range(0, n, 256):
    frames[i:i + 256] = base[: min(256, n - i)]
cap = 64
out = np.zeros((n, cap), A.TagDetector.TAG_DTYPE)
counts = np.zeros(n, np.uint32)
status = np.zeros(n, np.int32)
for mode in (0, 1):
    det = A.TagDetector("t36h11")
    det.set_option("device_tail", mode)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        rc, _, _, _ = det.detect_batch_raw(frames, n_threads=0, cap=cap, out=out, counts=counts, status=status)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print("%s tail: rc %d, %d frames%s in %.1f ms = %.0f frames/s, %.1f tags per frame%s"
          % ("device" if mode else "host", rc, n, " (pinned)" if pinned else "", best * 1e3, n / best, counts.mean(),
             "; handed back %d" % det.get_option("last_device_tail_fallbacks") if mode else ""), flush=True)
    det.close()
