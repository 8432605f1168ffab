"""The device tail (option "device_tail": board search + decode in tail_kernels.hip) against the host tail on the same frames:
tags per frame (ids, corners bit for bit, order), how many frames the kernel handed back, and the time of both.

    python tools/device_tail_check.py [n_frames] [first_frame] [format: L8 | RGB8 | L16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fmt = sys.argv[3] if len(sys.argv) > 3 else "L8"
W, H = 1280, 800
dev = torch.device("cuda", 0)
frames_t, _ = synth.render_batch(first, n, W, H, device=dev)
frames = frames_t.cpu().numpy()
if fmt == "RGB8":
    frames = np.ascontiguousarray(np.repeat(frames[..., None], 3, axis=3))
elif fmt == "L16":
    frames = (frames.astype(np.uint16) * 257)
det_h = A.TagDetector("t36h11")
det_d = A.TagDetector("t36h11")
det_h.set_option("device_tail", 0)
det_d.set_option("device_tail", 1)
cap = 64


def run(det, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        rc, out, counts, status = det.detect_batch_raw(frames, n_threads=0, cap=cap)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return rc, out, counts, status, best


rc_h, out_h, cnt_h, st_h, t_h = run(det_h)
rc_d, out_d, cnt_d, st_d, t_d = run(det_d)
print("host tail  : rc %d, %.1f tags per frame, %.2f ms per call = %.0f frames/s" % (rc_h, cnt_h.mean(), t_h * 1e3, n / t_h))
print("device tail: rc %d, %.1f tags per frame, %.2f ms per call = %.0f frames/s; handed back to the host %d of %d frames (%d of them for an angle in its guard band)"
      % (rc_d, cnt_d.mean(), t_d * 1e3, n / t_d, det_d.get_option("last_device_tail_fallbacks"), det_d.get_option("last_device_tail_frames"),
         det_d.get_option("last_device_tail_uncertain")))
bad = 0
for f in range(n):
    same = st_h[f] == st_d[f] and cnt_h[f] == cnt_d[f] and out_h[f, : cnt_h[f]].tobytes() == out_d[f, : cnt_d[f]].tobytes()
    if not same:
        bad += 1
        if bad <= 5:
            ih = list(out_h[f, : cnt_h[f]]["id"])
            idd = list(out_d[f, : cnt_d[f]]["id"])
            print("frame %d differs: status %d / %d, %d / %d tags; ids host %s device %s" % (first + f, st_h[f], st_d[f], cnt_h[f], cnt_d[f], ih[:40], idd[:40]))
print("%d of %d frames differ" % (bad, n))
sys.exit(1 if bad else 0)
