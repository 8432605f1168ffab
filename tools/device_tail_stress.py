"""Many synthetic frames through agx_detect_batch with the device tail and with the host tail: tags must be equal bit for bit,
frame by frame; counts the frames the kernel handed back.   python tools/device_tail_stress.py [frames per geometry]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth

per = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
det_h = A.TagDetector("t36h11")
det_h.set_option("device_tail", 0)
det_d = A.TagDetector("t36h11")
det_d.set_option("device_tail", 1)
cap = 128
total = bad = back = unc = tags = 0
t0 = time.time()
for (w, h, fmt, first) in ((1280, 800, "L8", 256), (640, 480, "L8", 100000), (960, 600, "RGB8", 200000), (800, 608, "L16", 300000), (1920, 1080, "L8", 400000)):
    n_geo = per if w < 1900 else per // 4
    for base in range(0, n_geo, 256):
        n = min(256, n_geo - base)
        fr, _ = synth.render_batch(first + base, n, w, h, device=dev, fmt=fmt)
        frames = fr.cpu().numpy()
        if fmt == "L16":
            frames = frames.view(np.uint16)
        rc_h, out_h, cnt_h, st_h = det_h.detect_batch_raw(frames, n_threads=0, cap=cap)
        rc_d, out_d, cnt_d, st_d = det_d.detect_batch_raw(frames, n_threads=0, cap=cap)
        assert rc_h == rc_d
        for f in range(n):
            same = st_h[f] == st_d[f] and cnt_h[f] == cnt_d[f] and out_h[f, : cnt_h[f]].tobytes() == out_d[f, : cnt_d[f]].tobytes()
            if not same:
                bad += 1
                print("DIFFERS: %dx%d %s frame %d: %d / %d tags" % (w, h, fmt, first + base + f, cnt_h[f], cnt_d[f]), flush=True)
        total += n
        tags += int(cnt_h.sum())
        back += det_d.get_option("last_device_tail_fallbacks")
        unc += det_d.get_option("last_device_tail_uncertain")
    print("%dx%d %s: %d frames so far, %d differ, %d handed back (%d for an angle in its guard band), %.1f tags per frame, %.0f s"
          % (w, h, fmt, total, bad, back, unc, tags / max(total, 1), time.time() - t0), flush=True)
print("device tail against host tail: %d frames, %d differ, %d handed back to the host tail (%d uncertain, %d capacity)" % (total, bad, back, unc, back - unc))
sys.exit(1 if bad else 0)
