"""Latency of the host-image entry (agx_refined_saddle_points: PCIe upload -> chain -> lists on the
host) for single frames, and what the five kernels take inside it (hipEvents around every launch in
a second pass)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
from tests.util import load_image

det = A.TagDetector("t36h11")
cases = [("EuRoC.png 752x480 L8", load_image("EuRoC.png")),
         ("synthetic 1280x800 L8", np.asarray(synth.render_frame(0, 1280, 800)[0])),
         ("synthetic 1920x1080 L8", np.asarray(synth.render_frame(1, 1920, 1080)[0])),
         ("iphone.png", load_image("iphone.png"))]
for name, img in cases:
    for _ in range(5):
        det.refined_saddle_points(img, as_array=True)
    ts = []
    for _ in range(60):
        t0 = time.perf_counter()
        s = det.refined_saddle_points(img, as_array=True)
        ts.append(time.perf_counter() - t0)
    det.profile_enable(2); det.profile_reset()
    for _ in range(20):
        det.refined_saddle_points(img, as_array=True)
    prof = det.profile_read(); det.profile_enable(0)
    k = {n: 1e3 * v[0] / max(v[1], 1) for n, v in prof.items()}
    print("%-26s %s %d saddles: median %.1f us, min %.1f us | kernels (us): %s = %.1f" % (
        name, img.shape, len(s), 1e6 * statistics.median(ts), 1e6 * min(ts),
        " ".join("%s %.1f" % (n.replace("k_", ""), v) for n, v in k.items()), sum(k.values())), flush=True)
