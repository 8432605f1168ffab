import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
from tests.util import load_image
det = A.TagDetector("t36h11")
for name in ("iphone.png", "two_boards.png", "EuRoC.png", "TUM_VI.png", 0, 7):
    img = load_image(name) if isinstance(name, str) else np.asarray(synth.render_frame(name, 1280, 800)[0])
    det.set_option("tail_threads", 1)
    ref = det.detect(img)
    line = []
    for thr in (1, 2, 4, 8, 16):
        det.set_option("tail_threads", thr)
        for _ in range(3): det.detect(img)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); got = det.detect(img); ts.append(time.perf_counter() - t0)
        same = list(got) == list(ref) and all(got[k].tobytes() == ref[k].tobytes() for k in ref)
        line.append("%d: %.2f ms%s" % (thr, 1e3 * sorted(ts)[len(ts)//2], "" if same else " DIFFERENT"))
    print(name, " | ".join(line), flush=True)
