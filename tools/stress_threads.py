"""Host-image entries from several threads at once, one handle per thread (the reference's
`detect(&self)` is callable from many threads): every call's saddle list and tag map must equal the
first result for that image, bit for bit.   usage: python tools/stress_threads.py [threads] [iterations]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
from tests.util import load_image

n_thr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
imgs = [load_image(n) for n in ("EuRoC.png", "iphone.png", "TUM_VI.png", "two_boards.png")]
imgs += [np.asarray(synth.render_frame(i, 1280, 800)[0]) for i in range(3)]
ref_det = A.TagDetector("t36h11")
want = []
for im in imgs:
    s = ref_det.refined_saddle_points(im, as_array=True)
    t = ref_det.detect(im)
    want.append((s.tobytes(), {k: v.tobytes() for k, v in t.items()}))
ref_det.close()
errors = []

def work(tid):
    det = A.TagDetector("t36h11", None, device=0)
    rng = np.random.default_rng(tid)
    try:
        for it in range(iters):
            i = int(rng.integers(0, len(imgs)))
            s = det.refined_saddle_points(imgs[i], as_array=True)
            if s.tobytes() != want[i][0]:
                errors.append((tid, it, i, "saddles"))
            if it % 3 == 0:
                t = det.detect(imgs[i])
                if {k: v.tobytes() for k, v in t.items()} != want[i][1]:
                    errors.append((tid, it, i, "tags"))
    except Exception as e:  # noqa: BLE001
        errors.append((tid, -1, -1, repr(e)))
    det.close()

t0 = time.time()
ths = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
for t in ths: t.start()
for t in ths: t.join()
print("done: %d threads x %d iterations in %.1f s, %d mismatches %s" % (n_thr, iters, time.time() - t0, len(errors), errors[:5]))
sys.exit(1 if errors else 0)
