import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
from oracle import oracle as O
from concurrent.futures import ThreadPoolExecutor
O.lib()
det = A.TagDetector("t36h11")
n = 768
fr, _ = synth.render_batch(5000, n, 640, 400, device="cuda")
host = fr.cpu().numpy()
with ThreadPoolExecutor(16) as ex:
    refs = list(ex.map(O.detect, host))
bad = 0
for thr in (1, 3, 16, 32):
    for rep in range(2):
        t0 = time.time()
        got = det.detect_batch(host, n_threads=thr)
        dt = time.time() - t0
        for i in range(n):
            if sorted(got[i]) != sorted(refs[i]) or any(got[i][k].tobytes() != refs[i][k].tobytes() for k in refs[i]):
                bad += 1
        print("threads", thr, "rep", rep, "%.0f frames/s" % (n / dt), "mismatching frames so far", bad, flush=True)
print("done", bad)
sys.exit(1 if bad else 0)
