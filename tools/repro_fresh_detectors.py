"""Fresh detectors on recycled device memory: create -> first batch -> fetch -> destroy, many times in
one process, with garbage-filled allocations freed in between (hipMalloc hands recycled blocks back
without clearing them).  Every result is compared with the first one of its configuration.
usage: python tools/repro_fresh_detectors.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(5)
configs = [(3, 640, 400), (2, 640, 400), (1, 1280, 800), (5, 320, 240), (4, 1284, 96), (2, 1920, 1080)]
frames = {c: synth.render_batch(70 + i, c[0], c[1], c[2], device="cuda")[0] for i, c in enumerate(configs)}
torch.cuda.synchronize()
first = {}
t0 = time.time()
for it in range(rounds):
    # garbage that goes back to the driver: all-ones words (NaN as f32, huge as u32)
    junk = [torch.full((int(rng.integers(1, 64)) << 18,), -1, dtype=torch.int32, device="cuda") for _ in range(int(rng.integers(1, 4)))]
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()
    c = configs[int(rng.integers(0, len(configs)))]
    kind = int(rng.integers(0, 3))
    if kind == 0:
        grp = A.DetectorGroup("t36h11", [0], transport="rccl")
        grp.saddles_enqueue([frames[c]])
        res, status = grp.saddles_fetch()
        grp.close()
    elif kind == 1:
        grp = A.DetectorGroup("t36h11", [0, 0], transport="peer")
        grp.saddles_enqueue([frames[c], frames[c]])
        res, status = grp.saddles_fetch()
        res, status = res[:c[0]], status[:c[0]]
        grp.close()
    else:
        det = A.TagDetector("t36h11", None, device=0)
        det.saddles_batch_enqueue(frames[c])
        res, status = det.saddles_batch_fetch()
        det.close()
    assert (np.asarray(status) == 0).all(), (it, c, kind, status)
    sig = [r.tobytes() for r in res]
    if c not in first:
        first[c] = sig
    assert sig == first[c], "round %d config %s kind %d: result differs from the first run" % (it, c, kind)
    if it % 10 == 9:
        print("round", it + 1, "of", rounds, "%.0f s" % (time.time() - t0), flush=True)
print("done:", rounds, "rounds, all results equal")
