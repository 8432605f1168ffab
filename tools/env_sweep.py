import os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
frames, _ = synth.render_batch(0, 256, 1280, 800, device="cuda")
det = A.TagDetector("t36h11")
def run(env):
    for k in ("AGX_G_FLOOD", "AGX_G_VERIFY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(5): det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
    det.profile_enable(True); det.profile_reset()
    for _ in range(10): det.saddles_batch_enqueue(frames)
    det.sync(); p = det.profile_read(); det.profile_enable(False)
    return wall, {k: v[0] / v[1] for k, v in p.items()}
configs = [{}, {"AGX_G_FLOOD": "48"}, {"AGX_G_VERIFY": "120"}, {"AGX_G_VERIFY": "140"}]
res = {i: [] for i in range(len(configs))}
for rnd in range(6):
    order = range(len(configs)) if rnd % 2 == 0 else reversed(range(len(configs)))
    for i in order:
        res[i].append(run(configs[i]))
for i, c in enumerate(configs):
    w = statistics.median(x[0] for x in res[i])
    ks = {k: statistics.median(x[1][k] for x in res[i]) for k in res[i][0][1]}
    print(c, "wall %.4f" % w, {k: round(v, 4) for k, v in ks.items()})
