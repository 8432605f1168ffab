"""Per-kernel times of the chain for a list of tuning environments, alternating in ONE process (6 rounds, the order
reversed every other round).  usage: python tools/env_sweep.py ['{"AGX_G_FLOOD": "20"}' '{}' ...]   (each argument one
environment as JSON; default: today's defaults against the settings they replaced; env WIDTH, HEIGHT, FRAMES, UNIQUE, FORMAT, NOISE)"""
import json, os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
W_, H_, F_ = int(os.environ.get("WIDTH", "1280")), int(os.environ.get("HEIGHT", "800")), int(os.environ.get("FRAMES", "256"))
U_ = int(os.environ.get("UNIQUE", str(F_)))
base, _ = synth.render_batch(0, U_, W_, H_, device="cuda", fmt=os.environ.get("FORMAT", "L8"), pure_noise=os.environ.get("NOISE", "0") == "1")
frames = base.repeat((F_ // U_ + 1,) + (1,) * (base.dim() - 1))[:F_].contiguous()
det = A.TagDetector("t36h11")
def run(env):
    for k in [k for k in os.environ if k.startswith(("AGX_G_", "AGX_K1_", "AGX_SP_"))]:
        os.environ.pop(k, None)
    os.environ.update(env)
    det.set_option("reload_tuning_env", 1)  # (the library reads each override once per process and keeps it)
    for _ in range(5): det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
    det.profile_enable(True); det.profile_reset()
    for _ in range(10): det.saddles_batch_enqueue(frames)
    det.sync(); p = det.profile_read(); det.profile_enable(False)
    return wall, {k: v[0] / max(v[1], 1) for k, v in p.items()}
configs = [json.loads(x) for x in sys.argv[1:]] or [{}, {"AGX_G_FLOOD": "48"}, {"AGX_K1_STRIP_COLS": "216"}, {"AGX_K1_ROWS": "128"}]
res = {i: [] for i in range(len(configs))}
for rnd in range(6):
    order = range(len(configs)) if rnd % 2 == 0 else reversed(range(len(configs)))
    for i in order:
        res[i].append(run(configs[i]))
for i, c in enumerate(configs):
    w = statistics.median(x[0] for x in res[i])
    ks = {k: statistics.median(x[1][k] for x in res[i]) for k in res[i][0][1]}
    print(c, "wall %.4f" % w, {k: round(v, 4) for k, v in ks.items()})
