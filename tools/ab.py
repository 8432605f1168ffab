"""A/B of library builds on ONE box: every build runs the same workload in its own process, the builds
take turns (ROUNDS rounds, the order reversed every other round) and the per-kernel medians over all rounds are printed -- single short runs
differ by a few percent from one to the next on the same box.

usage: python tools/ab.py libA.so libB.so ...        (env ROUNDS=3, FRAMES, UNIQUE, WIDTH, HEIGHT, FORMAT, NOISE)"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
F = int(os.environ.get("FRAMES", "256")); U = int(os.environ.get("UNIQUE", "256"))
W, H = int(os.environ.get("WIDTH", "1280")), int(os.environ.get("HEIGHT", "800"))
base, _ = synth.render_batch(0, U, W, H, device="cuda", fmt=os.environ.get("FORMAT", "L8"), pure_noise=os.environ.get("NOISE", "0") == "1")
frames = base.repeat((F // U + 1,) + (1,) * (base.dim() - 1))[:F].contiguous()
det = A.TagDetector("t36h11")
for _ in range(20):
    det.saddles_batch_enqueue(frames)
det.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40):
    det.saddles_batch_enqueue(frames)
det.sync(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 40 * 1e3
det.profile_enable(True); det.profile_reset()
for _ in range(40):
    det.saddles_batch_enqueue(frames)
det.sync()
p = det.profile_read()
out = {k: v[0] / max(v[1], 1) for k, v in p.items()}
out["wall"] = wall
print("AB " + json.dumps(out))
''' % ROOT
libs = sys.argv[1:]
rounds = int(os.environ.get("ROUNDS", "3"))
res = {l: [] for l in libs}
for r in range(rounds):
    # the order is reversed every other round: a process that follows another one finds the GPU warm (the second of
    # two identical builds measured 2.5 % faster in every kernel when it always ran second)
    for l in (libs if r % 2 == 0 else libs[::-1]):
        env = dict(os.environ, AGX_LIBRARY=os.path.abspath(l))
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [x for x in o.stdout.splitlines() if x.startswith("AB ")]
        if not line:
            print(l, "FAILED", o.stderr[-500:]); continue
        res[l].append(json.loads(line[0][3:]))
for l in libs:
    if not res[l]: continue
    keys = list(res[l][0].keys())
    med = {k: statistics.median(x[k] for x in res[l]) for k in keys}
    sparse = sum(v for k, v in med.items() if k not in ("wall", "k_blur_hessian"))
    print("%-44s" % os.path.basename(l), " ".join("%s %.4f" % (k.replace("k_", "")[:12], med[k]) for k in keys), "| sparse %.4f" % sparse, "| K1 runs:", " ".join("%.4f" % x["k_blur_hessian"] for x in res[l]), flush=True)
