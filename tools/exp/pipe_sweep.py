"""The in-flight mode by depth and sparse path (100 settle batches, 3 x 100 timed, alternating)."""
import os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth, sharding
dev = torch.device("cuda", 0)
frames, _ = synth.render_batch(0, 256, 1280, 800, device=dev)
cfgs = [(d, p) for d in (2, 3, 4) for p in (1, 3)]
res = {c: [] for c in cfgs}
for rnd in range(2):
    for c in (cfgs if rnd == 0 else cfgs[::-1]):
        d, p = c
        os.environ["AGX_SPARSE_PATH"] = str(p)
        pipe = sharding.ChainPipeline(A.TagFamily.T36H11, 256, dev, depth=d)
        for _ in range(100): pipe.submit(frames)
        pipe.finish(); torch.cuda.synchronize(dev)
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(100): pipe.submit(frames)
            pipe.finish(); torch.cuda.synchronize(dev)
            res[c].append((time.perf_counter() - t0) / 100 * 1e3)
        pipe.close()
for c in cfgs: print("depth %d path %d: median %.4f ms" % (c[0], c[1], statistics.median(res[c])), ["%.4f" % x for x in res[c]])
