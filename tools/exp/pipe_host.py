"""Where the in-flight mode's time goes: host time of the submit loop against the GPU time of the same batches."""
import os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth, sharding
dev = torch.device("cuda", 0)
frames, _ = synth.render_batch(0, 256, 1280, 800, device=dev)
for depth in (1, 2, 3):
    pipe = sharding.ChainPipeline(A.TagFamily.T36H11, 256, dev, depth=depth)
    for _ in range(12): pipe.submit(frames)
    pipe.finish(); torch.cuda.synchronize(dev)
    for steps in (20, 200):
        host, tot = [], []
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps): pipe.submit(frames)
            t1 = time.perf_counter()
            pipe.finish(); torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            host.append((t1 - t0) / steps * 1e3); tot.append((t2 - t0) / steps * 1e3)
        print("depth %d steps %3d: submit loop %.4f ms/step on the host, whole %.4f ms/step" % (depth, steps, statistics.median(host), statistics.median(tot)))
    pipe.close()
