"""Three batches in flight (sharding.ChainPipeline) with the sparse path forced: 1 = three launches, 3 = k_verify_seeds + k_sparse_frame."""
import os, sys, time, statistics
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth, sharding
dev = torch.device("cuda", 0)
frames, _ = synth.render_batch(0, 256, 1280, 800, device=dev)
res = {1: [], 3: []}
for rnd in range(5):
    for p in ((1, 3) if rnd % 2 == 0 else (3, 1)):
        os.environ["AGX_SPARSE_PATH"] = str(p)
        pipe = sharding.ChainPipeline(A.TagFamily.T36H11, 256, dev, depth=3)
        for _ in range(12): pipe.submit(frames)
        pipe.finish(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20): pipe.submit(frames)
        pipe.finish(); torch.cuda.synchronize(dev)
        res[p].append((time.perf_counter() - t0) / 20 * 1e3)
        pipe.close()
for p in res: print("path %d, three batches in flight: median %.4f ms" % (p, statistics.median(res[p])), ["%.4f" % x for x in res[p]])
