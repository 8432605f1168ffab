"""The serial chain eager and as a replayed HIP graph (one batch per graph), alternating, ms per batch."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth, sharding
dev = torch.device("cuda", 0)
F = int(os.environ.get("FRAMES", "256"))
frames, _ = synth.render_batch(0, F, 1280, 800, device=dev)
det = A.TagDetector("t36h11", None, device=0)
out, table = sharding.alloc_result_buffers(F, dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(6): det.saddles_batch_enqueue_to(frames, out, table)
s.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    det.saddles_batch_enqueue_to(frames, out, table)
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4, stream=s):
    for _ in range(4): det.saddles_batch_enqueue_to(frames, out, table)
def run(fn, n, per):
    with torch.cuda.stream(s):
        for _ in range(5): fn()
        s.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        s.synchronize()
        return (time.perf_counter() - t0) / (n * per) * 1e3
res = {"eager": [], "graph of 1 batch": [], "graph of 4 batches": []}
for r in range(6):
    res["eager"].append(run(lambda: det.saddles_batch_enqueue_to(frames, out, table), 100, 1))
    res["graph of 1 batch"].append(run(g.replay, 100, 1))
    res["graph of 4 batches"].append(run(g4.replay, 25, 4))
for k, v in res.items():
    print("%-20s median %.4f ms per batch  (%s)" % (k, statistics.median(v), " ".join("%.4f" % x for x in v)))
