"""Can the chain be captured into a HIP graph?  Two consecutive batches are captured with
torch.cuda.CUDAGraph on a side stream and replayed; then ONE batch is captured on its own and replayed
three times (every captured batch clears its own counter set, so any number of batches per graph
replays correctly); results must equal the eager ones, and the replay time per batch is printed next to
the eager time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth, sharding

dev = torch.device("cuda", 0)
F = int(os.environ.get("FRAMES", "256"))
frames, _ = synth.render_batch(0, F, 1280, 800, device=dev)
det = A.TagDetector("t36h11", None, device=0)
bufs = [sharding.alloc_result_buffers(F, dev) for _ in range(2)]
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(4):  # workspace, both counter sets, stream binding -- all before the capture
        det.saddles_batch_enqueue_to(frames, *bufs[0])
        det.saddles_batch_enqueue_to(frames, *bufs[1])
s.synchronize()
eager = [(b[0].cpu().numpy().copy(), b[1].cpu().numpy().copy()) for b in bufs]
for b in bufs:
    b[0].zero_(); b[1].zero_()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    det.saddles_batch_enqueue_to(frames, *bufs[0])
    det.saddles_batch_enqueue_to(frames, *bufs[1])
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
ok = True
for i, b in enumerate(bufs):
    t = b[1].cpu().numpy(); sd = b[0].cpu().numpy()
    same = np.array_equal(t[:, [0, 2, 3]], eager[i][1][:, [0, 2, 3]])
    if same:
        for f in range(F):
            a = sd[t[f, 1]: t[f, 1] + t[f, 0]]; e = eager[i][0][eager[i][1][f, 1]: eager[i][1][f, 1] + eager[i][1][f, 0]]
            if a.tobytes() != e.tobytes(): same = False; break
    print("batch", i, "graph replay equals eager:", same, "saddles", int(t[:, 0].sum()), flush=True)
    ok = ok and same
# one batch per graph, replayed three times, then an eager batch behind it
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s):
    det.saddles_batch_enqueue_to(frames, *bufs[0])
for rep in range(3):
    bufs[0][0].zero_(); bufs[0][1].zero_()
    torch.cuda.synchronize()
    g1.replay(); torch.cuda.synchronize()
    t = bufs[0][1].cpu().numpy(); sd = bufs[0][0].cpu().numpy()
    same = np.array_equal(t[:, [0, 2, 3]], eager[0][1][:, [0, 2, 3]])
    if same:
        for f in range(F):
            a = sd[t[f, 1]: t[f, 1] + t[f, 0]]; e = eager[0][0][eager[0][1][f, 1]: eager[0][1][f, 1] + eager[0][1][f, 0]]
            if a.tobytes() != e.tobytes(): same = False; break
    print("single-batch graph, replay", rep, "equals eager:", same, flush=True)
    ok = ok and same
with torch.cuda.stream(s):
    det.saddles_batch_enqueue_to(frames, *bufs[1])
s.synchronize()
t = bufs[1][1].cpu().numpy()
same = np.array_equal(t[:, [0, 2, 3]], eager[1][1][:, [0, 2, 3]])
print("eager batch after the graphs equals eager:", same, flush=True)
ok = ok and same
for name, fn in (("eager", lambda: (det.saddles_batch_enqueue_to(frames, *bufs[0]), det.saddles_batch_enqueue_to(frames, *bufs[1]))),
                 ("graph", g.replay)):
    with torch.cuda.stream(s):
        for _ in range(5): fn()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(25): fn()
        s.synchronize()
        print("%s: %.4f ms per batch" % (name, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
sys.exit(0 if ok else 1)
