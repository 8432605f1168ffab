"""Several batches in flight on one GPU for a long time: three detectors on three HIP streams take
batches of different content, size and format in turn (the dense kernel of one overlaps the sparse
kernels of another); every result must equal, bit for bit, the result of the same batch computed
alone.  Catches rare ordering bugs (hand-over of records between workgroups, counter sets, stream
order) that a single pass cannot.   usage: python tools/stress_concurrency.py [rounds] [detectors] [big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth, sharding

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_dets = int(sys.argv[2]) if len(sys.argv) > 2 else 3  # 1: no concurrency (control)
dev = torch.device("cuda", 0)
specs = [(24, 640, 400, "L8", False), (8, 1280, 800, "L8", False), (6, 1280, 800, "RGB8", False), (12, 644, 402, "L16", False),
         (4, 640, 400, "L8", True), (3, 1920, 1080, "L8", False), (40, 320, 240, "L8", False)]
if len(sys.argv) > 3 and sys.argv[3] == "big":  # the benchmarked geometry: many rounds of workgroups per launch
    specs = [(256, 1280, 800, "L8", False), (64, 1280, 800, "RGB8", False), (32, 3840, 2160, "L8", False)]
batches = []
ref = A.TagDetector("t36h11", None, device=0)
for i, (n, w, h, fmt, noise) in enumerate(specs):
    fr, _ = synth.render_batch(100 * i, n, w, h, device=dev, fmt=fmt, pure_noise=noise)
    slab = 8192 if noise else 1024
    out_s, out_t = sharding.alloc_result_buffers(n, dev, slab)
    ref.saddles_batch_enqueue_to(fr, out_s, out_t)
    ref.sync()
    t = out_t.cpu().numpy().copy()
    assert (t[:, 2] & 7 == 0).all(), (specs[i], t[:4])
    tot = int((t[:, 0]).sum())
    batches.append((fr, slab, out_s.cpu().numpy()[:tot].copy(), t))
print("reference results:", [int(b[3][:, 0].sum()) for b in batches], flush=True)
dets = [A.TagDetector("t36h11", None, device=0) for _ in range(3)]
streams = [torch.cuda.Stream(dev) for _ in range(3)]
rng = np.random.default_rng(1)
bad = 0
t0 = time.time()
for r in range(rounds):
    picks = [int(rng.integers(0, len(batches))) for _ in range(3)]
    outs = []
    doubled = [False] * 3
    for j, p in enumerate(picks):
        if j >= n_dets:
            break
        fr, slab, _, _ = batches[p]
        with torch.cuda.stream(streams[j]):
            # (allocated -- and zero-filled -- on the stream that uses them: a fill on the default stream
            # would not be ordered before the chain on this non-blocking stream)
            out_s, out_t = sharding.alloc_result_buffers(fr.shape[0], dev, slab)
            dets[j].saddles_batch_enqueue_to(fr, out_s, out_t)
            if rng.integers(0, 2):  # sometimes a second batch right behind it on the same detector (counter sets alternate)
                dets[j].saddles_batch_enqueue_to(fr, out_s, out_t)
                doubled[j] = True
        outs.append((out_s, out_t))
    torch.cuda.synchronize(dev)
    for j, p in enumerate(picks[:n_dets]):
        _, _, exp_s, exp_t = batches[p]
        t = outs[j][1].cpu().numpy()
        # offsets depend on the order in which the frames' tails claim their slice: compare frame by frame
        ok = np.array_equal(t[:, [0, 2, 3]], exp_t[:, [0, 2, 3]])
        if ok:
            s = outs[j][0].cpu().numpy()
            for f in range(len(t)):
                a = s[t[f, 1]: t[f, 1] + t[f, 0]]
                b = exp_s[exp_t[f, 1]: exp_t[f, 1] + exp_t[f, 0]]
                if a.tobytes() != b.tobytes():
                    ok = False
                    break
        if not ok:
            bad += 1
            s = outs[j][0].cpu().numpy()
            why = []
            for f in range(len(t)):
                if not np.array_equal(t[f, [0, 2, 3]], exp_t[f, [0, 2, 3]]):
                    why.append("frame %d table %s expected %s" % (f, t[f].tolist(), exp_t[f].tolist()))
                    continue
                a = s[t[f, 1]: t[f, 1] + t[f, 0]]
                b = exp_s[exp_t[f, 1]: exp_t[f, 1] + exp_t[f, 0]]
                d = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))[0]
                if len(d):
                    why.append("frame %d: %d of %d records differ, first %d: %s vs %s" % (f, len(d), len(a), d[0], a[d[0]].tolist(), b[d[0]].tolist()))
            print("MISMATCH round", r, "detector", j, "batch", specs[p], "doubled" if doubled[j] else "single", "|", "; ".join(why[:3]), flush=True)
    if r % 50 == 49:
        print("round", r + 1, "of", rounds, "%.0f s" % (time.time() - t0), "mismatches", bad, flush=True)
print("done:", rounds, "rounds,", bad, "mismatches")
sys.exit(1 if bad else 0)
