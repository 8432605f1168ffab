"""Where the floor of the step lies: ONE process, ONE box, the bench's 256-frame batch, everything
alternating over several rounds --

  serial step as shipped; three batches in flight (sharding.ChainPipeline, as bench.py's "pipelined");
      (round 5's one structural experiment -- the sparse kernels of a batch on a second, high-priority stream -- was measured
      with this tool and rejected: profiles/rejected/r5_sparse_priority_stream_experiment.patch, profiles/r5_step_floor_box_c.txt)
  K1 as shipped / without its blur stores (ablation 1: the arithmetic + the input reads + the mask) / blur only in
      registers (ablation 5) / without Hessian, mask and minimum (ablation 4: the blur + its stores);
  the traffic alone: a plain device copy moving K1's algorithmic byte count (1.3435 GB), scaled to the 1.450 GB the PMC
      counters see K1 move; a fill of the blur plane's 1.0486 GB (stores only);
  the sparse kernels alone: k_verify_seeds, k_sparse_frame by events around each (K1's products already in HBM).

Floor of THIS design with perfect overlap of its two limits = max(K1 arithmetic without a store, K1's traffic alone) +
the sparse kernels (they need whole CUs: they do not hide under K1 -- profiles/r4_box3_pipeline3_overlap.txt).

    python tools/step_floor.py > profiles/r5_step_floor.txt"""
import os
import statistics
import sys
import time

sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import sharding, synth

ROUNDS = int(os.environ.get("ROUNDS", "5"))
dev = torch.device("cuda", 0)
F_, W, H = 256, 1280, 800
frames, _ = synth.render_batch(0, F_, W, H, device=dev)
det = A.TagDetector("t36h11")
plane = torch.empty((F_, H, W), dtype=torch.float32, device=dev)
plane2 = torch.empty_like(plane)


def k1(dbg):
    det.set_option("debug_ablation", dbg)
    for _ in range(20):  # (right after other detectors' workspaces were freed the first batches run slower)
        det.saddles_batch_enqueue(frames)
    det.sync()
    det.set_option("profile_kernel", 0)
    det.set_option("profile_stride", 1)
    det.profile_enable(True)
    det.profile_reset()
    for _ in range(12):
        det.saddles_batch_enqueue(frames)
    det.sync()
    p = det.profile_read()
    det.profile_enable(False)
    det.set_option("debug_ablation", 0)
    return p["k_blur_hessian"][0] / p["k_blur_hessian"][1]


def kernel_alone(idx, name):
    det.set_option("profile_kernel", idx)
    det.set_option("profile_stride", 3)
    det.profile_enable(True)
    det.profile_reset()
    for _ in range(24):
        det.saddles_batch_enqueue(frames)
    det.sync()
    p = det.profile_read()
    det.profile_enable(False)
    det.set_option("profile_kernel", 0)
    det.set_option("profile_stride", 1)
    return p[name][0] / max(p[name][1], 1)


def serial():
    for _ in range(5):
        det.saddles_batch_enqueue(frames)
    det.sync()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(30):
        det.saddles_batch_enqueue(frames)
    det.sync()
    return (time.perf_counter() - t0) / 30 * 1e3


def in_flight():
    pipe = sharding.ChainPipeline(A.TagFamily.T36H11, F_, dev, depth=3)
    for _ in range(40):  # (the first ~30 batches after the detectors are created run slower: tools/exp/pipe_paths.py)
        pipe.submit(frames)
    pipe.finish()
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            pipe.submit(frames)
        pipe.finish()
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    pipe.close()
    return statistics.median(ts)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / n


names = list(det.profile_read().keys())
legs = {
    "serial step, as shipped": serial,
    "three batches in flight": in_flight,
    "K1 as shipped": lambda: k1(0),
    "K1 without its blur stores (ablation 1)": lambda: k1(1),
    "K1 without Hessian / mask / minimum (ablation 4)": lambda: k1(4),
    "K1 blur only, nothing stored (ablation 5)": lambda: k1(5),
    "plain device copy of 0.6718 GB (= 1.3435 GB moved)": lambda: timed(lambda: plane2.view(-1)[: 167936000].copy_(plane.view(-1)[: 167936000])),
    "fill of the blur plane (1.0486 GB of stores)": lambda: timed(lambda: plane.zero_()),
    "k_verify_seeds alone": lambda: kernel_alone(names.index("k_verify_seeds"), "k_verify_seeds"),
    "k_sparse_frame alone": lambda: kernel_alone(names.index("k_sparse_frame"), "k_sparse_frame"),
}
print("# kernels of the library:", names)
res = {k: [] for k in legs}
for rnd in range(ROUNDS):
    order = list(legs) if rnd % 2 == 0 else list(legs)[::-1]
    for k in order:
        res[k].append(legs[k]())
    print("# round %d done" % rnd, flush=True)
med = {k: statistics.median(v) for k, v in res.items()}
print("# tools/step_floor.py: %d frames %dx%d L8, %d alternating rounds in one process; ms (median; all rounds)" % (F_, W, H, ROUNDS))
for k, v in res.items():
    print("%-70s %.4f   %s" % (k, med[k], " ".join("%.4f" % x for x in v)))
valu = med["K1 without its blur stores (ablation 1)"]
copy = med["plain device copy of 0.6718 GB (= 1.3435 GB moved)"]
traffic = copy * 1.450 / 1.3435  # K1 moves 1.450 GB by the PMC counters (profiles/r4_box3_pmc_traffic.txt), at this box's copy rate
sparse = med["k_verify_seeds alone"] + med["k_sparse_frame alone"]
floor = max(valu, traffic) + sparse
print()
print("this box's copy rate: %.0f GB/s; K1's 1.450 GB at that rate: %.4f ms" % (1343.5 / copy, traffic))
print("K1's two limits: arithmetic without a store %.4f ms, its traffic alone %.4f ms; shipped %.4f (%.1f %% above the larger)"
      % (valu, traffic, med["K1 as shipped"], 100 * (med["K1 as shipped"] / max(valu, traffic) - 1)))
print("sparse kernels alone: %.4f ms" % sparse)
print("floor of this design = the larger of K1's limits (perfect overlap inside K1) + the sparse kernels (they take whole CUs and do not")
print("hide under the next batch's K1: 3 %% overlap measured, profiles/r4_box3_pipeline3_overlap.txt): %.4f ms" % floor)
print("shipped: serial %.4f ms, three in flight %.4f ms; chain_frac >= 0.50 needs 0.3523 ms" % (med["serial step, as shipped"], med["three batches in flight"]))
