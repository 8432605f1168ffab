"""Randomised GPU-vs-oracle parity sweep (sizes, formats incl. f32 planes, batch sizes, forced K1
segment heights, aligned and unaligned widths through the device-batch API); not part of the default
suite.  usage: python tools/stress_parity.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
from oracle import oracle as O
O.lib()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
det = A.TagDetector("t36h11")
bad = 0
t0 = time.time()
for case in range(n_cases):
    fmt = ["L8", "L16", "RGB8", "LF32"][int(rng.integers(0, 4))]
    w = int(rng.choice([rng.integers(8, 80) * 4, rng.integers(8, 80) * 4, rng.integers(200, 420) * 4, rng.integers(33, 900)]))
    h = int(rng.choice([rng.integers(9, 70), rng.integers(64, 400), rng.integers(400, 1100)]))
    n = int(rng.integers(1, 7))
    kind = int(rng.integers(0, 3))
    rows = int(rng.choice([0, 0, 32, 64, 96, 128]))
    src_fmt = "L8" if fmt == "LF32" else fmt
    fr, _ = synth.render_batch(int(rng.integers(0, 1000)), n, (w + 3) // 4 * 4, h, device="cuda", fmt=src_fmt, pure_noise=(kind == 0))
    frames = fr[:, :, :w].contiguous()
    if fmt == "LF32":
        frames = (frames.to(torch.float32) / 255.0).contiguous()
    host = frames.cpu().numpy()
    if fmt == "L16": host = host.view(np.uint16)
    det.set_option("k1_rows_per_segment", rows)
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch(raise_on_overflow=False)
    for i in range(n):
        ref = O.refined_saddle_points(host[i], cap=1 << 18)
        got = res[i]
        ok = status[i] == 0 and len(got) == len(ref) and all(np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)) for f in ("x", "y", "k"))
        ok = ok and (len(ref) == 0 or (np.max(np.abs(got["theta"] - ref["theta"])) <= 1e-3 and np.max(np.abs(got["phi"] - ref["phi"])) <= 1e-3))
        if not ok:
            bad += 1
            print("MISMATCH case", case, fmt, w, h, "rows", rows, "frame", i, "of", n, "status", status[i], "gpu", len(got), "oracle", len(ref), flush=True)
    if case % 20 == 19: print("case", case + 1, "of", n_cases, "%.0f s" % (time.time() - t0), "mismatches", bad, flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
