"""Randomised GPU-vs-oracle parity sweep (sizes, formats, batch sizes); not part of the default suite.
usage: python tools/stress_parity.py [n_cases] [seed]"""
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
    fmt = ["L8", "L16", "RGB8"][int(rng.integers(0, 3))]
    w = int(rng.choice([rng.integers(8, 80) * 4, rng.integers(8, 80) * 4, rng.integers(200, 420) * 4, rng.integers(33, 900)]))
    h = int(rng.choice([rng.integers(9, 70), rng.integers(64, 400), rng.integers(400, 900)]))
    n = int(rng.integers(1, 5))
    kind = int(rng.integers(0, 3))
    if w % 4 == 0:
        frames, _ = synth.render_batch(int(rng.integers(0, 1000)), n, w, h, device="cuda", fmt=fmt, pure_noise=(kind == 0))
        host = frames.cpu().numpy()
        if fmt == "L16": host = host.view(np.uint16)
        det.saddles_batch_enqueue(frames)
        res, status = det.saddles_batch_fetch(raise_on_overflow=False)
    else:  # unaligned widths only through the host API (rows must be 4-byte aligned on the device)
        n = 1
        fr, _ = synth.render_batch(int(rng.integers(0, 1000)), 1, (w + 3) // 4 * 4, h, device="cpu", fmt=fmt, pure_noise=(kind == 0))
        host = fr.numpy()
        if fmt == "L16": host = host.view(np.uint16)
        host = np.ascontiguousarray(host[:, :, :w])
        res, status = [det.refined_saddle_points(host[0], as_array=True)], np.zeros(1, np.int32)
    for i in range(n):
        ref = O.refined_saddle_points(host[i])
        got = res[i]
        ok = status[i] == 0 and len(got) == len(ref) and all(np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)) for f in ("x", "y", "k"))
        ok = ok and (len(ref) == 0 or (np.max(np.abs(got["theta"] - ref["theta"])) <= 1e-3 and np.max(np.abs(got["phi"] - ref["phi"])) <= 1e-3))
        if not ok and status[i] == -3 and len(ref) > 16384:
            print("capacity (by design): case", case, fmt, w, h, "frame", i, "oracle", len(ref), "saddles > limit 16384", flush=True)
            continue
        if not ok:
            bad += 1
            print("MISMATCH case", case, fmt, w, h, "frame", i, "status", status[i], "gpu", len(got), "oracle", len(ref), flush=True)
    if case % 10 == 9: print("case", case + 1, "of", n_cases, "%.0f s" % (time.time() - t0), "mismatches", bad, flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
