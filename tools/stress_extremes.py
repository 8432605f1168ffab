"""Extreme geometries against the oracle: very wide / very tall frames, an 8K frame, thousands of small
frames in one batch, all formats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
from oracle import oracle as O
from concurrent.futures import ThreadPoolExecutor
O.lib()
det = A.TagDetector("t36h11")
cases = [(2, 8192, 64, "L8", False), (1, 16380, 40, "L8", False), (2, 64, 8192, "L8", False), (1, 40, 20000, "L8", True),
         (1, 7680, 4320, "L8", False), (1, 7680, 4320, "RGB8", False), (2048, 320, 240, "L8", False), (3, 5000, 3000, "L16", False),
         (1, 2, 30000, "L8", True), (1, 30000, 2, "L8", True), (1, 32764, 33, "L8", True), (700, 100, 100, "RGB8", False)]
bad = 0
for (n, w, h, fmt, noise) in cases:
    t0 = time.time()
    fr, _ = synth.render_batch(17, n, (w + 3) // 4 * 4, h, device="cuda", fmt=fmt, pure_noise=noise)
    frames = fr[:, :, :w].contiguous()
    host = frames.cpu().numpy()
    if fmt == "L16": host = host.view(np.uint16)
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch(raise_on_overflow=False)
    idx = list(range(n)) if n <= 8 else list(np.random.default_rng(1).choice(n, 24, replace=False))
    with ThreadPoolExecutor(8) as ex:
        refs = list(ex.map(lambda i: O.refined_saddle_points(host[i], cap=1 << 19), idx))
    ok = True
    for i, ref in zip(idx, refs):
        got = res[i]
        good = status[i] == 0 and len(got) == len(ref) and all(np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)) for f in ("x", "y", "k"))
        good = good and (len(ref) == 0 or (np.max(np.abs(got["theta"] - ref["theta"])) <= 1e-3 and np.max(np.abs(got["phi"] - ref["phi"])) <= 1e-3))
        if not good:
            ok = False
            print("  MISMATCH frame", i, "status", status[i], "gpu", len(got), "oracle", len(ref), flush=True)
    bad += 0 if ok else 1
    print("%4d x %5d x %5d %-4s %s: %s (%d frames checked, %.0f saddles/frame, %.1f s)" % (
        n, w, h, fmt, "noise" if noise else "board", "ok" if ok else "MISMATCH", len(idx), np.mean([len(r) for r in res]), time.time() - t0), flush=True)
    del fr, frames
    torch.cuda.empty_cache()
print("done:", len(cases), "cases,", bad, "bad")
sys.exit(1 if bad else 0)
