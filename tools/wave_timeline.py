"""When do the waves of the sparse kernels start and end?  (debug_ablation & 4096: every workgroup of
k_verify_seeds / k_flood_refine / k_rare records s_memrealtime at its start and end.)  Prints, per kernel: the
span from the first start to the last end, how long the dispatch of all workgroups took, the distribution of
wave lifetimes, the resident-wave count over time and the longest waves with their (slot, frame).

usage: python tools/wave_timeline.py            (env FRAMES, UNIQUE, WIDTH, HEIGHT, AGX_LIBRARY as tools/sweep.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth

F = int(os.environ.get("FRAMES", "256"))
U = int(os.environ.get("UNIQUE", "256"))
W, H = int(os.environ.get("WIDTH", "1280")), int(os.environ.get("HEIGHT", "800"))
base, _ = synth.render_batch(0, U, W, H, device="cuda")
frames = base.repeat((F // U + 1,) + (1,) * (base.dim() - 1))[:F].contiguous()
det = A.TagDetector("t36h11")
det.set_option("debug_ablation", 4096)
for _ in range(4):
    det.saddles_batch_enqueue(frames)
det.sync()
TICK_US = 0.01  # s_memrealtime: 100 MHz
grids = {1: int(os.environ.get("AGX_G_VERIFY", "0")) or None, 2: None, 3: None}
bounds = {}
for k, name in ((1, "k_verify_seeds"), (2, "k_flood_refine"), (3, "k_rare_emit")):
    t = det.debug_fetch(k, "wave_times", 1 << 20).astype(np.int64)
    live = t[:, 1] > 0
    n = int(np.nonzero(live)[0].max()) + 1 if live.any() else 0
    t = t[:n]
    t0 = t[:, 0].min()
    bounds[k] = (int(t[:, 0].min()), int(t[:, 1].max()))
    start = (t[:, 0] - t0) * TICK_US
    end = (t[:, 1] - t0) * TICK_US
    dur = end - start
    span = end.max()
    print("%s: %d waves, span %.1f us, last start at %.1f us, lifetimes us: median %.2f  p90 %.2f  p99 %.2f  max %.2f, sum %.0f us (= %.0f resident waves on average)"
          % (name, n, span, start.max(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.sum(), dur.sum() / span))
    hist_edges = [0, 1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 1e9]
    print("   lifetime histogram (us):", ", ".join("<%g: %d" % (b, int(((dur >= a) & (dur < b)).sum())) for a, b in zip(hist_edges[:-1], hist_edges[1:])))
    edges = np.linspace(0, span, 11)
    occ = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
    print("   waves alive in each tenth of the span:", occ)
    started = [int((start < b).sum()) for b in edges[1:]]
    print("   workgroups started by the end of each tenth:", started)
    wpw = {1: 4, 2: 1, 3: 16}[k]  # waves per workgroup (k_verify_seeds: VS_WAVES; k_rare: 1024 threads)

    def ident(i):  # record index -> (slot, frame)
        wg, wv = divmod(int(i), wpw)
        return (wg // F) * wpw + wv, (F - 1 - wg % F) if k != 3 else wg % F
    order = np.argsort(-dur)[:8]
    print("   longest:", ", ".join("slot %d frame %d: %.1f us from %.1f" % (ident(i) + (dur[i], start[i])) for i in order))
    late = np.argsort(-end)[:5]
    print("   last to end:", ", ".join("slot %d frame %d: started %.1f ran %.1f" % (ident(i) + (start[i], dur[i])) for i in late))
# between the launches: the last wave of one kernel ends -> the first wave of the next starts (same clock)
print("gaps: verify -> flood_refine %.2f us, flood_refine -> rare %.2f us" % ((bounds[2][0] - bounds[1][1]) * TICK_US, (bounds[3][0] - bounds[2][1]) * TICK_US))
