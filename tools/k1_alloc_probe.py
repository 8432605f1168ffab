"""Does the blur kernel's time depend on where the workspace lands?  Fresh detectors in one process (each allocates its
own workspace), K1 timed by events over 30 batches each; prints the blur plane's device address next to the time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
frames, _ = synth.render_batch(0, 256, 1280, 800, device="cuda")
keep = []
for trial in range(int(os.environ.get("TRIALS", "10"))):
    if trial % 3 == 2:  # shift the allocator's state between some trials
        keep.append(torch.empty(int(np.random.default_rng(trial).integers(1, 64)) << 20, dtype=torch.uint8, device="cuda"))
    det = A.TagDetector("t36h11")
    for _ in range(12): det.saddles_batch_enqueue(frames)
    det.sync()
    det.profile_enable(1); det.profile_reset()
    for _ in range(30): det.saddles_batch_enqueue(frames)
    det.sync()
    ms, n = det.profile_read()["k_blur_hessian"]
    addr = det.debug_fetch(0, "redzones")["buffer0_address"]
    print("trial %2d: K1 %.4f ms   blur plane at 0x%x (mod 2 MiB: 0x%06x, mod 4 KiB: 0x%03x)  frames at 0x%x" % (trial, ms / n, addr, addr & 0x1fffff, addr & 0xfff, frames.data_ptr()), flush=True)
    det.close()
