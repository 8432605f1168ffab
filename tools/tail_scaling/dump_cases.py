"""Saddles + u8 luma of the first N frames of the bench's generator (configs[1]) for tools/tail_scaling:
the saddles come from the CPU oracle here (no GPU needed), or from the HIP chain with --gpu.
    python tools/tail_scaling/dump_cases.py /tmp/cases.bin 64 [--gpu]"""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth

out, n = sys.argv[1], int(sys.argv[2])
W, H = 1280, 800
if "--gpu" in sys.argv:
    det = A.TagDetector("t36h11", None, device=0)
    fr, _ = synth.render_batch(0, n, W, H, device="cuda")
    det.saddles_batch_enqueue(fr)
    lists, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    frames = fr.cpu().numpy()
else:
    from oracle import oracle as O
    frames = np.stack([np.asarray(synth.render_frame(i, W, H)[0]) for i in range(n)])
    lists = [O.refined_saddle_points(f) for f in frames]
with open(out, "wb") as f:
    f.write(struct.pack("iii", W, H, n))
    for s, g in zip(lists, frames):
        f.write(struct.pack("i", len(s)))
        f.write(np.ascontiguousarray(s).tobytes())
        f.write(np.ascontiguousarray(g).tobytes())
print(n, "frames,", sum(len(s) for s in lists) / n, "saddles per frame")
