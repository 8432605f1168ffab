"""K1 on frames that do not take its aligned form: widths that are not multiples of 4 and row pitches
that are not multiples of 4 bytes run the non-A4 instantiation (byte-gathering loads, scalar non-temporal stores, per-row
pointers).  Per-pixel K1 time against the aligned 1280x800 batch, alternating in one process.
    python tools/k1_unaligned.py [frames=64]"""
import statistics
import sys
sys.path.insert(0, ".")
import torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import _ffi, synth

import os
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FMT = os.environ.get("FORMAT", "L8")
BPP = {"L8": 1, "L16": 2, "RGB8": 3}[FMT]
CODE = {"L8": _ffi.AGX_L8, "L16": _ffi.AGX_L16, "RGB8": _ffi.AGX_RGB8}[FMT]
H = 800
det = A.TagDetector("t36h11")
base, _ = synth.render_batch(0, n, 1284, H, device="cuda", fmt=FMT)


def case(width, pad):
    """n frames of `width` columns with `pad` bytes after every row in one allocation"""
    rb = width * BPP
    big = torch.zeros((n, H, rb + pad), dtype=torch.uint8, device="cuda")
    big[:, :, :rb] = base[:, :, :width].contiguous().view(torch.uint8).reshape(n, H, rb)
    return big


cases = {"1280 wide, pitch 1280 (aligned form)": (1280, 0), "1282 wide, pitch 1282 (tight, width % 4 = 2)": (1282, 0),
         "1283 wide, pitch 1283 (tight, odd)": (1283, 0), "1280 wide, pitch 1284 (padded, 4-byte aligned)": (1280, 4),
         "1280 wide, 1 pad byte per row (L16: 2; odd pitch)": (1280, 2 if FMT == "L16" else 1), "1282 wide, 2 pad bytes per row": (1282, 2)}
bufs = {k: case(*v) for k, v in cases.items()}
torch.cuda.synchronize()


def k1(name):
    w, pad = cases[name]
    pitch = w * BPP + pad
    b = bufs[name]
    for _ in range(4):
        det.saddles_batch_enqueue_ptr(b.data_ptr(), n, w, H, pitch, pitch * H, CODE)
    det.sync()
    det.set_option("profile_stride", 1)
    det.profile_enable(True)
    det.profile_reset()
    for _ in range(10):
        det.saddles_batch_enqueue_ptr(b.data_ptr(), n, w, H, pitch, pitch * H, CODE)
    det.sync()
    p = det.profile_read()
    det.profile_enable(False)
    return p["k_blur_hessian"][0] / p["k_blur_hessian"][1]


res = {k: [] for k in cases}
for rnd in range(4):
    for k in (list(cases) if rnd % 2 == 0 else list(cases)[::-1]):
        res[k].append(k1(k))
ref = statistics.median(res["1280 wide, pitch 1280 (aligned form)"]) / (1280 * H * n)
print("K1 (k_blur_hessian), %d frames x %d rows, " % (n, H) + FMT + " (pitches in pixels of this format + pad bytes); ms per launch (median of 4 alternating rounds), ns per pixel relative to the aligned form")
for k, v in res.items():
    m = statistics.median(v)
    print("%-52s %.4f ms   %.2f x" % (k, m, m / (cases[k][0] * H * n) / ref))
