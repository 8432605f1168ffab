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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = 800
det = A.TagDetector("t36h11")
base, _ = synth.render_batch(0, n, 1284, H, device="cuda")


def case(width, pitch):
    """n frames of `width` columns at `pitch` bytes per row in one allocation"""
    big = torch.zeros((n, H, pitch), dtype=torch.uint8, device="cuda")
    big[:, :, :width] = base[:, :, :width]
    return big


cases = {"1280 wide, pitch 1280 (aligned form)": (1280, 1280), "1282 wide, pitch 1282 (tight, width % 4 = 2)": (1282, 1282),
         "1283 wide, pitch 1283 (tight, odd)": (1283, 1283), "1280 wide, pitch 1284 (padded, 4-byte aligned)": (1280, 1284),
         "1280 wide, pitch 1281 (padded, odd)": (1280, 1281), "1282 wide, pitch 1284 (padded to 4 bytes)": (1282, 1284)}
bufs = {k: case(*v) for k, v in cases.items()}
torch.cuda.synchronize()


def k1(name):
    w, pitch = cases[name]
    b = bufs[name]
    for _ in range(4):
        det.saddles_batch_enqueue_ptr(b.data_ptr(), n, w, H, pitch, pitch * H, _ffi.AGX_L8)
    det.sync()
    det.set_option("profile_stride", 1)
    det.profile_enable(True)
    det.profile_reset()
    for _ in range(10):
        det.saddles_batch_enqueue_ptr(b.data_ptr(), n, w, H, pitch, pitch * H, _ffi.AGX_L8)
    det.sync()
    p = det.profile_read()
    det.profile_enable(False)
    return p["k_blur_hessian"][0] / p["k_blur_hessian"][1]


res = {k: [] for k in cases}
for rnd in range(4):
    for k in (list(cases) if rnd % 2 == 0 else list(cases)[::-1]):
        res[k].append(k1(k))
ref = statistics.median(res["1280 wide, pitch 1280 (aligned form)"]) / (1280 * H * n)
print("K1 (k_blur_hessian), %d frames x %d rows, L8; ms per launch (median of 4 alternating rounds), ns per pixel relative to the aligned form" % (n, H))
for k, v in res.items():
    m = statistics.median(v)
    print("%-52s %.4f ms   %.2f x" % (k, m, m / (cases[k][0] * H * n) / ref))
