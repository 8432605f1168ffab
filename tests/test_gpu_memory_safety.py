"""No kernel of the chain writes outside its buffers.  A handle created with AGX_REDZONE_BYTES set puts
guard bytes (0xA5) in front of and behind every workspace buffer; after each workload -- the paths
where an index could run away: list overflows, the generic clustering path, oversized components,
rows that are not 4-byte aligned (dummy-row stores), lists beyond the LDS sort, tiny frames, every
segment height -- the guards must be intact (agx_debug_fetch AGX_DBG_REDZONES) and the results still
equal the oracle's."""
import os

import numpy as np
import pytest

from tests.util import check_saddles, load_image, oracle_saddles_parallel, synth_module

pytestmark = pytest.mark.gpu

GUARD = 1 << 16


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


def _guarded_detector():
    import aprilgrid_rs_amd as A
    os.environ["AGX_REDZONE_BYTES"] = str(GUARD)
    try:
        return A.TagDetector("t36h11", None, device=0)
    finally:
        del os.environ["AGX_REDZONE_BYTES"]


@pytest.fixture
def gdet():
    d = _guarded_detector()
    yield d
    d.close()


def _intact(det, what):
    r = det.debug_fetch(0, "redzones")
    assert r["buffers"] >= 20, r
    assert r["damaged_bytes"] == 0, "%s: %s" % (what, r)


def _run(det, oracle, frames, host, what, n_check=2):
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch(raise_on_overflow=False)
    _intact(det, what)
    refs = oracle_saddles_parallel(oracle, host[:n_check], threads=2)
    for i in range(min(n_check, len(res))):
        assert status[i] == 0
        check_saddles(res[i], refs[i], "%s frame %d" % (what, i))
    return res, status


def test_guard_bytes_are_checked(gdet):
    """The check itself: a deliberate write into a guard is reported with its buffer and offset."""
    import torch
    synth = synth_module()
    fr, _ = synth.render_batch(1, 1, 320, 240, device="cuda")
    gdet.saddles_batch_enqueue(fr)
    gdet.saddles_batch_fetch()
    _intact(gdet, "before")
    blur_ptr = gdet.debug_fetch(0, "redzones")["buffer0_address"]
    import ctypes as C
    # the HIP runtime this process already runs on (never a second copy): its path from the memory map
    path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l)
    hip = C.CDLL(path)
    one = np.zeros(4, np.uint8)
    assert hip.hipMemcpy(C.c_void_p(blur_ptr - 8), C.c_void_p(one.ctypes.data), C.c_size_t(4), 1) == 0  # host to device
    r = gdet.debug_fetch(0, "redzones")
    assert r["damaged_bytes"] == 4 and r["first_buffer"] == 0 and r["first_offset"] == -8, r


def test_boards_noise_formats_and_unaligned_rows(gdet, oracle):
    import torch
    synth = synth_module()
    fr, _ = synth.render_batch(10, 8, 640, 400, device="cuda")
    _run(gdet, oracle, fr, fr.cpu().numpy(), "boards 640x400")
    fr, _ = synth.render_batch(3, 4, 640, 400, device="cuda", pure_noise=True)
    _run(gdet, oracle, fr, fr.cpu().numpy(), "noise 640x400")
    for fmt, width in (("L8", 301), ("RGB8", 203), ("L16", 250), ("L8", 1283)):
        fr, _ = synth.render_batch(20, 3, (width + 3) // 4 * 4, 97, device="cuda", fmt=fmt)
        frames = fr[:, :, :width].contiguous()
        host = frames.cpu().numpy()
        if fmt == "L16":
            host = host.view(np.uint16)
        _run(gdet, oracle, frames, host, "%s width %d" % (fmt, width))
    fr, _ = synth.render_batch(30, 3, 644, 131, device="cuda")
    planes = (fr.to(torch.float32) / 255.0).contiguous()
    _run(gdet, oracle, planes, planes.cpu().numpy(), "LF32")


def test_tiny_frames_and_every_segment_height(gdet, oracle):
    import torch
    rng = np.random.default_rng(8)
    for shape in ((2, 2), (3, 5), (9, 9), (10, 11), (37, 53), (33, 260)):
        img = rng.integers(0, 256, (2,) + shape, dtype=np.uint8)
        _run(gdet, oracle, torch.from_numpy(img).cuda(), img, "tiny %dx%d" % shape)
    synth = synth_module()
    fr, _ = synth.render_batch(5, 6, 1280, 810, device="cuda")
    host = fr.cpu().numpy()
    for rows in (32, 64, 96, 128, 0):
        gdet.set_option("k1_rows_per_segment", rows)
        _run(gdet, oracle, fr, host, "segments of %d rows" % rows, n_check=1)


def test_oversized_components_and_the_generic_path(gdet, oracle):
    import torch
    h, w = 240, 320
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    big = (32768 + 25000 * np.sin(2 * np.pi / 150.0 * xx) * np.sin(2 * np.pi / 150.0 * yy)).astype(np.uint16)
    mid = (32768 + 25000 * np.sin(2 * np.pi / 64.0 * xx) * np.sin(2 * np.pi / 64.0 * yy)).astype(np.uint16)
    frames = np.stack([big, mid, big])
    t = torch.from_numpy(frames.view(np.int16)).cuda()
    _run(gdet, oracle, t, frames, "oversized components", n_check=3)
    c = gdet.debug_fetch(0, "counters")
    assert c["flags"] & 16, c  # the generic path ran
    gdet.set_option("force_generic", 1)
    img = load_image("EuRoC.png")
    got = gdet.refined_saddle_points(img, as_array=True)
    _intact(gdet, "forced generic path")
    check_saddles(got, oracle.refined_saddle_points(img), "forced generic path")
    gdet.set_option("force_generic", 0)


def test_list_overflows_stay_inside_their_buffers(oracle):
    """Every capacity too small at once (candidates, clusters, saddles): the kernels flag the frames
    and write nothing past the short lists."""
    import aprilgrid_rs_amd as A
    synth = synth_module()
    d = _guarded_detector()
    d.set_limits(max_candidates=4096, max_clusters=256, max_saddles=64)
    for pure_noise, force in ((False, 0), (True, 0), (False, 1), (True, 1)):
        d.set_option("force_generic", force)
        fr, _ = synth.render_batch(40, 4, 640, 400, device="cuda", pure_noise=pure_noise)
        d.saddles_batch_enqueue(fr)
        res, status = d.saddles_batch_fetch(raise_on_overflow=False)
        assert (status == -3).any(), status
        _intact(d, "overflow noise=%s generic=%d" % (pure_noise, force))
    # caller-owned result buffers that are too small
    d.set_limits(0, 0, 0)
    d.set_option("force_generic", 0)
    fr, _ = synth.render_batch(40, 4, 640, 400, device="cuda")
    import torch
    whole = torch.full((64 + 8192, 5), 7.0, dtype=torch.float32, device="cuda")
    out_s = whole[:64]  # 64 records for ~2000 saddles; the rest of the allocation must stay untouched
    out_t = torch.zeros((4, 4), dtype=torch.int32, device="cuda")
    d.saddles_batch_enqueue_to(fr, out_s, out_t)
    d.sync()
    _intact(d, "short external buffer")
    assert ((out_t.cpu().numpy()[:, 2] & 4) != 0).any()  # AGX_FRAME_SADDLE_OVERFLOW reported
    assert bool((whole[64:] == 7.0).all()), "records written past the caller's buffer"
    d.close()


def test_lists_beyond_the_lds_sort(gdet, oracle):
    """> 16384 saddles per frame: ordered in global memory (k_rare's large-list path)."""
    synth = synth_module()
    fr, _ = synth.render_batch(3, 1, 2048, 1536, device="cuda", pure_noise=True)
    res, status = _run(gdet, oracle, fr, fr.cpu().numpy(), "2048x1536 noise", n_check=1)
    assert len(res[0]) > 16384
