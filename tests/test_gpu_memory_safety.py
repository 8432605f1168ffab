"""No kernel of the chain writes outside its buffers.  A handle created with AGX_REDZONE_BYTES set puts
guard bytes (0xA5) in front of and behind every workspace buffer; after each workload -- the paths
where an index could run away: list overflows, the generic clustering path, oversized components,
rows that are not 4-byte aligned (dummy-row stores), lists beyond the LDS sort, tiny frames, every
segment height, and agx_detect_batch's device tail (its code list, its tag rows and frame table in
mapped pinned host memory, the staging and luma planes carry the same guards) -- the guards must be intact (agx_debug_fetch AGX_DBG_REDZONES) and the results still
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


def _tail_buffers_intact(det, what, at_least):
    r = det.debug_fetch(0, "redzones")
    assert r["buffers"] >= at_least, (what, r)
    assert r["damaged_bytes"] == 0, "%s: %s" % (what, r)
    return r["buffers"]


def test_the_device_tail_stays_inside_its_buffers(oracle):
    """agx_detect_batch with the board search + decode on the device: the kernel writes tags and the frame table straight into
    mapped pinned host memory and reads the family's code list, the staged frames and (L16 / RGB8) the luma planes -- all of
    them allocated with the handle's guard bytes.  Workloads: the benchmark's frames, frames of > 512 saddles, 10 x 10 boards
    (handed back), a tag capacity too small for the frame's tags, L16 / RGB8 chunks, a call of several 1024-frame chunks on two
    host threads, the wide debug band (hand-backs that read the compact saddle list).  Guards intact, tags the host tail's."""
    import aprilgrid_rs_amd as A
    from tests.util import check_tags, oracle_detect_parallel
    synth = synth_module()
    d = _guarded_detector()
    d.set_option("device_tail", 1)
    host = A.TagDetector("t36h11", None, device=0)
    host.set_option("device_tail", 0)

    def both(frames, what, cap=128, threads=0, **kw):
        rc_h, out_h, cnt_h, st_h = host.detect_batch_raw(frames, n_threads=threads, cap=cap, **kw)
        rc_d, out_d, cnt_d, st_d = d.detect_batch_raw(frames, n_threads=threads, cap=cap, **kw)
        assert rc_h == rc_d and np.array_equal(st_h, st_d) and np.array_equal(cnt_h, cnt_d), what
        for f in range(len(frames)):
            if st_h[f] == 0:
                assert out_h[f, : cnt_h[f]].tobytes() == out_d[f, : cnt_d[f]].tobytes(), "%s frame %d" % (what, f)
        assert d.get_option("last_device_tail_frames") == len(frames)
        return cnt_d, st_d

    # the benchmark's frames (host memory in: the staging buffer; a device copy as well)
    fr, _ = synth.render_batch(0, 64, 1280, 800, device="cuda")
    frames = fr.cpu().numpy()
    cnt, _ = both(frames, "bench frames")
    n0 = _tail_buffers_intact(d, "bench frames", 24)  # workspace + staging + code list + tag rows + frame table
    both(frames, "bench frames from a device copy", device_frames=fr)
    _tail_buffers_intact(d, "bench frames, device copy", n0)
    refs = oracle_detect_parallel(oracle, frames[:8], threads=8)
    got = d.detect_batch(frames[:8], n_threads=2)
    for i in range(8):
        check_tags(got[i], refs[i], "guarded handle, bench frame %d" % i)
    # the caller's tag capacity below the frames' tags: the rows of the table keep their stride, nothing is written past cap
    cap = int(cnt.max()) - 3
    _, st = both(frames[:32], "capacity", cap=cap)
    assert (st != 0).any()
    _tail_buffers_intact(d, "tag capacity too small", n0)
    # > 512 saddles per frame; 10 x 10 boards (more cells than the kernel's board holds: handed back)
    a = synth.render_batch(500, 24, 640, 480, device="cuda")[0].cpu().numpy()
    wide = np.ascontiguousarray(np.concatenate([a[0::3], a[1::3], a[2::3]], axis=2))
    both(wide, "> 512 saddles")
    _tail_buffers_intact(d, "> 512 saddles", n0)
    big = np.stack([synth.render_frame(5 + i, 1280, 800, spec=synth.BoardSpec(rows=10, cols=10))[0].numpy() for i in range(3)])
    both(big, "10 x 10 boards", cap=256)
    assert d.get_option("last_device_tail_fallbacks") == 3
    _tail_buffers_intact(d, "10 x 10 boards", n0)
    # L16 / RGB8: the device's luma planes join the guarded buffers
    for fmt in ("L16", "RGB8"):
        f = synth.render_batch(300, 40, 320, 240, device="cuda", fmt=fmt)[0].cpu().numpy()
        if fmt == "L16":
            f = f.view(np.uint16)
        both(f, fmt)
        _tail_buffers_intact(d, fmt, n0 + 1)
    # pure noise (every frame beyond the kernel's saddle list) and frames with nothing in them
    noise = synth.render_batch(3, 3, 640, 480, device="cuda", pure_noise=True)[0].cpu().numpy()
    both(np.concatenate([noise, np.full((2, 480, 640), 128, np.uint8)]), "noise and flat frames")
    _tail_buffers_intact(d, "noise and flat frames", n0)
    # several chunks (2 150 frames = three of up to 1 024) on two host threads, the staging slots in turn
    small = synth.render_batch(1200, 96, 320, 240, device="cuda")[0].cpu().numpy()
    many = np.concatenate([small] * 23)[:2150]
    rc, out, cnt_m, st_m = d.detect_batch_raw(many, n_threads=2, cap=64)
    rc_h, out_h, cnt_h, _ = host.detect_batch_raw(small, n_threads=4, cap=64)
    assert rc == 0 == rc_h and d.get_option("last_device_tail_frames") == 2150
    for f in range(2150):
        assert cnt_m[f] == cnt_h[f % 96] and out[f, : cnt_m[f]].tobytes() == out_h[f % 96, : cnt_h[f % 96]].tobytes(), f
    _tail_buffers_intact(d, "2 150 frames in three chunks", n0)
    # the hand-back path on purpose (frames that read the chain's compact list on the host)
    d.set_option("tail_debug_band", 50)
    both(frames, "wide debug band")
    assert d.get_option("last_device_tail_uncertain") > 0
    _tail_buffers_intact(d, "wide debug band", n0)
    d.set_option("tail_debug_band", 0)
    d.close()
    host.close()


def test_a_write_into_a_tail_buffers_guard_is_seen():
    """The check covers the side buffers: a byte written behind the device tail's frame table (mapped pinned host memory) is
    reported with that buffer's number."""
    import ctypes as C
    synth = synth_module()
    d = _guarded_detector()
    d.set_option("device_tail", 1)
    frames = synth.render_batch(300, 8, 320, 240, device="cuda")[0].cpu().numpy()
    d.detect_batch(frames, n_threads=2)
    r = d.debug_fetch(0, "redzones")
    assert r["damaged_bytes"] == 0 and r["buffers"] >= 24
    n = C.c_size_t(0)
    addr = np.zeros(2, np.uint64)
    # (debug item 11: host address and payload bytes of the device tail's frame table)
    d._check(d._lib.agx_debug_fetch(d._h, 0, 11, addr.ctypes.data, addr.nbytes, C.byref(n)))
    C.memset(int(addr[0]) + int(addr[1]) + 5, 0, 2)  # two bytes of the guard behind the table
    r = d.debug_fetch(0, "redzones")
    assert r["damaged_bytes"] == 2 and r["first_buffer"] == r["buffers"] - 1 and r["first_offset"] == int(addr[1]) + 5, r
    d.close()
