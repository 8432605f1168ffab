"""GPU parity: the HIP chain (through the C ABI) against the CPU oracle on the same inputs.

Bar (DESIGN.md "Parity"): blur plane, response plane, per-frame min, cluster table
(first pixel, size, centroid) and the saddle coordinates x, y and strength k are BIT-EXACT;
theta and phi go through acos/atan2 (device libm vs glibc) and must agree within
ANGLE_TOL_DEG; saddle order and count are identical; tag ids identical and tag corners
bit-identical (they are saddle coordinates).
"""
import numpy as np
import pytest

from tests.util import (ALL_IMAGES, ANGLE_TOL_DEG, REFERENCE_TAG_COUNTS, bits_equal, check_frame, check_saddles, load_image,
                        oracle_saddles_parallel, synth_module)

pytestmark = pytest.mark.gpu



@pytest.fixture(scope="module")
def det():
    import aprilgrid_rs_amd as A
    d = A.TagDetector(A.TagFamily.T36H11, None, device=0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="module")
def det_resp():
    """Detector whose blur kernel also stores the response it evaluates in registers."""
    import aprilgrid_rs_amd as A
    d = A.TagDetector(A.TagFamily.T36H11, None, device=0)
    d.set_option("store_response", 1)
    yield d
    d.close()


@pytest.mark.parametrize("name", ALL_IMAGES)
def test_fixture_images_in_register_response(det_resp, oracle, name):
    """The Hessian response as the blur kernel itself evaluates it (image_util.rs:88-106), bit for bit."""
    img = load_image(name)
    got = det_resp.refined_saddle_points(img, as_array=True)
    check_saddles(got, check_frame(det_resp, oracle, img, 0, name), name)


@pytest.mark.parametrize("shape", [(2, 2), (3, 5), (10, 11), (37, 53), (33, 260), (100, 1023), (24, 2100)])
def test_ragged_sizes_in_register_response(det_resp, oracle, shape):
    rng = np.random.default_rng(shape[0] * 977 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    got = det_resp.refined_saddle_points(img, as_array=True)
    check_saddles(got, check_frame(det_resp, oracle, img, 0, "random %dx%d" % shape), str(shape))


@pytest.mark.parametrize("name", ALL_IMAGES)
def test_fixture_images_chain(det, oracle, name):
    img = load_image(name)
    got = det.refined_saddle_points(img, as_array=True)
    ref = check_frame(det, oracle, img, 0, name)
    check_saddles(got, ref, name)
    c = det.debug_fetch(0, "counters")
    assert not (c["flags"] & 16) and c["generic_candidates"] == 0, c  # real frames stay on the flood path
    if name in ("EuRoC.png", "TUM_VI.png", "two_boards.png"):
        assert c["big_seeds"] > 0, c  # ... and these have components for the wave-wide second tier


@pytest.mark.parametrize("name,expected", REFERENCE_TAG_COUNTS)
def test_fixture_images_detect(det, oracle, name, expected):
    """The reference's own end-to-end assertions (tests/test_detector.rs:26-32) through the
    product path, plus id / corner parity with the oracle."""
    img = load_image(name)
    tags = det.detect(img)
    assert len(tags) == expected
    ref = oracle.detect(img)
    assert sorted(tags) == sorted(ref)
    for tid in ref:
        assert bits_equal(tags[tid], ref[tid]), "corners of tag %d" % tid


@pytest.mark.parametrize("name", ALL_IMAGES)
def test_the_images_own_geometry_through_the_hip_path(det, name):
    """tests/grid_pins.py through the product: agx_detect, and agx_detect_batch with the board search + decode on the device
    -- ids exactly 0 .. N-1, convex quads of one winding, every corner within 1 px of one regular planar grid seen through a
    smooth camera.  No oracle involved: the images themselves are the reference here (tests/test_detector.rs:21-32 asserts len())."""
    import aprilgrid_rs_amd as A
    from tests import grid_pins
    img = load_image(name)
    expected = dict(REFERENCE_TAG_COUNTS).get(name)
    r = grid_pins.check_image(name, det.detect(img), img.shape, expected)
    assert r["camera_max_px"] < 0.75, r
    d = A.TagDetector("t36h11", None, device=0)
    d.set_option("device_tail", 1)
    try:
        tags = d.detect_batch(img[None], n_threads=2, cap=128)[0]
        assert d.get_option("last_device_tail_frames") == 1
        r2 = grid_pins.check_image(name, tags, img.shape, expected)
        assert r2 == r  # (the same corners, hence the same residuals)
    finally:
        d.close()


@pytest.mark.parametrize("family", ["T16H5", "T25H7", "T25H9", "T36H11B1"])
def test_detect_other_tag_families(oracle, family):
    """TagDetector::new with the other families (src/detector.rs:369-405) on boards rendered with their
    code tables: detect() through the GPU chain equals the oracle's and finds the 25 drawn tags."""
    import aprilgrid_rs_amd as A
    synth = synth_module()
    d = A.TagDetector(family, None, device=0)
    img, gt = synth.render_frame(21, 800, 600, spec=synth.BoardSpec(rows=5, cols=5), family=family)
    img = img.numpy()
    got, ref = d.detect(img), oracle.detect(img, family=family)
    assert sorted(got) == sorted(ref) == sorted(gt)
    for tid in ref:
        assert bits_equal(got[tid], ref[tid])
    d.close()


def test_device_luma8_of_detect_equals_the_hosts(det, oracle):
    """agx_detect converts L16 / RGB8 frames to the u8 luma of the decode (detector.rs:507) on the device:
    every u16 value, random RGB triples and the grey axis must give the bytes of the host's agx_luma8 and
    of the oracle's restatement of image 0.25.9."""
    import aprilgrid_rs_amd as A
    rng = np.random.default_rng(21)
    l16 = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    rgb = rng.integers(0, 256, (300, 401, 3), dtype=np.uint8)
    grey = np.repeat(np.arange(256, dtype=np.uint8), 3).reshape(16, 16, 3)
    sat = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0], [1, 1, 1], [254, 255, 255], [13, 200, 77]]] * 4, np.uint8)
    for img in (l16, rgb, grey, sat, load_image("iphone.png"), load_image("TUM_VI.png")):
        det.detect(img)
        got = det.debug_fetch(0, "luma8", img.shape[:2])
        assert np.array_equal(got, A.TagDetector.luma8(img)) and np.array_equal(got, oracle.luma_u8(img)), img.shape


def test_detect_with_tail_threads(oracle):
    """Option "tail_threads": one frame's board search on several host threads gives the tags of the
    sequential search, in the same order, on every fixture image."""
    import aprilgrid_rs_amd as A
    d = A.TagDetector("t36h11", None, device=0)
    for name in ALL_IMAGES:
        img = load_image(name)
        d.set_option("tail_threads", 1)
        one = d.detect(img)
        for n in (2, 5, 16):
            d.set_option("tail_threads", n)
            assert d.get_option("tail_threads") == n
            par = d.detect(img)
            assert list(par) == list(one) and all(bits_equal(par[t], one[t]) for t in one), (name, n)
    d.close()


def test_detect_kornia_front_end(det, oracle):
    """tests/test_detector.rs:35-43: Image<u8,3> -> 66 tags; u8c1 works; other N is refused."""
    import aprilgrid_rs_amd as A
    img = load_image("iphone.png")
    assert len(det.detect_kornia(img)) == 66
    g = load_image("EuRoC.png")[:, :, None]
    assert len(det.detect_kornia(g)) == 36
    with pytest.raises(A.AgxError):
        det.detect_kornia(np.zeros((16, 16, 4), np.uint8))


def test_all_pixel_values_exact(det, oracle):
    """Every u8 and every u16 input value goes through the division-free luma conversion:
    planes must still be bit-identical to the oracle's true division."""
    v8 = np.repeat(np.repeat(np.arange(256, dtype=np.uint8).reshape(16, 16), 8, 0), 8, 1)
    v8 = np.ascontiguousarray(np.hstack([v8, v8[::-1]]))
    det.refined_saddle_points(v8, as_array=True)
    check_frame(det, oracle, v8, 0, "all u8 values")
    v16 = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    det.refined_saddle_points(v16, as_array=True)
    check_frame(det, oracle, v16, 0, "all u16 values")
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    det.refined_saddle_points(rgb, as_array=True)
    check_frame(det, oracle, rgb, 0, "random rgb")


@pytest.mark.parametrize("shape", [(2, 2), (3, 5), (9, 9), (10, 11), (37, 53), (64, 64), (33, 260), (100, 1023),
                                   (16, 2044), (24, 2100), (12, 4100)])
def test_ragged_sizes(det, oracle, shape):
    """Sizes that are not multiples of 4 / of the wave, narrower than the blur radius, wider
    than one strip (multi-strip with halo lanes), tiny."""
    rng = np.random.default_rng(shape[0] * 10007 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    got = det.refined_saddle_points(img, as_array=True)
    ref = check_frame(det, oracle, img, 0, "random %dx%d" % shape)
    check_saddles(got, ref, str(shape))


def test_flat_and_empty(det, oracle):
    """Flat image: min = 0, threshold 0, no candidates -> empty result (detector.rs:432-434)."""
    img = np.full((64, 80), 77, np.uint8)
    assert len(det.refined_saddle_points(img, as_array=True)) == 0
    assert det.detect(img) == {}
    check_frame(det, oracle, img, 0, "flat")


def test_degenerate_and_bad_arguments(det):
    import aprilgrid_rs_amd as A
    with pytest.raises(A.AgxError) as e:
        det.refined_saddle_points(np.zeros((1, 40), np.uint8))  # reference: height()-1 underflow panic
    assert e.value.status == -1
    with pytest.raises(A.AgxError) as e:
        det.refined_saddle_points(np.zeros((8, 8), np.float64))
    assert e.value.status == -2


def test_params_change_filter(oracle):
    import aprilgrid_rs_amd as A
    img = load_image("EuRoC.png")
    p = A.DetectorParams(0.3, 40.0, 50.0, 1)
    d = A.TagDetector("t36h11", p, device=0)
    op = oracle.default_params()
    op.min_saddle_angle, op.max_saddle_angle, op.max_num_of_boards = 40.0, 50.0, 1
    check_saddles(d.refined_saddle_points(img, as_array=True), oracle.refined_saddle_points(img, params=op), "params")
    d.close()


def test_batch_matches_oracle_and_is_deterministic(det, oracle):
    """A batch of synthetic frames resident in HBM, every frame against the oracle; a second
    run of the same batch must reproduce the first bit for bit (union-find / atomics order
    must not leak into results)."""
    import torch
    synth = synth_module()
    frames, gts = synth.render_batch(0, 6, 640, 400, device="cuda")
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    host = frames.cpu().numpy()
    for i in range(frames.shape[0]):
        ref = check_frame(det, oracle, host[i], i, "synthetic frame %d" % i)
        check_saddles(res[i], ref, "synthetic frame %d" % i)
    det.saddles_batch_enqueue(frames)
    res2, _ = det.saddles_batch_fetch()
    for a, b in zip(res, res2):
        assert a.tobytes() == b.tobytes()
    # ground truth by construction: every detected tag has the right id and corners near truth
    tags = det.detect(host[0])
    assert len(tags) >= 30
    for tid, c in tags.items():
        g = gts[0][tid]
        for p in c:
            assert np.min(np.hypot(*(g - p).T)) < 0.5


@pytest.mark.parametrize("bits", [16384, 32768, 65536, 4096, 8192, 128, 2048])
def test_diagnostic_instantiations_do_not_change_results(det, bits):
    """debug_ablation 16384 runs the flood + refine kernel's second instantiation (phase clock), 32768 K1's ascending
    segment order (the A/B of the middle-outwards dispatch), 65536 the uniform refresh interval of its threshold, 4096 / 8192 the wave and phase timers,
    128 / 2048 k_verify_seeds' statistics (8192, 128, 2048: its instantiation with the debug tests, on the linear grid): the superset K1
    leaves depends on the order in which its waves learn the frame's minimum, the results must not."""
    import aprilgrid_rs_amd as A
    synth = synth_module()
    frames, _ = synth.render_batch(40, 5, 1280, 800, device="cuda")
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    d2 = A.TagDetector(A.TagFamily.T36H11, None, device=0)
    try:
        d2.set_option("debug_ablation", bits)
        d2.saddles_batch_enqueue(frames)
        res2, status2 = d2.saddles_batch_fetch()
        assert (status2 == 0).all()
        for a, b in zip(res, res2):
            assert a.tobytes() == b.tobytes()
        if bits == 16384:  # the clock has run: chunks counted, time recorded in every phase of a working wave
            st = d2.debug_fetch(0, "verify_stats").astype(np.int64)
            assert st[19] > 0 and st[18] > 0 and all(st[k] > 0 for k in (8, 9, 10, 11, 13, 16, 7))
    finally:
        d2.close()


def test_verify_kernel_linear_grid_equals_tile_grid(det, monkeypatch):
    """k_verify_seeds is launched as a (frame, column group, row chunk) grid, one tile per wave; the tuning override AGX_G_VERIFY keeps the
    linear slot-major grid with several tiles per wave.  Same lists either way."""
    import aprilgrid_rs_amd as A
    synth = synth_module()
    frames, _ = synth.render_batch(7, 3, 1000, 600, device="cuda")
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    monkeypatch.setenv("AGX_G_VERIFY", "37")
    d2 = A.TagDetector(A.TagFamily.T36H11, None, device=0)
    try:
        d2.saddles_batch_enqueue(frames)
        res2, status2 = d2.saddles_batch_fetch()
        assert (status2 == 0).all()
        for a, b in zip(res, res2):
            assert a.tobytes() == b.tobytes()
    finally:
        d2.close()
        monkeypatch.delenv("AGX_G_VERIFY")
        det.set_option("reload_tuning_env", 1)  # (the overrides are read at detector creation and kept)


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_chain_pipeline_batches_in_flight(det, oracle, depth):
    """Several batches in flight on separate streams (sharding.ChainPipeline): every batch's
    device-resident result must equal what one detector gives for that batch alone."""
    import torch
    import aprilgrid_rs_amd as A
    from aprilgrid_rs_amd import sharding
    synth = synth_module()
    batches = [synth.render_batch(100 + 10 * b, 4, 480, 320, device="cuda")[0] for b in range(5)]
    want = []
    for fr in batches:
        det.saddles_batch_enqueue(fr)
        res, status = det.saddles_batch_fetch()
        assert (status == 0).all()
        want.append(res)
    ref0 = oracle.refined_saddle_points(batches[0][0].cpu().numpy())
    check_saddles(want[0][0], ref0, "pipeline reference frame")
    pipe = sharding.ChainPipeline(A.TagFamily.T36H11, 4, torch.device("cuda", 0), depth=depth)
    try:
        kept = []
        for b, fr in enumerate(batches):
            pipe.submit(fr)
            slot = pipe.gather.i
            kept.append((b, pipe.gather.bufs[slot]))
            # a slot is reused `len(bufs)` submissions later: collect the batch that is about to be overwritten
            if len(kept) == len(pipe.gather.bufs):
                pipe.finish()
                for bb, (s_buf, t_buf) in kept:
                    got = sharding.unpack_frames(s_buf, t_buf)
                    for i in range(4):
                        assert got[i] is not None
                        g = np.zeros(len(got[i]), want[bb][i].dtype)
                        for j, f in enumerate(("x", "y", "k", "theta", "phi")):
                            g[f] = got[i][:, j]
                        assert g.tobytes() == want[bb][i].tobytes(), "batch %d frame %d" % (bb, i)
                kept = []
        pipe.finish()
    finally:
        pipe.close()


@pytest.mark.parametrize("fmt", ["L16", "RGB8"])
def test_batch_other_formats(det, oracle, fmt):
    synth = synth_module()
    frames, _ = synth.render_batch(40, 3, 320, 240, device="cuda", fmt=fmt)
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    host = frames.cpu().numpy()
    if fmt == "L16":
        host = host.view(np.uint16)
    for i in range(len(res)):
        ref = check_frame(det, oracle, host[i], i, "%s frame %d" % (fmt, i))
        check_saddles(res[i], ref, "%s frame %d" % (fmt, i))


def test_noise_frames_many_clusters(det, oracle):
    """Worst case for the sparse stages: pure noise (about 30 % of the pixels are candidates)."""
    synth = synth_module()
    frames, _ = synth.render_batch(7, 2, 320, 200, device="cuda", pure_noise=True)
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    host = frames.cpu().numpy()
    for i in range(2):
        ref = check_frame(det, oracle, host[i], i, "noise frame %d" % i)
        check_saddles(res[i], ref, "noise %d" % i)


@pytest.mark.parametrize("name", ["EuRoC.png", "r45.png", "two_boards.png"])
def test_generic_cluster_path_equals_flood_path(oracle, name):
    """The guarded union-find fallback must give exactly what the windowed flood fill gives."""
    import aprilgrid_rs_amd as A
    d = A.TagDetector("t36h11", None, device=0)
    d.set_option("force_generic", 1)
    img = load_image(name)
    got = d.refined_saddle_points(img, as_array=True)
    ref = check_frame(d, oracle, img, 0, name + " (generic path)")
    check_saddles(got, ref, name + " (generic path)")
    d.close()


def test_oversized_components_take_second_tier_and_generic_path(det, oracle):
    """Components larger than the 32x32 flood window go to the wave-wide 128x64 flood; larger
    ones still (here sin*sin in 16 bit: diamonds of ~9000 px) send the frame to the generic
    kernels.  Results must match the oracle in every case, and the neighbouring frame of the
    batch stays on the fast path."""
    import torch
    h, w = 240, 320
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    om = 2 * np.pi / 150.0
    big = (32768 + 25000 * np.sin(om * xx) * np.sin(om * yy)).astype(np.uint16)
    om2 = 2 * np.pi / 64.0
    mid = (32768 + 25000 * np.sin(om2 * xx) * np.sin(om2 * yy)).astype(np.uint16)  # ~30 px diamonds
    synth = synth_module()
    board, _ = synth.render_batch(3, 1, w, h, device="cpu", fmt="L16")
    frames = np.stack([big, board[0].numpy().view(np.uint16), mid])
    t = torch.from_numpy(frames.view(np.int16)).cuda()
    det.saddles_batch_enqueue(t)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    for i in range(3):
        ref = check_frame(det, oracle, frames[i], i, "frame %d" % i)
        check_saddles(res[i], ref, "frame %d" % i)
    c = [det.debug_fetch(i, "counters") for i in range(3)]
    assert c[0]["flags"] & 16 and c[0]["generic_roots"] > 0, c[0]      # generic path
    assert not (c[1]["flags"] & 16) and c[1]["generic_candidates"] == 0, c[1]  # fast path
    assert c[2]["big_seeds"] > 0, c[2]                                 # second tier exercised


def test_randomised_sizes_formats_batches(det, oracle):
    """Seeded random sweep: widths (aligned and not), heights, formats, batch sizes, board and
    noise frames -- every frame against the oracle (tools/stress_parity.py is the long version)."""
    import torch
    synth = synth_module()
    rng = np.random.default_rng(5)
    for case in range(36):
        fmt = ["L8", "L16", "RGB8"][int(rng.integers(0, 3))]
        w = int(rng.choice([rng.integers(8, 80) * 4, rng.integers(200, 420) * 4, rng.integers(33, 700)]))
        h = int(rng.choice([rng.integers(9, 70), rng.integers(64, 300), rng.integers(300, 700)]))
        noise = bool(rng.integers(0, 3) == 0)
        first = int(rng.integers(0, 1000))
        if w % 4 == 0:
            n = int(rng.integers(1, 5))
            frames, _ = synth.render_batch(first, n, w, h, device="cuda", fmt=fmt, pure_noise=noise)
            host = frames.cpu().numpy()
            if fmt == "L16":
                host = host.view(np.uint16)
            det.saddles_batch_enqueue(frames)
            res, status = det.saddles_batch_fetch()
            assert (status == 0).all()
        else:  # device batches need 4-byte aligned rows: unaligned widths go through the host API
            n = 1
            fr, _ = synth.render_batch(first, 1, (w + 3) // 4 * 4, h, device="cpu", fmt=fmt, pure_noise=noise)
            host = fr.numpy()
            if fmt == "L16":
                host = host.view(np.uint16)
            host = np.ascontiguousarray(host[:, :, :w])
            res = [det.refined_saddle_points(host[0], as_array=True)]
        for i in range(n):
            check_saddles(res[i], oracle.refined_saddle_points(host[i]), "case %d %s %dx%d frame %d" % (case, fmt, w, h, i))


@pytest.mark.parametrize("fmt", ["L8", "L16", "RGB8"])
@pytest.mark.parametrize("width", [33, 250, 701, 1283])
def test_unaligned_width_device_batches(det, oracle, fmt, width):
    """Tightly packed device batches whose rows are not 4-byte aligned: the batch API
    takes them as they are (byte-gathering loads), like the single-frame API and the reference."""
    import torch
    synth = synth_module()
    fr, _ = synth.render_batch(77, 3, (width + 3) // 4 * 4, 96, device="cuda", fmt=fmt)
    frames = fr[:, :, :width].contiguous()
    host = frames.cpu().numpy()
    if fmt == "L16":
        host = host.view(np.uint16)
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    for i in range(3):
        ref = check_frame(det, oracle, host[i], i, "%s width %d frame %d" % (fmt, width, i))
        check_saddles(res[i], ref, "%s width %d frame %d" % (fmt, width, i))


@pytest.mark.parametrize("fmt,width,pad", [("L8", 1282, 0), ("L8", 1281, 2), ("L8", 1280, 1), ("L8", 1283, 1), ("L8", 6, 1), ("L16", 1282, 0),
                                            ("L16", 1281, 2), ("L16", 1280, 2), ("RGB8", 1282, 0), ("RGB8", 1281, 1), ("RGB8", 1280, 3), ("L16", 5, 0),
                                            ("RGB8", 7, 2)])
def test_unaligned_frames_take_the_dword_form_with_the_same_results(oracle, monkeypatch, fmt, width, pad):
    """Round 5: frames whose width is not a multiple of 4 or whose rows are not 4-byte aligned run K1 with the aligned form's
    loads and tap table on unaligned 4 / 8 / 12-byte loads (template parameter UF) instead of gathering bytes.  Full-height
    frames at the bench's size (pad = extra bytes after every row): every intermediate product against the oracle (blur plane,
    the response K1 evaluates in registers, minimum, clusters, lists), and the byte-gathering form (AGX_K1_UNALIGNED_FAST=0)
    gives the same bytes."""
    import torch
    import aprilgrid_rs_amd as A
    from aprilgrid_rs_amd import _ffi
    synth = synth_module()
    n, h = 3, 800 if width > 100 else 37
    fr, _ = synth.render_batch(123, n, (max(width, 8) + 3) // 4 * 4, h, device="cuda", fmt=fmt)
    tight = fr[:, :, :width].contiguous()
    host = tight.cpu().numpy()
    if fmt == "L16":
        host = host.view(np.uint16)
    bpp = {"L8": 1, "L16": 2, "RGB8": 3}[fmt]
    row_bytes = width * bpp
    pitch = row_bytes + pad
    big = torch.full((n * h * pitch + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    big[: n * h * pitch].view(n, h, pitch)[:, :, :row_bytes] = tight.view(torch.uint8).reshape(n, h, row_bytes)
    torch.cuda.synchronize()
    code = {"L8": _ffi.AGX_L8, "L16": _ffi.AGX_L16, "RGB8": _ffi.AGX_RGB8}[fmt]
    got = {}
    for form in ("1", "0"):
        monkeypatch.setenv("AGX_K1_UNALIGNED_FAST", form)
        d = A.TagDetector("t36h11", None, device=0)  # (reads the override)
        try:
            for store_resp in (0, 1):
                d.set_option("store_response", store_resp)
                d.saddles_batch_enqueue_ptr(big.data_ptr(), n, width, h, pitch, pitch * h, code)
                res, status = d.saddles_batch_fetch()
                assert (status == 0).all()
                for i in range(n):
                    ref = check_frame(d, oracle, host[i], i, "%s width %d pitch %d form %s frame %d" % (fmt, width, pitch, form, i))
                    check_saddles(res[i], ref, "%s width %d pitch %d form %s frame %d" % (fmt, width, pitch, form, i))
                got[(form, store_resp)] = [r.tobytes() for r in res]
        finally:
            d.close()
    monkeypatch.delenv("AGX_K1_UNALIGNED_FAST")
    A.TagDetector("t36h11", None, device=0).close()  # (the override is read at creation: gone again)
    assert got[("1", 0)] == got[("0", 0)] == got[("1", 1)] == got[("0", 1)]


@pytest.mark.parametrize("fmt,width,pad_bytes,gap_rows", [("L8", 300, 20, 3), ("L8", 301, 2, 0), ("L16", 250, 12, 1),
                                                          ("RGB8", 100, 4, 2), ("RGB8", 203, 7, 0), ("L8", 1280, 256, 5)])
def test_padded_row_and_frame_strides_on_the_device(det, oracle, fmt, width, pad_bytes, gap_rows):
    """agx_saddles_batch_enqueue takes row_stride_bytes / frame_stride_bytes like the host entry: frames
    cut out of a larger device allocation (padding after every row, unused rows between frames;
    4-byte aligned pitches take the buffer-load path, odd ones the byte-gathering one) give the lists
    of the same frames packed tightly."""
    import torch
    from aprilgrid_rs_amd import _ffi
    synth = synth_module()
    n, h = 3, 72
    fr, _ = synth.render_batch(41, n, (width + 3) // 4 * 4, h, device="cuda", fmt=fmt)
    tight = fr[:, :, :width].contiguous()
    host = tight.cpu().numpy()
    if fmt == "L16":
        host = host.view(np.uint16)
    bpp = {"L8": 1, "L16": 2, "RGB8": 3}[fmt]
    row_bytes = width * bpp
    pitch = row_bytes + pad_bytes
    frame_stride = pitch * (h + gap_rows)
    big = torch.full((n * frame_stride + 64,), 0xA5, dtype=torch.uint8, device="cuda")  # padding holds garbage
    src = tight.view(torch.uint8).reshape(n, h, row_bytes)
    for i in range(n):
        dst = big[i * frame_stride: i * frame_stride + h * pitch].view(h, pitch)
        dst[:, :row_bytes] = src[i]
    torch.cuda.synchronize()
    det.saddles_batch_enqueue_ptr(big.data_ptr(), n, width, h, pitch, frame_stride,
                                  {"L8": _ffi.AGX_L8, "L16": _ffi.AGX_L16, "RGB8": _ffi.AGX_RGB8}[fmt])
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    for i in range(n):
        check_saddles(res[i], oracle.refined_saddle_points(host[i]), "%s %d pitch %d frame %d" % (fmt, width, pitch, i))


def test_plain_c_client_of_the_abi(oracle, tmp_path):
    """examples/c_client.c (C99, no Python / torch in the process) on a synthetic board frame:
    the same counts and first saddle as the oracle."""
    import subprocess
    from tests.test_abi_cpu import _build_c_client
    synth = synth_module()
    img = np.asarray(synth.render_frame(4, 640, 480)[0])
    raw = tmp_path / "frame.raw"
    img.tofile(raw)
    r = subprocess.run([_build_c_client(tmp_path), str(raw), "640", "480"], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    ref = oracle.refined_saddle_points(img)
    tags = oracle.detect(img)
    assert "%d saddles, %d tags" % (len(ref), len(tags)) in r.stdout, (r.stdout, len(ref), len(tags))
    assert "first saddle (%.3f, %.3f) k=%.5f" % (ref["x"][0], ref["y"][0], ref["k"][0]) in r.stdout, r.stdout


def test_plain_c_client_runs_a_batch_through_agx_detect_batch(oracle, tmp_path):
    """examples/c_client.c with five frames in one file: agx_detect_batch from plain C (no Python / torch in the process; the
    default thread count = agx_host_parallelism()) returns the oracle's tag count for every frame."""
    import subprocess
    from tests.test_abi_cpu import _build_c_client
    synth = synth_module()
    frames = np.stack([np.asarray(synth.render_frame(40 + i, 640, 480)[0]) for i in range(5)])
    raw = tmp_path / "frames.raw"
    frames.tofile(raw)
    r = subprocess.run([_build_c_client(tmp_path), str(raw), "640", "480", "5"], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    want = [len(oracle.detect(f)) for f in frames]
    assert "batch of 5 frames" in r.stdout and (":" + "".join(" %d" % n for n in want) + " tags") in r.stdout, (r.stdout, want)


def test_capacity_overflow_is_reported_not_truncated(oracle):
    import aprilgrid_rs_amd as A
    d = A.TagDetector("T36H11", None, device=0)
    d.set_limits(max_candidates=4096, max_clusters=256, max_saddles=64)
    img = load_image("EuRoC.png")
    with pytest.raises(A.AgxError) as e:
        d.refined_saddle_points(img)
    assert e.value.status == -3
    d.set_limits(0, 0, 0)
    check_saddles(d.refined_saddle_points(img, as_array=True), oracle.refined_saddle_points(img), "after reset")
    d.close()


def test_constants_match_oracle(det, oracle):
    w, cone, pmat = det.constants()
    assert bits_equal(w, oracle.blur_weights(1.5))
    op, ok = oracle.refine_constants(2)
    assert bits_equal(cone, ok)
    assert bits_equal(pmat, op)


def test_full_size_batch_properties(det, oracle):
    """BASELINE configuration size (1280x800), a 16-frame slice: per-frame results are
    independent of batch position and batch composition (frame i alone == frame i in the
    batch), and two spot frames match the oracle."""
    synth = synth_module()
    frames, _ = synth.render_batch(100, 16, 1280, 800, device="cuda")
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    host = frames.cpu().numpy()
    for i in (0, 15):
        check_saddles(res[i], oracle.refined_saddle_points(host[i]), "1280x800 frame %d" % i)
    for i in (3, 9):
        det.saddles_batch_enqueue(frames[i:i + 1].contiguous())
        single, _ = det.saddles_batch_fetch()
        assert single[0].tobytes() == res[i].tobytes()
    rev = torch_flip(frames)
    det.saddles_batch_enqueue(rev)
    res_rev, _ = det.saddles_batch_fetch()
    for i in range(16):
        assert res_rev[15 - i].tobytes() == res[i].tobytes()


def torch_flip(frames):
    import torch
    return torch.flip(frames, dims=[0]).contiguous()


# ---- AGX_LF32: the caller's own to_luma32f plane (any DynamicImage variant) ---------------------
def _luma_planes_like_image_crate(kind, base_u8, rng):
    """(pixel array of the variant, f32 luma plane, u8 luma plane) for DynamicImage variants the ABI
    does not take natively.  The plane arithmetic is what a Rust caller gets from img.to_luma32f() /
    img.to_luma8(); it is restated here only to have realistic planes -- the library and the oracle
    are both fed THE SAME planes, which is the point of AGX_LF32."""
    h, w = base_u8.shape
    if kind == "La8":  # luma + alpha: alpha dropped
        px = np.stack([base_u8, rng.integers(0, 256, (h, w), dtype=np.uint8)], -1)
        return px, (base_u8.astype(np.float32) / np.float32(255.0)), base_u8.copy()
    if kind == "Rgba8":
        rgb = np.stack([np.clip(base_u8.astype(np.int32) + d, 0, 255).astype(np.uint8) for d in (-2, 0, 3)], -1)
        px = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], -1)
        r32 = rgb.astype(np.uint32)
        l8 = ((2126 * r32[..., 0] + 7152 * r32[..., 1] + 722 * r32[..., 2]) // 10000).astype(np.uint8)
        return px, (l8.astype(np.float32) / np.float32(255.0)), l8
    if kind == "Rgb16":
        v = base_u8.astype(np.uint32) * 257
        rgb = np.stack([np.clip(v.astype(np.int64) + d, 0, 65535).astype(np.uint32) for d in (-300, 0, 500)], -1)
        r64 = rgb.astype(np.uint64)
        l16 = (2126 * r64[..., 0] + 7152 * r64[..., 1] + 722 * r64[..., 2]) // 10000
        return rgb.astype(np.uint16), (l16.astype(np.float32) / np.float32(65535.0)), ((l16 + 128) // 257).astype(np.uint8)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["La8", "Rgba8", "Rgb16"])
def test_lf32_planes_of_other_image_variants(det, oracle, kind):
    """La8 / Rgba8 / Rgb16 images go through the chain as their f32 luma plane, bit
    for bit like the oracle fed the same plane; detect_planes == oracle tail on (saddles, u8 luma)."""
    synth = synth_module()
    base = np.asarray(synth.render_frame(21, 640, 480)[0])
    _, f32, l8 = _luma_planes_like_image_crate(kind, base, np.random.default_rng(3))
    got = det.refined_saddle_points(f32, as_array=True)
    ref = check_frame(det, oracle, f32, 0, kind + " as LF32")
    check_saddles(got, ref, kind)
    tags = det.detect_planes(f32, l8)
    ref_tags = oracle.detect_tail(l8, ref)
    assert sorted(tags) == sorted(ref_tags) and len(tags) >= 30
    for t in tags:
        assert bits_equal(tags[t], ref_tags[t])
    import aprilgrid_rs_amd as A
    with pytest.raises(A.AgxError) as e:
        det.detect(f32)  # no u8 luma to derive from an f32 plane
    assert e.value.status == -2


def test_lf32_equals_the_fused_l8_path(det, oracle):
    """v / 255 handed over as a plane gives exactly what the fused L8 conversion gives (two different
    kernels instantiations: product-table path vs plain path), at an aligned and a ragged width, and
    as a device batch."""
    import torch
    synth = synth_module()
    for w in (640, 613):
        img = np.ascontiguousarray(np.asarray(synth.render_frame(33, 640, 400)[0])[:, :w])
        a = det.refined_saddle_points(img, as_array=True)
        f32 = img.astype(np.float32) / np.float32(255.0)
        b = det.refined_saddle_points(f32, as_array=True)
        assert a.tobytes() == b.tobytes() and len(a) > 100
        check_frame(det, oracle, f32, 0, "LF32 width %d" % w)
    fr, _ = synth.render_batch(5, 3, 320, 240, device="cuda")
    planes = (fr.to(torch.float32) / 255.0).contiguous()
    det.saddles_batch_enqueue(planes)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    host = planes.cpu().numpy()
    for i in range(3):
        check_saddles(res[i], check_frame(det, oracle, host[i], i, "LF32 batch frame %d" % i), "LF32 batch %d" % i)


def test_lf32_arbitrary_float_planes(det_resp, oracle):
    """Planes that no integer image produces (tiny, subnormal, irregular values): the LF32 kernel
    evaluates image_util.rs:100-104 literally, so even the in-register response is bit-exact."""
    rng = np.random.default_rng(11)
    a = rng.random((96, 132), dtype=np.float32)
    a[10:20, 10:60] *= np.float32(1e-20)
    a[30:40, 40:90] = np.float32(1e-41) * rng.integers(0, 50, (10, 50)).astype(np.float32)  # subnormals
    a[50:60, :] = rng.random((10, 132), dtype=np.float32) * np.float32(3e-19)
    # negative zeros: the reference's sums start from +0.0 (image_util.rs:146,190), so a region of -0.0 blurs
    # to +0.0 -- check_frame compares the blur plane bit for bit, signs of zero included
    a[70:84, 20:40] = np.float32(-0.0)
    a[86:90, :] = np.float32(-0.0)
    got = det_resp.refined_saddle_points(a, as_array=True)
    check_saddles(got, check_frame(det_resp, oracle, a, 0, "arbitrary f32 plane"), "arbitrary f32")
    blur = det_resp.debug_fetch(0, "blur", a.shape)
    assert not np.signbit(blur[74:80, 25:35]).any(), "a -0.0 region must blur to +0.0"


@pytest.mark.parametrize("fmt", ["L8", "L16", "RGB8"])
def test_detect_batch_equals_per_frame_detect(det, oracle, fmt):
    """agx_detect_batch (chain per chunk on the device, board search + decode on a pool of host
    threads): every frame's tag map equals detect() of that frame and the oracle's, with the frames
    uploaded by the call and with a device copy handed in; a second call reuses the pool."""
    synth = synth_module()
    n = 70  # more than one chunk with 4 threads (chunk = 32 frames)
    fr, gts = synth.render_batch(300, n, 320, 240, device="cuda", fmt=fmt)
    host = fr.cpu().numpy()
    if fmt == "L16":
        host = host.view(np.uint16)
    got = det.detect_batch(host, n_threads=4)
    got_dev = det.detect_batch(host, n_threads=4, device_frames=fr)
    assert len(got) == n
    n_tags = 0
    for i in range(n):
        one = det.detect(host[i]) if i % 7 == 0 else oracle.detect(host[i])
        assert sorted(got[i]) == sorted(one) == sorted(got_dev[i]), "frame %d" % i
        for t in one:
            assert bits_equal(got[i][t], one[t]) and bits_equal(got_dev[i][t], one[t])
        n_tags += len(one)
    assert n_tags > 20 * n


def test_detect_batch_single_frame_ignores_frame_stride(det, oracle):
    """with ONE frame the stride between frames carries no information (0 is a natural value); the
    upload must still cover the whole frame."""
    import ctypes as C
    synth = synth_module()
    fr, _ = synth.render_batch(77, 1, 320, 240, device="cuda")
    host = np.ascontiguousarray(fr.cpu().numpy()[0])
    ref = oracle.detect(host)
    assert len(ref) > 10
    out = np.zeros(256, np.dtype([("id", "u4"), ("xy", "f4", (8,))]))
    cnt = np.zeros(1, np.uint32)
    st = np.zeros(1, np.int32)
    for stride in (0, 100, 320 * 240):
        cnt[:] = 0
        rc = det._lib.agx_detect_batch(det._h, host.ctypes.data, None, 1, 320, 240, 320, stride, 0, out.ctypes.data, 256,
                                       cnt.ctypes.data, st.ctypes.data, 2)
        assert rc == 0 and st[0] == 0, (stride, rc)
        got = {int(t["id"]): t["xy"].reshape(4, 2) for t in out[: cnt[0]]}
        assert sorted(got) == sorted(ref), stride
        for t in ref:
            assert bits_equal(got[t], ref[t])


def test_detect_batch_reports_overflow_per_frame_without_raising(det):
    """one frame over `cap` tags must not throw the other frames' results away."""
    synth = synth_module()
    fr, _ = synth.render_batch(300, 6, 320, 240, device="cuda")
    host = fr.cpu().numpy()
    full = det.detect_batch(host, n_threads=2)
    res, status = det.detect_batch(host, n_threads=2, cap=max(len(t) for t in full) - 1, raise_on_overflow=False)
    over = [i for i in range(6) if status[i] != 0]
    assert over and all(res[i] is None for i in over)
    for i in range(6):
        if status[i] == 0:
            assert sorted(res[i]) == sorted(full[i])
    with pytest.raises(Exception):
        det.detect_batch(host, n_threads=2, cap=max(len(t) for t in full) - 1)
    with pytest.raises(Exception):  # a device tensor that is not the frames
        det.detect_batch(host, n_threads=2, device_frames=fr[:3])


def test_more_than_16384_saddles_per_frame(det, oracle):
    """the reference's Vec<Saddle> has no size limit.  Pure-noise frames of a few
    megapixels hold more saddles than the LDS sort takes (16384): they are ordered in global memory
    and come back complete, through the batch API and the single-frame API, equal to the oracle."""
    synth = synth_module()
    frames, _ = synth.render_batch(3, 2, 2048, 1536, device="cuda", pure_noise=True)
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    host = frames.cpu().numpy()
    refs = oracle_saddles_parallel(oracle, host, threads=2)
    for i in range(2):
        assert len(refs[i]) > 16384, len(refs[i])
        check_saddles(res[i], refs[i], "noise 2048x1536 frame %d" % i)
        c = det.debug_fetch(i, "counters")
        assert c["flags"] & 64, c  # emitted by the large-list path
    one = det.refined_saddle_points(host[1], as_array=True)
    check_saddles(one, refs[1], "single-frame API")
    assert det.detect(host[0]) == {}  # no board in noise; the tail runs on > 16384 saddles without overflow
    assert det.detect_batch(host, n_threads=2) == [{}, {}]  # the batch call grows its saddle staging likewise


def test_4k_pure_noise_frame(det, oracle):
    """A 3840x2160 pure-noise frame (about 57 000 saddles): what round 1 reported as AGX_ERR_CAPACITY."""
    synth = synth_module()
    frames, _ = synth.render_batch(8, 1, 3840, 2160, device="cuda", pure_noise=True)
    det.saddles_batch_enqueue(frames)
    res, status = det.saddles_batch_fetch()
    assert (status == 0).all()
    ref = oracle.refined_saddle_points(frames[0].cpu().numpy(), cap=1 << 18)
    assert len(ref) > 40000
    check_saddles(res[0], ref, "4K noise")
