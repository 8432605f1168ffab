"""Option "device_tail": agx_detect_batch's board search + decode as a HIP kernel (csrc/tail_kernels.hip) behind the chain.

The kernel must return exactly what the host tail returns for every frame it keeps (ids, corners bit for bit, insertion
order, per-frame status), and hand every other frame back to the host tail -- so the option never changes a result.  The
host tail itself is pinned against the oracle and the reference's tag counts in tests/test_gpu_parity.py; here the two
tails are compared with each other on the same saddles, and with the oracle where the frames are small."""
import numpy as np
import pytest

from util import ALL_IMAGES, REFERENCE_TAG_COUNTS, bits_equal, check_tags, load_image, oracle_detect_parallel, synth_module

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="module")
def pair():
    import aprilgrid_rs_amd as A
    host = A.TagDetector("t36h11", None, device=0)
    dev = A.TagDetector("t36h11", None, device=0)
    assert dev.get_option("device_tail") == -1  # the default: by the batch's size, where this process's atan2f is glibc's routine
    host.set_option("device_tail", 0)
    dev.set_option("device_tail", 1)
    assert dev.get_option("device_tail") == 1 and host.get_option("device_tail") == 0
    yield host, dev
    host.close()
    dev.close()


def same_results(host, dev, frames, cap=128, device_frames=None, threads=4):
    """Both detectors over the same batch -> (tags per frame, frames the kernel handed back)."""
    rc_h, out_h, cnt_h, st_h = host.detect_batch_raw(frames, n_threads=threads, cap=cap, device_frames=device_frames)
    rc_d, out_d, cnt_d, st_d = dev.detect_batch_raw(frames, n_threads=threads, cap=cap, device_frames=device_frames)
    assert rc_h == rc_d, (rc_h, rc_d)
    assert np.array_equal(st_h, st_d), (st_h, st_d)
    assert np.array_equal(cnt_h, cnt_d), np.nonzero(cnt_h != cnt_d)
    for f in range(len(frames)):
        if st_h[f] == 0:
            assert out_h[f, : cnt_h[f]].tobytes() == out_d[f, : cnt_d[f]].tobytes(), "frame %d" % f
    assert dev.get_option("last_device_tail_frames") == len(frames)
    return cnt_h, dev.get_option("last_device_tail_fallbacks")


def test_the_benchmarks_frames_keep_their_tags(pair, oracle):
    """configs[1]'s 256 frames (1280 x 800, one board each): the device tail's tags are the host tail's AND the oracle's detect
    on every one of the 256 frames (ids equal, corners bit for bit), and the kernel keeps nearly every frame (a frame goes back
    only for an angle within 1e-4 degrees of 60 / 120 in a quad that otherwise passes)."""
    host, dev = pair
    synth = synth_module()
    fr, _ = synth.render_batch(0, 256, 1280, 800, device="cuda")
    frames = fr.cpu().numpy()
    counts, back = same_results(host, dev, frames, threads=0)
    assert counts.mean() > 30
    assert back <= 8, back
    got = dev.detect_batch(frames, n_threads=0)
    assert dev.get_option("last_device_tail_frames") == 256
    refs = oracle_detect_parallel(oracle, frames, threads=16)
    for i in range(256):
        check_tags(got[i], refs[i], "bench frame %d" % i)
    assert sum(len(r) for r in refs) == int(counts.sum())
    # configs[2]'s last shard, from a device copy as well
    fr, _ = synth.render_batch(1792, 64, 1280, 800, device="cuda")
    counts, back = same_results(host, dev, fr.cpu().numpy(), device_frames=fr, threads=0)
    assert counts.mean() > 30 and back <= 4


def test_the_benchmarks_frames_against_their_ground_truth(pair):
    """No oracle, no second implementation: configs[1]'s 256 frames are rendered from known homographies (synth.py), so every
    tag the HIP path reports -- chain + device tail, and chain + host tail -- must carry an id the renderer drew, each of its
    four corners within 0.3 px of a DIFFERENT projected corner of that very tag, and nearly every drawn tag must be found
    (the renderer adds blur and noise; measured: 9 055 of 9 216 tags = 35.4 of 36 per frame, corner error median 0.032 px,
    99th percentile 0.10 px, maximum 0.24 px)."""
    host, dev = pair
    synth = synth_module()
    fr, gts = synth.render_batch(0, 256, 1280, 800, device="cuda")
    frames = fr.cpu().numpy()
    for det in (dev, host):
        got = det.detect_batch(frames, n_threads=0, device_frames=fr)
        found, errs = 0, []
        for i in range(256):
            for tid, c in got[i].items():
                assert tid in gts[i], "frame %d: tag %d was never drawn" % (i, tid)
                d = np.hypot(c[:, None, 0] - gts[i][tid][None, :, 0], c[:, None, 1] - gts[i][tid][None, :, 1])  # [detected, drawn]
                nearest = d.argmin(axis=1)
                assert sorted(nearest) == [0, 1, 2, 3], "frame %d tag %d: corners %s" % (i, tid, nearest)
                assert d.min(axis=1).max() < 0.3, "frame %d tag %d: a corner %.2f px from the ground truth" % (i, tid, d.min(axis=1).max())
                errs.extend(d.min(axis=1))
            found += len(got[i])
        assert found >= 0.95 * 36 * 256, found
        assert np.median(errs) < 0.05 and np.percentile(errs, 99) < 0.15, (np.median(errs), np.percentile(errs, 99))


@pytest.mark.parametrize("fmt", ["L8", "L16", "RGB8"])
def test_formats_and_the_oracle(pair, oracle, fmt):
    """Small frames in the three formats (the decode reads the device's own to_luma8 for L16 / RGB8): equal to the host tail
    and, frame by frame, to the oracle's detect."""
    host, dev = pair
    synth = synth_module()
    n = 70
    fr, _ = synth.render_batch(300, n, 320, 240, device="cuda", fmt=fmt)
    frames = fr.cpu().numpy()
    if fmt == "L16":
        frames = frames.view(np.uint16)
    same_results(host, dev, frames)
    got = dev.detect_batch(frames, n_threads=4)
    n_tags = 0
    for i in range(0, n, 3):
        ref = oracle.detect(frames[i])
        assert sorted(got[i]) == sorted(ref), i
        for t in ref:
            assert bits_equal(got[i][t], ref[t])
        n_tags += len(ref)
    assert n_tags > 20 * (n // 3)


@pytest.mark.parametrize("name", ALL_IMAGES)
def test_fixture_images(pair, name):
    """The reference's images (two boards, 66 tags, 16-bit, RGB): a batch of one through the device tail gives detect()'s map
    -- and the reference's own tag counts (tests/test_detector.rs:26-32)."""
    host, dev = pair
    img = load_image(name)
    counts, _ = same_results(host, dev, img[None], cap=128)
    one = host.detect(img)
    got = dev.detect_batch(img[None], n_threads=2)[0]
    assert list(got) == list(one)
    for t in one:
        assert bits_equal(got[t], one[t])
    expected = dict(REFERENCE_TAG_COUNTS).get(name)
    if expected is not None:
        assert counts[0] == expected


@pytest.mark.parametrize("family", ["T16H5", "T25H7", "T25H9", "T36H11B1"])
def test_other_families(oracle, family):
    import aprilgrid_rs_amd as A
    synth = synth_module()
    d = A.TagDetector(family, None, device=0)
    d.set_option("device_tail", 1)
    imgs = np.stack([synth.render_frame(21 + i, 800, 600, spec=synth.BoardSpec(rows=5, cols=5), family=family)[0].numpy() for i in range(4)])
    got = d.detect_batch(imgs, n_threads=2)
    assert d.get_option("last_device_tail_fallbacks") <= 1
    for i in range(4):
        ref = oracle.detect(imgs[i], family=family)
        assert sorted(got[i]) == sorted(ref) and len(ref) >= 20
        for t in ref:
            assert bits_equal(got[i][t], ref[t])
    d.close()


@pytest.mark.parametrize("boards", [1, 3])
def test_max_num_of_boards(boards):
    import aprilgrid_rs_amd as A
    p = A.DetectorParams.default_params()
    p.max_num_of_boards = boards
    host = A.TagDetector("t36h11", p, device=0)
    dev = A.TagDetector("t36h11", p, device=0)
    host.set_option("device_tail", 0)
    dev.set_option("device_tail", 1)
    two = load_image("two_boards.png")
    counts, _ = same_results(host, dev, two[None])
    assert counts[0] == (36 if boards == 1 else 72)
    synth = synth_module()
    fr, _ = synth.render_batch(40, 24, 640, 480, device="cuda")
    same_results(host, dev, fr.cpu().numpy())
    host.close()
    dev.close()


def test_frames_the_kernel_cannot_take(pair):
    """More saddles than the kernel's lists hold (pure noise), no saddles at all, and more tags than the caller's capacity:
    the first go to the host tail, the second need nothing, the third report AGX_ERR_CAPACITY -- as without the option."""
    host, dev = pair
    synth = synth_module()
    noise, _ = synth.render_batch(3, 3, 640, 480, device="cuda", pure_noise=True)
    flat = np.full((2, 480, 640), 128, np.uint8)
    board, _ = synth.render_batch(300, 3, 640, 480, device="cuda")
    frames = np.concatenate([noise.cpu().numpy(), flat, board.cpu().numpy()])
    counts, back = same_results(host, dev, frames)
    assert back == 3 and list(counts[:5]) == [0] * 5 and counts[5:].min() > 10
    cap = int(counts[5:].max()) - 1
    rc_h, _, cnt_h, st_h = host.detect_batch_raw(frames, n_threads=2, cap=cap)
    rc_d, _, cnt_d, st_d = dev.detect_batch_raw(frames, n_threads=2, cap=cap)
    assert rc_h == rc_d != 0 and np.array_equal(st_h, st_d) and np.array_equal(cnt_h, cnt_d) and (st_d != 0).any()


def test_frames_with_more_than_512_saddles_stay_on_the_device(pair):
    """Three board frames side by side: 600 or so saddles, three boards for the two rounds of the search -- within the kernel's 1024."""
    host, dev = pair
    synth = synth_module()
    fr, _ = synth.render_batch(500, 24, 640, 480, device="cuda")
    a = fr.cpu().numpy()
    wide = np.ascontiguousarray(np.concatenate([a[0::3], a[1::3], a[2::3]], axis=2))  # 8 frames of 1920 x 480
    n_saddles = [len(host.refined_saddle_points(w, as_array=True)) for w in wide]
    assert max(n_saddles) > 512, n_saddles
    counts, back = same_results(host, dev, wide, cap=128)
    assert back <= 1 and counts.min() >= 30, (back, counts, n_saddles)  # (the boards carry the same 36 ids: the later board's corners replace the earlier's)


def test_boards_larger_than_the_kernels_lists(pair, oracle):
    """A 10 x 10 board has more cells than one board of the kernel holds (128, within 12 cells of the seed): the frame is handed
    to the host tail -- the same 100 tags as without the option, and as the oracle's.  A 7 x 9 board still fits."""
    host, dev = pair
    synth = synth_module()
    big = np.stack([synth.render_frame(5 + i, 1280, 800, spec=synth.BoardSpec(rows=10, cols=10))[0].numpy() for i in range(3)])
    counts, back = same_results(host, dev, big, cap=256)
    assert back == 3 and counts.min() >= 90, (back, counts)
    got = dev.detect_batch(big[:1], n_threads=2, cap=256)[0]
    ref = oracle.detect(big[0])
    assert sorted(got) == sorted(ref) and all(bits_equal(got[t], ref[t]) for t in ref)
    mid = np.stack([synth.render_frame(9 + i, 1280, 800, spec=synth.BoardSpec(rows=7, cols=9))[0].numpy() for i in range(3)])
    counts, back = same_results(host, dev, mid, cap=256)
    assert back <= 1 and counts.min() >= 55, (back, counts)


@pytest.mark.parametrize("w,h", [(2, 2), (17, 5), (64, 48), (1280, 64), (192, 128), (333, 217)])
def test_odd_and_tiny_geometries(pair, w, h):
    """Frames with no, few or many saddles in sizes that are nobody's default (odd widths take the blur kernel's unaligned form):
    checkerboards and noise (saddles without tags), small boards."""
    host, dev = pair
    synth = synth_module()
    if w >= 192 and h >= 128:
        frames = synth.render_batch(900, 48, w, h, device="cuda")[0].cpu().numpy()
    else:
        frames = np.random.default_rng(w * 1000 + h).integers(0, 256, (48, h, w), dtype=np.uint8)
        frames[::2] = (np.indices((h, w)).sum(0) // 4 % 2 * 200 + 20).astype(np.uint8)
    same_results(host, dev, frames, cap=64)


def test_the_hand_back_path_on_purpose(pair, oracle):
    """TAIL_UNCERTAIN protects exactness: a frame in which a white-block angle falls inside the kernel's guard band takes the host
    tail.  The band is 1e-4 degrees, so the suite's frames never go that way by themselves; option "tail_debug_band" (thousandths
    of a degree) widens it to 0.05 degrees for this test (measured on these 96 frames: 1e-3 degrees -> 2 frames, 0.01 -> 13, 0.05 -> 38, 0.3 -> 87, 1.5 -> all): a known share of frames is handed back as uncertain, the statistics say
    so, and every frame's tags are still the host tail's and the oracle's."""
    host, dev = pair
    synth = synth_module()
    fr, _ = synth.render_batch(0, 96, 1280, 800, device="cuda")
    frames = fr.cpu().numpy()
    same_results(host, dev, frames, threads=0)
    assert dev.get_option("last_device_tail_uncertain") <= 2
    dev.set_option("tail_debug_band", 50)
    try:
        assert dev.get_option("tail_debug_band") == 50
        counts, back = same_results(host, dev, frames, threads=0)
        unc = dev.get_option("last_device_tail_uncertain")
        assert 15 <= unc <= back <= 70, (unc, back)  # some frames, not all of them (measured: 38 of 96)
        got = dev.detect_batch(frames, n_threads=0)
        assert dev.get_option("last_device_tail_uncertain") == unc  # deterministic
        refs = oracle_detect_parallel(oracle, frames[::4], threads=8)
        for i, ref in zip(range(0, 96, 4), refs):
            check_tags(got[i], ref, "frame %d with the wide band" % i)
        # L16 frames handed back decode from the host's own to_luma8 of the caller's frame
        fr16, _ = synth.render_batch(300, 40, 320, 240, device="cuda", fmt="L16")
        f16 = fr16.cpu().numpy().view(np.uint16)
        same_results(host, dev, f16, threads=2)
        assert dev.get_option("last_device_tail_uncertain") > 0
    finally:
        dev.set_option("tail_debug_band", 0)
    same_results(host, dev, frames[:32], threads=0)
    assert dev.get_option("last_device_tail_uncertain") <= 1


_MANY_CHUNKS = """
import sys, numpy as np, aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
threads = [int(t) for t in sys.argv[1].split(",")]
fr = synth.render_batch(1200, 96, 320, 240, device="cuda")[0].cpu().numpy()
big = np.concatenate([fr] * 54)[:5130]  # six chunks of up to 1024 frames: the staging slots are refilled three times
host = A.TagDetector("t36h11", None, device=0); host.set_option("device_tail", 0)
dev = A.TagDetector("t36h11", None, device=0); dev.set_option("device_tail", 1)
rc_h, out_h, cnt_h, st_h = host.detect_batch_raw(big[:96], n_threads=4, cap=64)
assert rc_h == 0
for t in threads:
    for rep in range(2):
        rc_d, out_d, cnt_d, st_d = dev.detect_batch_raw(big, n_threads=t, cap=64)
        assert rc_d == 0 and dev.get_option("last_device_tail_frames") == len(big), (t, rc_d)
        for f in range(len(big)):
            g = f % 96
            assert cnt_d[f] == cnt_h[g] and out_d[f, : cnt_d[f]].tobytes() == out_h[g, : cnt_h[g]].tobytes(), (t, f)
    print("threads", t, "ok", flush=True)
"""


def test_more_chunks_than_staging_slots_on_one_to_six_host_threads():
    """Calls of six chunks (> AGX_UPLOAD_STREAMS = 3: the staging slots are refilled while earlier chunks' parts may still be
    queued) on pools of 1 .. 6 and 8 threads.  An uploader task never waits for another task -- a pool of <= 6 threads used to
    deadlock here -- so the run sits under a timeout: a hang fails the test instead of stalling the suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _MANY_CHUNKS, "1,2,3,4,5,6,8"], capture_output=True, text=True, timeout=420, cwd=root)
    assert r.returncode == 0 and "threads 8 ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("threads", [1, 2, 3])
def test_several_chunks_on_few_host_threads(pair, threads):
    """A call of more than two 1024-frame chunks: the chunks' uploads go up in parts on the pool's threads, in order, under the
    kernels of the chunk before -- also when the pool is a single thread (the parts then simply follow one another)."""
    host, dev = pair
    synth = synth_module()
    fr = synth.render_batch(1200, 96, 320, 240, device="cuda")[0].cpu().numpy()
    big = np.concatenate([fr] * 23)[:2150]
    rc_h, out_h, cnt_h, st_h = host.detect_batch_raw(big[:96], n_threads=4, cap=64)
    rc_d, out_d, cnt_d, st_d = dev.detect_batch_raw(big, n_threads=threads, cap=64)
    assert rc_d == rc_h == 0 and dev.get_option("last_device_tail_frames") == len(big)
    for f in range(len(big)):
        g = f % 96
        assert cnt_d[f] == cnt_h[g] and out_d[f, : cnt_d[f]].tobytes() == out_h[g, : cnt_h[g]].tobytes(), f


def test_a_board_that_repeats_its_tags(pair):
    """HashMap::insert (detector.rs:520): a tag decoded from several quads of one board keeps the place of its first insertion
    and the corners of its last -- a board drawn with the same five tags over and over."""
    host, dev = pair
    synth = synth_module()

    class Repeating(synth.BoardSpec):
        def tag_id(self, ix, iy_from_top):
            return (ix + 2 * iy_from_top) % 5

    imgs = np.stack([synth.render_frame(40 + i, 960, 720, spec=Repeating(rows=6, cols=6))[0].numpy() for i in range(6)])
    counts, _ = same_results(host, dev, imgs, cap=64)
    assert list(counts) == [5] * 6, counts


def test_several_detectors_in_threads(pair):
    """The reference's detect(&self) may be called from any number of threads; here that is a handle per thread.
    Three threads, a detector each, the device tail's kernels side by side on the GPU: every call gives the host tail's tags."""
    import threading
    import aprilgrid_rs_amd as A
    host, _ = pair
    synth = synth_module()
    fr, _ = synth.render_batch(700, 96, 640, 480, device="cuda")
    frames = fr.cpu().numpy()
    rc0, out0, cnt0, st0 = host.detect_batch_raw(frames, n_threads=4, cap=64)
    bad = {}

    def work(k):
        d = A.TagDetector("t36h11", None, device=0)
        d.set_option("device_tail", 1)
        n_bad = 0
        for _ in range(4):
            rc, out, cnt, st = d.detect_batch_raw(frames, n_threads=2, cap=64)
            ok = rc == rc0 and np.array_equal(cnt, cnt0) and all(out[f, : cnt[f]].tobytes() == out0[f, : cnt0[f]].tobytes() for f in range(len(frames)))
            n_bad += not ok
        bad[k] = n_bad
        d.close()

    threads = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert bad == {0: 0, 1: 0, 2: 0}, bad


def test_left_to_itself_a_call_chooses_by_its_size():
    """Option -1 (the default): the device tail's launch costs its slowest frame (2 .. 5 ms) whatever the batch, the host tail ~0.9 ms
    per frame and thread -- a call of fewer than four frames per host thread keeps the host tail, a larger one takes the device's."""
    import aprilgrid_rs_amd as A
    synth = synth_module()
    d = A.TagDetector("t36h11", None, device=0)
    fr, _ = synth.render_batch(300, 64, 320, 240, device="cuda")
    frames = fr.cpu().numpy()
    small = d.detect_batch(frames[:12], n_threads=4)
    assert d.get_option("last_device_tail_frames") == 0
    large = d.detect_batch(frames, n_threads=4)
    assert d.get_option("last_device_tail_frames") == 64 and d.get_option("device_tail") == -1
    for i in range(12):
        assert list(small[i]) == list(large[i]) and all(bits_equal(small[i][t], large[i][t]) for t in small[i])
    d.close()


def test_the_option_is_refused_where_libm_differs():
    """The kernel evaluates atan2f by glibc's routine; the option checks the host's atan2f against it first (here: equal).  A process
    whose libm differs (pretended: AGX_DEBUG_LIBM_MISMATCH=1) is refused the option with AGX_ERR_STATE, and the default quietly
    stays on the host tail -- with the same tags."""
    import ctypes as C
    import os
    import subprocess
    import sys
    import aprilgrid_rs_amd as A
    bad = C.c_uint64(1)
    assert A._ffi.lib().agx_debug_libm_atan2f_check(1 << 22, 7, C.byref(bad)) == 0 and bad.value == 0
    code = """
import numpy as np, torch, aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
d = A.TagDetector("t36h11", None, device=0)
try:
    d.set_option("device_tail", 1)
    raise SystemExit("not refused")
except A.AgxError as e:
    assert e.status == A._ffi.AGX_ERR_STATE and "atan2f" in str(e), e
fr, _ = synth.render_batch(300, 64, 320, 240, device="cuda")
d.set_option("device_tail", -1)
tags = d.detect_batch(fr.cpu().numpy(), n_threads=2)
assert d.get_option("device_tail") == -1 and d.get_option("last_device_tail_frames") == 0, (d.get_option("device_tail"), d.get_option("last_device_tail_frames"))
assert sum(len(t) for t in tags) > 600, [len(t) for t in tags]
print("refused, host tail:", [len(t) for t in tags])
"""
    env = dict(os.environ, AGX_DEBUG_LIBM_MISMATCH="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "refused, host tail" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
