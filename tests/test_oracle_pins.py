"""The oracle (oracle/agx_oracle.c) against everything the reference's own tests pin for this
path, plus implementation-independent ground truth.  CPU only."""
import numpy as np
import pytest

from tests.util import ALL_IMAGES, REFERENCE_TAG_COUNTS, bits_equal, check_saddles_against_ground_truth, load_image, synth_module
from oracle import oracle as O


def test_hessian_response_known_answer():
    """src/image_util.rs:270-293: a spike of 10 at (2,2) of a 5x5 zero image -> det = 400 > 0."""
    img = np.zeros((5, 5), np.float32)
    img[2, 2] = 10.0
    r = O.hessian_response(img)
    assert r[2, 2] == 400.0
    assert (r[0] == 0).all() and (r[-1] == 0).all() and (r[:, 0] == 0).all() and (r[:, -1] == 0).all()


def test_pixel_bfs_known_answer():
    """src/image_util.rs:295-316: two dark pixels -> cluster {(2,2),(2,3)}, visited := f32::MAX."""
    import ctypes as C
    img = np.full((5, 5), 100.0, np.float32)
    img[2, 2] = 10.0
    img[3, 2] = 10.0  # (x=2, y=3)
    out = np.zeros(32, np.uint32)
    n = O.lib().orc_pixel_bfs(img.ctypes.data, 5, 5, 2, 2, C.c_float(50.0), out.ctypes.data, 16)
    assert n == 2
    assert {(int(out[0]), int(out[1])), (int(out[2]), int(out[3]))} == {(2, 2), (2, 3)}
    assert img[2, 2] == np.finfo(np.float32).max


def test_math_util_known_answers():
    """src/math_util.rs:39-89."""
    import ctypes as C
    lib = O.lib()
    x, y = C.c_float(), C.c_float()
    lib.orc_find_xy(1.0, 1.0, -2.0, 1.0, -1.0, 0.0, C.addressof(x), C.addressof(y))
    assert abs(x.value - 1.0) < 1e-6 and abs(y.value - 1.0) < 1e-6
    td = lib.orc_theta_distance_degree
    for (a, b, e) in [(0, 0, 0), (0, 90, 90), (0, 45, 45), (0, 180, 0), (10, 20, 10)]:
        assert abs(td(a, b) - e) < 1e-6
    assert abs(lib.orc_cross(1, 0, 0, 1) - 1) < 1e-6 and abs(lib.orc_cross(0, 1, 1, 0) + 1) < 1e-6
    assert abs(lib.orc_dot(1, 0, 0, 1)) < 1e-6 and abs(lib.orc_dot(1, 0, 1, 1) - 1) < 1e-6
    assert abs(lib.orc_angle_degree(1, 0, 0, 1) - 90) < 1e-5 and abs(lib.orc_angle_degree(1, 0, 1, 1) - 45) < 1e-5


def test_is_valid_quad_known_answers():
    """src/saddle.rs:91-173."""
    lib = O.lib()

    def sd(x, y, th):
        return O.Saddle(x, y, 0.0, th, 0.0)
    d0, s1, d1 = sd(10, 0, 0), sd(10, 10, 0), sd(0, 10, 0)
    import ctypes as C
    f = lib.orc_is_valid_quad
    f.argtypes = [C.c_void_p] * 4
    q = lambda s0: f(C.addressof(s0), C.addressof(d0), C.addressof(s1), C.addressof(d1))
    assert q(sd(0, 0, 45.0)) == 0
    assert q(sd(0, 0, 135.0)) == 1


def test_tag_affine_known_answer():
    """src/image_util.rs:259-268: last row [0,0,1] is implicit; the fit maps the corner grid."""
    corners = np.array([0, 0, 0, 10, 10, 10, 10, 0], np.float32)
    h = np.zeros(6, np.float32)
    O.lib().orc_tag_affine(corners.ctypes.data, 10, 0.0, h.ctypes.data)
    # source (0,0)->(0,0), (0,9)->(0,10), (9,9)->(10,10), (9,0)->(10,0): scale 10/9
    assert np.allclose(h, [10 / 9, 0, 0, 0, 10 / 9, 0], atol=1e-5)


def test_rotate_bits_is_a_quarter_turn():
    """src/detector.rs:124-140: four applications are the identity."""
    lib = O.lib()
    for bits in (0x1, 0xD5D628584, 0xFFFFFFFFF, 0x123456789):
        b = bits
        for _ in range(4):
            b = lib.orc_rotate_bits(b, 6)
        assert b == bits


@pytest.mark.parametrize("name,expected", REFERENCE_TAG_COUNTS)
def test_reference_tag_counts(name, expected):
    """The reference's only end-to-end assertions, tests/test_detector.rs:26-32."""
    assert len(O.detect(load_image(name))) == expected


def test_blur_matches_scipy_structure():
    """Independent cross-check of the restated blur: scipy's correlate1d with the same 7 taps and
    edge replication agrees to float rounding (it accumulates in double, so not bitwise)."""
    from scipy import ndimage
    rng = np.random.default_rng(1)
    img = rng.random((40, 50)).astype(np.float32)
    w = O.blur_weights(1.5)
    ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), w.astype(np.float64), axis=1, mode="nearest"),
                              w.astype(np.float64), axis=0, mode="nearest")
    assert np.max(np.abs(O.gaussian_blur_f32(img) - ref)) < 2e-6
    assert len(w) == 7 and abs(float(w.sum()) - 1.0) < 1e-6 and bits_equal(w, w[::-1])


def test_pinv_is_the_exact_pseudo_inverse():
    """rochade_refine's 25x6 matrix (src/detector.rs:208-237) equals the exact rational
    pseudo-inverse rounded once to f32."""
    from fractions import Fraction as F
    A = [[F(x * x), F(x * y), F(y * y), F(x), F(y), F(1)] for y in range(-2, 3) for x in range(-2, 3)]
    N = [[sum(A[k][i] * A[k][j] for k in range(25)) for j in range(6)] for i in range(6)]
    M = [row[:] + [F(int(i == j)) for j in range(6)] for i, row in enumerate(N)]
    for c in range(6):
        p = next(r for r in range(c, 6) if M[r][c] != 0)
        M[c], M[p] = M[p], M[c]
        d = M[c][c]
        M[c] = [v / d for v in M[c]]
        for r in range(6):
            if r != c:
                f = M[r][c]
                M[r] = [a - f * b for a, b in zip(M[r], M[c])]
    inv = [row[6:] for row in M]
    exact = np.array([[float(sum(inv[j][k] * A[i][k] for k in range(6))) for j in range(6)] for i in range(25)])
    pm, cone = O.refine_constants(2)
    assert bits_equal(pm, exact.astype(np.float32))
    assert abs(float(cone.sum()) - 1.0) < 1e-6 and cone.min() > 0


def test_synthetic_ground_truth_ids_and_corners():
    """Implementation-independent evidence: frames rendered from known homographies.  Every tag
    the oracle reports carries the id the renderer drew there, and its corners lie within
    0.5 px of the projected ground truth."""
    synth = synth_module()
    total = 0
    for i in (0, 1, 3):
        frame, gt = synth.render_frame(i, 640, 400)
        tags = O.detect(frame.numpy())
        assert len(tags) >= 30
        for tid, c in tags.items():
            assert tid in gt
            for p in c:
                assert np.min(np.hypot(*(gt[tid] - p).T)) < 0.5
        total += len(tags)
    assert total >= 100


def test_synthetic_ground_truth_saddles():
    """The hot path's output against the renderer's own truth: every drawn tag corner inside the frame has a refined saddle of
    the oracle within 0.3 px (the GPU suite asserts the same of the HIP chain on all 256 bench frames)."""
    synth = synth_module()
    dist = []
    for i in (0, 5, 17):
        frame, gt = synth.render_frame(i, 1280, 800)
        s = O.refined_saddle_points(frame.numpy())
        dist.append(check_saddles_against_ground_truth(np.stack([s["x"], s["y"]], 1), gt, 1280, 800, "frame %d" % i))
    dist = np.concatenate(dist)
    assert np.median(dist) < 0.05 and np.percentile(dist, 99) < 0.15, (np.median(dist), np.percentile(dist, 99))


def test_flat_image_gives_nothing():
    img = np.full((32, 48), 200, np.uint8)
    assert len(O.refined_saddle_points(img)) == 0 and O.detect(img) == {}


def test_luma_conversions():
    """image 0.25.9 to_luma32f / to_luma8 as restated."""
    g16 = np.array([[0, 128, 257, 65535]], np.uint16)
    assert np.array_equal(O.luma_u8(g16)[0], [0, 0, 1, 255])
    assert bits_equal(O.luma_f32(g16)[0], (g16[0].astype(np.float32) / np.float32(65535)))
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], np.uint8)
    assert np.array_equal(O.luma_u8(rgb)[0], [54, 182, 18, 255])


@pytest.mark.parametrize("name", ALL_IMAGES)
def test_the_images_own_geometry(name):
    """What the reference's images say whoever reads them (tests/grid_pins.py): the ids are exactly 0 .. N-1, every tag is a convex
    quad wound like every other, and all corners of a board lie within 1 px of ONE regular planar grid (pitch 1.3 tag edges, ids
    row by row) seen through a smooth 14-parameter camera.  The reference asserts len() only (tests/test_detector.rs:21-32)."""
    from tests import grid_pins
    img = load_image(name)
    tags = O.detect(img)
    r = grid_pins.check_image(name, tags, img.shape, dict(REFERENCE_TAG_COUNTS).get(name))
    assert r["camera_max_px"] < 0.75 and r["camera_rms_px"] < 0.25, r  # (measured: max 0.12 .. 0.52, rms 0.05 .. 0.18)


def test_the_geometry_pin_notices_a_wrong_detector():
    """The check of the check: one corner moved by 2 px, two tags' ids exchanged, one tag's corners in rotated order, a missing
    tag -- each is refused."""
    from tests import grid_pins
    img = load_image("EuRoC.png")
    tags = O.detect(img)
    grid_pins.check_image("EuRoC.png", tags, img.shape, 36)

    def refused(bad):
        try:
            grid_pins.check_image("EuRoC.png", bad, img.shape, 36)
        except AssertionError:
            return True
        return False

    moved = {t: c.copy() for t, c in tags.items()}
    moved[17][2] += np.float32(2.0)
    assert refused(moved)
    swapped = dict(tags)
    swapped[8], swapped[9] = tags[9], tags[8]
    assert refused(swapped)
    rotated = dict(tags)
    rotated[20] = np.roll(tags[20], 1, axis=0)
    assert refused(rotated)
    mirrored = dict(tags)
    mirrored[3] = tags[3][::-1].copy()
    assert refused(mirrored)
    missing = {t: c for t, c in tags.items() if t != 35}
    assert refused(missing)
