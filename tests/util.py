"""Shared helpers of the test-suite."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# reference tests/test_detector.rs:26-32 : (file, expected number of tags)
REFERENCE_TAG_COUNTS = [
    ("iphone.png", 66), ("EuRoC.png", 36), ("TUM_VI.png", 36), ("right.png", 36), ("r45.png", 36),
    ("top.png", 36), ("two_boards.png", 72),
]
ALL_IMAGES = [n for n, _ in REFERENCE_TAG_COUNTS] + ["top_right.png", "1520525725372653511.png"]


def load_image(name):
    """Fixture image as the numpy stand-in of the reference's DynamicImage (L8 / L16 / RGB8)."""
    from PIL import Image
    im = Image.open(os.path.join(GOLDEN, "images", name))
    a = np.array(im)
    if a.dtype == np.int32:  # PIL mode I for 16-bit
        a = a.astype(np.uint16)
    assert a.dtype in (np.uint8, np.uint16), (name, a.dtype)
    return np.ascontiguousarray(a)


def bits_equal(a, b):
    """Bitwise equality of two float32 arrays (so that -0.0 != +0.0 and NaNs compare)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def synth_module():
    import aprilgrid_rs_amd  # noqa: F401  (registers the package)
    from aprilgrid_rs_amd import synth
    return synth


ANGLE_TOL_DEG = 1e-3  # |theta|, |phi| difference allowed, degrees (observed ~1e-5)


def check_saddles(gpu, ref, what=""):
    assert len(gpu) == len(ref), "%s: %d saddles vs oracle %d" % (what, len(gpu), len(ref))
    for f in ("x", "y", "k"):
        assert bits_equal(gpu[f], ref[f]), "%s: field %s differs" % (what, f)
    for f in ("theta", "phi"):
        if len(ref):
            assert np.max(np.abs(gpu[f] - ref[f])) <= ANGLE_TOL_DEG, (what, f)


def check_frame(det, O, img, frame=0, what="", params=None):
    """Compare every intermediate product of `frame` of the detector's last batch with the oracle:
    blur plane, the response recomputed from it (separate kernel), per-frame min, cluster table,
    unfiltered refine output -- and, when the detector runs with store_response, the response
    the blur kernel itself evaluated in registers."""
    h, w = img.shape[:2]
    ref, d = O.refined_saddle_points(img, params=params, debug=True)
    assert bits_equal(det.debug_fetch(frame, "blur", (h, w)), d["blur"]), what + ": blur plane"
    assert bits_equal(det.debug_fetch(frame, "resp_recomputed", (h, w)), d["resp"]), what + ": response plane (recomputed)"
    if det.get_option("store_response"):
        assert bits_equal(det.debug_fetch(frame, "resp", (h, w)), d["resp"]), what + ": response plane (K1 registers)"
    assert bits_equal(np.float32(det.debug_fetch(frame, "min")), np.float32(d["min_resp"])), what + ": min"
    c = det.debug_fetch(frame, "centers")
    assert len(c) == len(d["centers"]), what + ": cluster count %d vs %d" % (len(c), len(d["centers"]))
    assert np.array_equal(c["first_index"], d["first_index"]), what + ": cluster first pixels"
    assert np.array_equal(c["size"], d["sizes"]), what + ": cluster sizes"
    assert bits_equal(c["cx"], d["centers"][:, 0]) and bits_equal(c["cy"], d["centers"][:, 1]), what + ": centroids"
    check_saddles(det.debug_fetch(frame, "refined"), d["refined"], what + " (unfiltered)")
    # informational flags: AGX_FRAME_CENTROID_INEXACT (8) only where a coordinate sum can reach 2^24 (a cluster
    # refined twice -- a race between k_refine's workgroups that round 2 fixed -- used to raise it spuriously)
    flags = det.debug_fetch(frame, "counters")["flags"]
    assert not (flags & 7), what + ": overflow flags %d" % flags
    if len(d["sizes"]) == 0 or float(d["sizes"].max()) * max(h, w) < 2.0 ** 24:
        assert not (flags & 8), what + ": AGX_FRAME_CENTROID_INEXACT set without cause"
    return ref


def oracle_saddles_parallel(O, frames_host, threads=8, params=None):
    """Oracle saddle lists of many frames (ctypes releases the GIL: frame-parallel)."""
    from concurrent.futures import ThreadPoolExecutor
    O.lib()
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda f: O.refined_saddle_points(f, params=params), frames_host))


def oracle_detect_parallel(O, frames_host, threads=8, family="T36H11", params=None):
    """Oracle detect() maps of many frames (ctypes releases the GIL: frame-parallel)."""
    from concurrent.futures import ThreadPoolExecutor
    O.lib()
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda f: O.detect(f, family=family, params=params), frames_host))


def check_tags(got, ref, what=""):
    """One frame's {id: 4x2 corners}: the oracle's ids, the oracle's corners bit for bit."""
    assert sorted(got) == sorted(ref), "%s: ids %s vs oracle %s" % (what, sorted(got), sorted(ref))
    for t in ref:
        assert bits_equal(got[t], ref[t]), "%s: corners of tag %d" % (what, t)


def check_saddles_against_ground_truth(saddles_xy, gt, width, height, what="", tol=0.3, margin=6.0):
    """Implementation-independent check of the hot path: `gt` = {tag id: 4x2 projected corners} of a frame rendered from a
    known homography (synth.py).  Every drawn tag corner inside the frame (a saddle of the printed pattern: the tag's black
    border meets the small black square) must have a refined saddle within `tol` px.  Returns the distances."""
    s = np.asarray(saddles_xy, np.float64)
    g = np.unique(np.concatenate([gt[t] for t in gt]).round(6), axis=0)
    g = g[(g[:, 0] > margin) & (g[:, 0] < width - margin) & (g[:, 1] > margin) & (g[:, 1] < height - margin)]
    assert len(g) >= 100 and len(s) >= len(g), (what, len(g), len(s))
    d = np.hypot(g[:, None, 0] - s[None, :, 0], g[:, None, 1] - s[None, :, 1]).min(axis=1)
    assert d.max() < tol, "%s: the corner at %s has no saddle within %.2f px (nearest %.3f)" % (what, g[d.argmax()], tol, d.max())
    return d
