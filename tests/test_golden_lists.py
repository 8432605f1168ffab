"""The golden lists of tests/golden/saddles_<image>.json (tools/gen_golden_saddles.py): saddle x / y / k as f32 bit
patterns, theta / phi, tag ids and corner bits for the reference's nine fixture images.  They are the hand-off for
pinning the oracle against the real crate ("Pinning the oracle"): the CPU suite holds the oracle to
them, the GPU suite holds the HIP path to them WITHOUT loading the oracle -- so the two stay the same thing a cargo
owner compares tests/parity_dump.rs's output with."""
import json
import os
import sys

import numpy as np
import pytest

from tests.util import ALL_IMAGES, GOLDEN, load_image

ANGLE_TOL_DEG = 1e-3  # theta / phi come out of the platform's acosf / atan2f


def golden(name):
    with open(os.path.join(GOLDEN, "saddles_%s.json" % os.path.splitext(name)[0])) as f:
        return json.load(f)


def bits(hexes):
    return np.array([int(h, 16) for h in hexes], np.uint32)


def check_saddle_fields(x, y, k, theta, phi, g, what):
    s = g["saddles"]
    assert len(x) == len(s["x_bits"]), "%s: %d saddles, golden list has %d" % (what, len(x), len(s["x_bits"]))
    for got, key in ((x, "x_bits"), (y, "y_bits"), (k, "k_bits")):
        assert np.array_equal(np.ascontiguousarray(got, np.float32).view(np.uint32), bits(s[key])), "%s: %s" % (what, key)
    assert np.max(np.abs(np.asarray(theta, np.float64) - np.array(s["theta_deg"])), initial=0.0) <= ANGLE_TOL_DEG, what
    assert np.max(np.abs(np.asarray(phi, np.float64) - np.array(s["phi_deg"])), initial=0.0) <= ANGLE_TOL_DEG, what


def check_tags(tags, g, what):
    want = g["tags"]
    assert sorted(int(i) for i in tags) == sorted(int(i) for i in want), what + ": tag ids"
    for i, corners in tags.items():
        got = np.ascontiguousarray(corners, np.float32).view(np.uint32).reshape(4, 2)
        assert np.array_equal(got, np.array([bits(c) for c in want[str(int(i))]])), "%s: corners of tag %d" % (what, int(i))


@pytest.mark.parametrize("name", ALL_IMAGES)
def test_oracle_reproduces_the_golden_lists(name):
    from oracle import oracle as O
    img = load_image(name)
    g = golden(name)
    assert (g["width"], g["height"]) == (img.shape[1], img.shape[0])
    s = O.refined_saddle_points(img)
    check_saddle_fields(s["x"], s["y"], s["k"], s["theta"], s["phi"], g, name + " (oracle)")
    check_tags(O.detect(img), g, name + " (oracle)")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL_IMAGES)
def test_hip_path_reproduces_the_golden_lists_without_the_oracle(name):
    import aprilgrid_rs_amd as A
    det = A.TagDetector("t36h11")
    try:
        img = load_image(name)
        g = golden(name)
        s = det.refined_saddle_points(img, as_array=True)
        check_saddle_fields(s["x"], s["y"], s["k"], s["theta"], s["phi"], g, name + " (HIP)")
        check_tags(det.detect(img), g, name + " (HIP)")
    finally:
        det.close()
