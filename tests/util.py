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
