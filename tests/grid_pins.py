"""Implementation-independent pins on the reference's own images.

The reference's tests assert only how many tags each image yields (tests/test_detector.rs:21-32).  What the images
themselves say, whatever code reads them:

  ids       every image shows whole AprilGrid boards printed from id 0: the id set is exactly range(N)
            (36: one 6 x 6 board; 66: the 11 x 6 board of iphone.png; 72: two 6 x 6 boards, the second from id 36);
  quads     a tag's four corners form a convex quadrilateral, and every tag of every image winds the same way
            (the corner order is fixed by the decode: rotate_left(rotation), reverse -- src/detector.rs:467-470);
  grid      the corners are the image of a REGULAR PLANAR GRID -- tag (row r, column c) = id // columns, id % columns, tag
            pitch 1 + 0.3 tag edges (the chart script's layout, scripts/generate_aprilgrid.py:1114-1167: ids row by row,
            `tag_spacing` between tags; the detector's default tag_spacing_ratio 0.3, src/detector.rs:34) -- seen through ONE
            smooth camera: a homography followed by a radially symmetric lens with 6 parameters (centre, focal length,
            three odd-polynomial terms of the equidistant model; the EuRoC / TUM-VI images are wide-angle: a homography
            alone misses their corners by 6 .. 48 px).  All 144 .. 264 corners of a board lie within 1 px of the fitted
            model (measured: max 0.12 .. 0.52 px, rms 0.05 .. 0.18 px).  14 parameters against 288 .. 528 coordinates: a
            detector that misplaced a corner by a pixel, swapped two ids or mis-ordered a tag's corners could not pass.

Used by tests/test_oracle_pins.py (the oracle, CPU) and tests/test_gpu_parity.py (agx_detect / agx_detect_batch with the
device tail, GPU)."""
import numpy as np

# image -> list of boards: (first id, columns, rows)
BOARDS = {
    "iphone.png": [(0, 11, 6)],
    "two_boards.png": [(0, 6, 6), (36, 6, 6)],
}
DEFAULT_BOARDS = [(0, 6, 6)]
SPACING = 0.3
UNIT = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]])  # a tag's corners in the detector's order, in tag edges
MAX_RESIDUAL_PX = 1.0


def boards_of(name):
    return BOARDS.get(name, DEFAULT_BOARDS)


def _project(H, p):
    q = (H @ np.c_[p, np.ones(len(p))].T).T
    return q[:, :2] / q[:, 2:]


def fit_homography(src, dst):
    """Least-squares homography src -> dst (normalised DLT, binary64)."""
    def normalise(p):
        m = p.mean(0)
        s = np.sqrt(2.0) / np.mean(np.hypot(*(p - m).T))
        return (p - m) * s, np.array([[s, 0, -s * m[0]], [0, s, -s * m[1]], [0, 0, 1.0]])
    a, ta = normalise(src)
    b, tb = normalise(dst)
    rows = []
    for (x, y), (u, v) in zip(a, b):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    h = np.linalg.svd(np.array(rows))[2][-1].reshape(3, 3)
    return np.linalg.inv(tb) @ h @ ta


def _camera(p, src):
    """Homography (8) -> radially symmetric lens about (cx, cy): r_d = f (t + k1 t^3 + k2 t^5 + k3 t^7), t = atan(r_u / f)."""
    q = _project(np.append(p[:8], 1.0).reshape(3, 3), src)
    c, f = p[8:10], p[10]
    d = q - c
    ru = np.hypot(*d.T) + 1e-12
    t = np.arctan(ru / f)
    rd = f * (t + p[11] * t ** 3 + p[12] * t ** 5 + p[13] * t ** 7)
    return c + d * (rd / ru)[:, None]


def board_model_points(first_id, cols, rows):
    """Corner coordinates of every tag of a board in tag edges: [rows * cols * 4, 2], tag order = id order."""
    pts = []
    for t in range(rows * cols):
        r, c = divmod(t, cols)
        pts.append(np.array([c * (1.0 + SPACING), r * (1.0 + SPACING)]) + UNIT)
    return np.concatenate(pts)


def grid_residuals(tags, first_id, cols, rows, width, height):
    """-> (residuals of the homography alone, residuals of homography + lens), pixels, one per corner."""
    from scipy.optimize import least_squares
    src = board_model_points(first_id, cols, rows)
    dst = np.concatenate([np.asarray(tags[first_id + t], np.float64) for t in range(rows * cols)])
    H = fit_homography(src, dst)
    res_h = np.hypot(*(_project(H, src) - dst).T)
    best = None
    for f0 in (150.0, 250.0, 400.0, 700.0, 1500.0, 5000.0):  # (the focal length is what a local fit cannot find from afar)
        p0 = np.concatenate([(H / H[2, 2]).ravel()[:8], [width / 2.0, height / 2.0, f0], [0.0, 0.0, 0.0]])
        try:
            r = least_squares(lambda p: (_camera(p, src) - dst).ravel(), p0, x_scale="jac", max_nfev=3000)
        except (ValueError, np.linalg.LinAlgError):
            continue
        res = np.hypot(*(_camera(r.x, src) - dst).T)
        if best is None or res.max() < best.max():
            best = res
    return res_h, best


def check_image(name, tags, shape, expected_count=None):
    """All three pins on one image's {id: 4x2 corners}; returns a summary dict (for the tests' messages / DESIGN)."""
    boards = boards_of(name)
    n = sum(c * r for _, c, r in boards)
    if expected_count is not None:
        assert n == expected_count, (name, n, expected_count)
    assert sorted(tags) == list(range(n)), "%s: ids %s" % (name, sorted(tags))
    signs = set()
    for t, c in tags.items():
        c = np.asarray(c, np.float64)
        e = np.roll(c, -1, 0) - c
        z = e[:, 0] * np.roll(e, -1, 0)[:, 1] - e[:, 1] * np.roll(e, -1, 0)[:, 0]
        assert (z < 0).all() or (z > 0).all(), "%s: tag %d is not a convex quad: %s" % (name, t, c.tolist())
        signs.add(bool(z[0] < 0))
        edges = np.hypot(*e.T)
        assert edges.min() > 4.0 and edges.max() < 3.0 * edges.min(), "%s: tag %d degenerate (edges %s)" % (name, t, edges)
    assert signs == {True}, "%s: winding differs between tags (or from the other images)" % name
    out = {"tags": n, "homography_max_px": 0.0, "camera_max_px": 0.0, "camera_rms_px": 0.0}
    h, w = shape[:2]
    for first, cols, rows in boards:
        res_h, res = grid_residuals(tags, first, cols, rows, w, h)
        assert res is not None, name
        assert res.max() < MAX_RESIDUAL_PX, "%s board from id %d: a corner %.2f px off the planar-grid model (corner %d)" % (
            name, first, res.max(), int(res.argmax()))
        out["homography_max_px"] = max(out["homography_max_px"], float(res_h.max()))
        out["camera_max_px"] = max(out["camera_max_px"], float(res.max()))
        out["camera_rms_px"] = max(out["camera_rms_px"], float(np.sqrt((res ** 2).mean())))
    return out
