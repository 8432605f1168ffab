"""A second, independent restatement of the hot path -- numpy, written from SURVEY.md Appendix A (the
numeric spec), array-at-a-time instead of the oracle's pixel loops, scipy's labelling instead of the
stack flood fill -- must agree with oracle/agx_oracle.c BIT FOR BIT on blur, response, minimum,
cluster table, centroids and saddle x / y / k (angles: numpy's arctan2 / arccos vs glibc, 1e-4 deg).
Guards the oracle against transcription errors; it cannot pin it to the Rust crate (no toolchain)."""
import math

import numpy as np
import pytest
from scipy import ndimage

from tests.util import bits_equal, load_image, synth_module

F = np.float32


def np_blur_weights():
    x = np.arange(-3, 4).astype(F)
    arg = (-(x * x) / F(2.0 * 1.5 * 1.5)).astype(F)
    # A.1: expf per tap -- the correctly rounded one (glibc's expf, which Rust's f32::exp calls on Linux,
    # is; numpy's own float32 exp is 1 ulp off at x = +-2: 0.4111123 instead of 0.41111228) ...
    w = np.array([F(math.exp(float(v))) for v in arg], F)
    s = F(0.0)
    for v in w:
        s = F(s + v)  # ... summed in index order
    return (w / s).astype(F)


def np_blur(luma, w):
    h, wd = luma.shape
    xs = np.arange(wd)
    tmp = np.zeros((h, wd), F)
    for i in range(7):  # A.2: taps in index order, mul then add, clamp-to-edge
        tmp = (tmp + luma[:, np.clip(xs + i - 3, 0, wd - 1)] * w[i]).astype(F)
    ys = np.arange(h)
    out = np.zeros((h, wd), F)
    for i in range(7):  # A.3
        out = (out + tmp[np.clip(ys + i - 3, 0, h - 1), :] * w[i]).astype(F)
    return out


def np_hessian(b):
    out = np.zeros_like(b)
    v11, v12, v13 = b[:-2, :-2], b[:-2, 1:-1], b[:-2, 2:]
    v21, v22, v23 = b[1:-1, :-2], b[1:-1, 1:-1], b[1:-1, 2:]
    v31, v32, v33 = b[2:, :-2], b[2:, 1:-1], b[2:, 2:]
    lxx = ((v21 - (v22 * F(2.0))) + v23).astype(F)  # A.4
    lyy = ((v12 - (v22 * F(2.0))) + v32).astype(F)
    lxy = ((((v13 - v11) + v31) - v33) * F(0.25)).astype(F)
    out[1:-1, 1:-1] = (lxx * lyy).astype(F) - (lxy * lxy).astype(F)
    return out


def np_chain(luma, pmat, cone, min_angle=30.0, max_angle=60.0):
    h, wd = luma.shape
    blur = np_blur(luma, np_blur_weights())
    resp = np_hessian(blur)
    mn = resp.min()
    thr = F(mn * F(0.05))
    lab, n = ndimage.label(resp < thr, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])  # 4-connected
    idx = np.arange(1, n + 1)
    flat = np.arange(h * wd).reshape(h, wd)
    first = ndimage.minimum(flat, lab, idx).astype(np.int64) if n else np.zeros(0, np.int64)
    order = np.argsort(first)  # A.5: emission order = ascending first (smallest) pixel index
    yy, xx = np.mgrid[0:h, 0:wd]
    cnt = ndimage.sum(np.ones_like(lab), lab, idx)[order] if n else np.zeros(0)
    cx = (ndimage.sum(xx, lab, idx)[order].astype(F) / cnt.astype(F)).astype(F) if n else np.zeros(0, F)
    cy = (ndimage.sum(yy, lab, idx)[order].astype(F) / cnt.astype(F)).astype(F) if n else np.zeros(0, F)
    first = first[order]
    # A.6 rochade_refine, all candidates at once
    rx = np.where(cx >= 0, np.floor(cx + F(0.5)), np.ceil(cx - F(0.5))).astype(np.int64)  # round half away from zero
    ry = np.where(cy >= 0, np.floor(cy + F(0.5)), np.ceil(cy - F(0.5))).astype(np.int64)
    ok = (ry - 4 >= 0) & (ry + 4 < h) & (rx - 4 >= 0) & (rx + 4 < wd)
    rx, ry = rx[ok], ry[ok]
    m = len(rx)
    win = np.stack([np.stack([blur[ry - 4 + a, rx - 4 + b] for b in range(9)], 1) for a in range(9)], 1) if m else np.zeros((0, 9, 9), F)
    patch = np.zeros((m, 25), F)
    for r in range(5):
        for c in range(5):
            acc = np.zeros(m, F)
            for pr in range(5):
                for pc in range(5):
                    acc = (acc + win[:, r + pr, c + pc] * cone[pr * 5 + pc]).astype(F)
            patch[:, r * 5 + c] = acc
    prm = np.zeros((m, 6), F)
    for j in range(6):
        acc = np.zeros(m, F)
        for i in range(25):
            acc = (acc + pmat[i, j] * patch[:, i]).astype(F)
        prm[:, j] = acc
    a1, a2, a3, a4, a5 = (prm[:, j] for j in range(5))
    fxx, fyy, fxy = (F(2.0) * a1).astype(F), (F(2.0) * a3).astype(F), a2
    d = ((fxx * fyy).astype(F) - (fxy * fxy).astype(F)).astype(F)
    keep = d < 0
    # 2x2 partial-pivot LU of [[2a1, a2], [a2, 2a3]] x = [-a4, -a5]
    A0, B0, R0, A1, B1, R1 = fxx, a2, -a4, a2, fyy, -a5
    sw = np.abs(A1) > np.abs(A0)
    pa, pb, pr_ = np.where(sw, A1, A0), np.where(sw, B1, B0), np.where(sw, R1, R0)
    qa, qb, qr = np.where(sw, A0, A1), np.where(sw, B0, B1), np.where(sw, R0, R1)
    with np.errstate(all="ignore"):
        l = (qa / pa).astype(F)
        u22 = (qb - (l * pb).astype(F)).astype(F)
        y0 = ((qr - (l * pr_).astype(F)).astype(F) / u22).astype(F)
        x0 = ((pr_ - (pb * y0).astype(F)).astype(F) / pa).astype(F)
        keep &= (np.abs(x0) <= 1) & (np.abs(y0) <= 1)
        c5 = ((a1 + a3).astype(F) / F(2.0)).astype(F)
        c4 = ((a1 - a3).astype(F) / F(2.0)).astype(F)
        c3 = (a2 / F(2.0)).astype(F)
        k = np.sqrt(((c4 * c4).astype(F) + (c3 * c3).astype(F)).astype(F)).astype(F)
        keep &= np.abs(c5) < k
        phi = (np.arccos((-c5 / k).astype(F)).astype(F) / F(2.0) / F(np.pi) * F(180.0)).astype(F)
        theta = (np.arctan2(c3, c4).astype(F) / F(2.0) / F(np.pi) * F(180.0)).astype(F)
    x = (rx.astype(F) + x0).astype(F)[keep]
    y = (ry.astype(F) + y0).astype(F)[keep]
    k, phi, theta = k[keep], phi[keep], theta[keep]
    refined = dict(x=x, y=y, k=k, theta=theta, phi=phi)
    if len(k):
        f = (k >= F(k.max() / F(10.0))) & (phi >= F(min_angle)) & (phi <= F(max_angle))  # A.7
    else:
        f = np.zeros(0, bool)
    return dict(blur=blur, resp=resp, min=mn, first=first, sizes=cnt.astype(np.int64), cx=cx, cy=cy, refined=refined,
                saddles={q: v[f] for q, v in refined.items()})


def _frames():
    synth = synth_module()
    board = np.asarray(synth.render_frame(17, 320, 240)[0])
    noise = np.asarray(synth.render_frame(5, 128, 96, pure_noise=True)[0])
    real = load_image("EuRoC.png")[100:340, 200:560]
    return [("synthetic board", board), ("pure noise", noise), ("EuRoC crop", np.ascontiguousarray(real))]


@pytest.mark.parametrize("name,img", _frames(), ids=lambda v: v if isinstance(v, str) else "")
def test_numpy_restatement_equals_the_oracle(name, img):
    from oracle import oracle as O
    pmat, cone = O.refine_constants(2)
    assert bits_equal(np_blur_weights(), O.blur_weights(1.5))
    luma = (img.astype(F) / F(255.0)).astype(F)
    got = np_chain(luma, pmat, cone)
    ref, d = O.refined_saddle_points(img, debug=True)
    assert bits_equal(got["blur"], d["blur"]), "blur"
    assert bits_equal(got["resp"], d["resp"]), "response"
    assert bits_equal(F(got["min"]), F(d["min_resp"]))
    assert np.array_equal(got["first"], d["first_index"]) and np.array_equal(got["sizes"], d["sizes"]), "cluster table"
    assert bits_equal(got["cx"], d["centers"][:, 0]) and bits_equal(got["cy"], d["centers"][:, 1]), "centroids"
    for key, want in (("refined", d["refined"]), ("saddles", ref)):
        g = got[key]
        assert len(g["x"]) == len(want) and len(want) > 10, (key, len(g["x"]), len(want))
        for f in ("x", "y", "k"):
            assert bits_equal(g[f], want[f]), (key, f)
        for f in ("theta", "phi"):
            assert np.max(np.abs(g[f] - want[f])) <= 1e-4, (key, f)


def test_pseudo_inverse_equals_numpy_pinv():
    """The 25x6 pseudo-inverse of the quadratic design matrix (detector.rs:208-237): numpy's SVD-based
    pinv in binary64, rounded to f32, gives the oracle's (and the library's) table to the last bit or
    the one next to it."""
    from oracle import oracle as O
    pmat, _ = O.refine_constants(2)
    rows = [[x * x, x * y, y * y, x, y, 1.0] for y in range(-2, 3) for x in range(-2, 3)]
    pinv = np.linalg.pinv(np.asarray(rows, np.float64)).T.astype(F)  # [25][6]
    ulp = np.abs(pinv.view(np.int32).astype(np.int64) - pmat.view(np.int32).astype(np.int64))
    nz = (np.abs(pmat) > 1e-12)
    assert ulp[nz].max() <= 1, ulp.max()
    assert np.max(np.abs(pinv[~nz])) < 1e-15
