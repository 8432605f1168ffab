"""Coefficients of the odd polynomial z * P(z^2) ~ atan(z) on [0, 1] used by the board search's
bounded angle approximation (aprilgrid-rs_amd/csrc/host_tail.cpp, LazyAngle): iteratively re-weighted
least squares on Chebyshev nodes (a poor man's Remez), then the maximum error on a dense grid."""
import numpy as np


def fit(terms):
    t = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000)
    z = (t + 1) / 2
    A = np.stack([z ** (2 * k + 1) for k in range(terms)], 1)
    w = np.ones_like(z)
    for _ in range(60):
        c = np.linalg.lstsq(A * w[:, None], np.arctan(z) * w, rcond=None)[0]
        e = np.abs(A @ c - np.arctan(z))
        w = w * (1 + 4 * e / e.max())
        w /= w.mean()
    zz = np.linspace(0, 1, 2000001)
    err = np.abs(sum(c[k] * zz ** (2 * k + 1) for k in range(terms)) - np.arctan(zz)).max()
    return c, err


if __name__ == "__main__":
    for terms in (3, 5, 6, 7, 8):  # 3: the coarse first level (float), 7: the fine one (binary64)
        c, err = fit(terms)
        print("%d terms: max error %.3e rad = %.3e degrees" % (terms, err, np.degrees(err)))
        if terms in (3, 7):
            print("  coefficients:", ", ".join(repr(float(v)) for v in c))
