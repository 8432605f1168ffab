#!/usr/bin/env python3
"""How far can the real crate's saddles lie from the restatement's?  (missing #2.)

north_star asks for tag ids bit-identical and corners "within a stated pixel tolerance" against the reference CPU path.
The Rust crate cannot be built here, so the tolerance is derived: the oracle (oracle/agx_oracle.c, the C restatement
every parity test compares with) differs KNOWINGLY from the crate in exactly two places, both inside rochade_refine --

  p_mat     the crate: faer's f32 Householder QR + 25 least-squares solves (/root/reference/src/detector.rs:224-236);
            the oracle: the exact pseudo-inverse (binary64 normal equations), rounded once to f32
  find_xy   the crate: faer's 2x2 partial-pivot LU solve (/root/reference/src/math_util.rs:5-12), operation order unknown;
            the oracle: textbook elimination with true divisions

-- plus one place where Rust's `iter().sum::<f32>()` is restated as a left fold (the cone kernel's normaliser,
detector.rs:253).  This script re-evaluates rochade_refine (numpy, float32, the oracle's operation order: the baseline
variant must reproduce the oracle's list BIT FOR BIT, and is asserted to) with plausible alternatives for each:

  pmat:householder_f32      25x6 design matrix factored by a textbook Householder QR in float32, 25 unit right-hand
                            sides through Q^T and a float32 back substitution -- the operation faer performs
  pmat:householder_f32_b    the same with the reflector norms accumulated pairwise (numpy's sum) instead of left to right
  pmat:normal_eq_f32        float32 normal equations + Cholesky: cruder than any QR (condition number squared)
  pmat:ulp1 .. pmat:ulp4    the exact table with every entry moved by a random -n..n ulp (3 seeds each)
  find_xy:recip             the two divisions of the elimination as multiplications by a float32 reciprocal
  find_xy:nopivot           no row exchange
  find_xy:cramer            determinant form
  cone:pairwise / cone:f64  the normaliser summed pairwise / in binary64
  all:worst                 householder_f32 + recip + pairwise together

and reports, per variant, over the reference's 9 images and the bench's 256 synthetic frames: the largest |dx|, |dy| (px),
|dk| (relative), |dtheta|, |dphi| (degrees) over the saddles both lists keep; how many saddles appear / disappear and at
which decision (d < 0, |x0|,|y0| <= 1, |c5| < k, k >= max_k / 10, the phi window); and whether tag ids or corners change
when the variant's list goes through the host tail (oracle's detect_tail).

    python tests/sensitivity_study.py [--frames 256] [--jobs 8] > profiles/r5_sensitivity.json
(imports the oracle: test infrastructure, lives under tests/; the short form runs in tests/test_sensitivity.py)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F = np.float32


# ---------------------------------------------------------------------------------------------- p_mat variants
def design_matrix():
    rows = [[x * x, x * y, y * y, x, y, 1.0] for y in range(-2, 3) for x in range(-2, 3)]  # detector.rs:209-222
    return np.asarray(rows, F)


def pmat_householder_f32(pairwise_norm=False):
    """min ||A x - e_i|| for the 25 unit vectors by Householder QR, every operation in float32 (textbook: Golub & Van
    Loan alg. 5.2.1; reflector v with v[0] = x0 + sign(x0) * ||x||, H = I - 2 v v^T / (v^T v))."""
    A = design_matrix().copy()
    m, n = A.shape
    B = np.eye(m, dtype=F)  # the 25 right-hand sides, transformed along

    def dot(a, b):
        if pairwise_norm:
            return F(np.sum((a * b).astype(F), dtype=F))
        s = F(0.0)
        for u, v in zip(a, b):
            s = F(s + F(u * v))
        return s

    for k in range(n):
        x = A[k:, k].copy()
        nrm = F(np.sqrt(dot(x, x)))
        if nrm == 0:
            continue
        v = x.copy()
        v[0] = F(x[0] + (nrm if x[0] >= 0 else -nrm))
        vtv = dot(v, v)
        for M in (A, B):
            for j in range(M.shape[1]):
                if M is A and j < k:
                    continue
                col = M[k:, j]
                tau = F(F(2.0) * dot(v, col) / vtv)
                M[k:, j] = (col - (tau * v).astype(F)).astype(F)
    R = A[:n, :n]
    P = np.zeros((m, n), F)
    for i in range(m):
        y = B[:n, i]
        x = np.zeros(n, F)
        for r in range(n - 1, -1, -1):
            s = y[r]
            for c in range(r + 1, n):
                s = F(s - F(R[r, c] * x[c]))
            x[r] = F(s / R[r, r])
        P[i] = x
    return P


def pmat_normal_eq_f32():
    A = design_matrix()
    G = (A.T @ A).astype(F)
    n = 6
    L = np.zeros((n, n), F)
    for i in range(n):
        for j in range(i + 1):
            s = G[i, j]
            for k in range(j):
                s = F(s - F(L[i, k] * L[j, k]))
            L[i, j] = F(np.sqrt(s)) if i == j else F(s / L[j, j])
    P = np.zeros((25, 6), F)
    for i in range(25):
        b = A[i]  # A^T e_i
        y = np.zeros(n, F)
        for r in range(n):
            s = b[r]
            for c in range(r):
                s = F(s - F(L[r, c] * y[c]))
            y[r] = F(s / L[r, r])
        x = np.zeros(n, F)
        for r in range(n - 1, -1, -1):
            s = y[r]
            for c in range(r + 1, n):
                s = F(s - F(L[c, r] * x[c]))
            x[r] = F(s / L[r, r])
        P[i] = x
    return P


def pmat_perturbed(exact, ulps, seed):
    rng = np.random.default_rng(seed)
    d = rng.integers(-ulps, ulps + 1, exact.shape)
    out = exact.copy()
    nz = exact != 0
    bits = exact.view(np.int32).astype(np.int64)
    out[nz] = (bits[nz] + np.where(exact[nz] < 0, -d[nz], d[nz]) * 1).astype(np.int32).view(F)  # +-n ulp in magnitude order
    # entries that are exactly 0 (e.g. the xy column at x = 0): absolute noise of n ulp of the column's largest entry
    colmax = np.abs(exact).max(axis=0)
    noise = (d * (colmax * F(2.0 ** -24))[None, :]).astype(F)
    out[~nz] = noise[~nz]
    return out


def ulp_distance(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


# ---------------------------------------------------------------------------------------------- cone variants
def cone_kernel(how="fold"):
    g = F(2.0)
    v = np.zeros(25, F)
    for i in range(5):
        for j in range(5):
            a, b = F(g - F(i)), F(g - F(j))
            v[i * 5 + j] = max(F(0.0), F(F(g + F(1.0)) - F(np.sqrt(F(F(a * a) + F(b * b))))))
    if how == "fold":
        s = F(0.0)
        for x in v:
            s = F(s + x)
    elif how == "pairwise":
        s = F(0.0)
        parts = list(v)
        while len(parts) > 1:  # a balanced tree
            parts = [F(parts[i] + parts[i + 1]) if i + 1 < len(parts) else parts[i] for i in range(0, len(parts), 2)]
        s = parts[0]
    else:
        s = F(np.sum(v.astype(np.float64)))
    return (v / s).astype(F)


# ---------------------------------------------------------------------------------------------- rochade_refine in numpy
def patches(blur, cx, cy, cone):
    """detector.rs:266-298: the cone-filtered 5x5 patch of every centre whose 9x9 window lies inside the image."""
    h, wd = blur.shape
    rx = np.where(cx >= 0, np.floor(cx + F(0.5)), np.ceil(cx - F(0.5))).astype(np.int64)  # f32::round
    ry = np.where(cy >= 0, np.floor(cy + F(0.5)), np.ceil(cy - F(0.5))).astype(np.int64)
    ok = (ry - 4 >= 0) & (ry + 4 < h) & (rx - 4 >= 0) & (rx + 4 < wd)
    rx, ry = rx[ok], ry[ok]
    m = len(rx)
    win = np.zeros((m, 9, 9), F)
    for a in range(9):
        for b in range(9):
            win[:, a, b] = blur[ry - 4 + a, rx - 4 + b]
    patch = np.zeros((m, 25), F)
    for r in range(5):
        for c in range(5):
            acc = np.zeros(m, F)
            for pr in range(5):
                for pc in range(5):
                    acc = (acc + win[:, r + pr, c + pc] * cone[pr * 5 + pc]).astype(F)
            patch[:, r * 5 + c] = acc
    return rx, ry, patch


def solve(pmat, patch, rx, ry, find_xy="div", min_angle=30.0, max_angle=60.0):
    """detector.rs:320-355 + the k / phi filter :432-445 for every patch at once.  Returns per-candidate arrays and the
    decision masks (each evaluated for every candidate, whatever the earlier ones said)."""
    m = len(rx)
    prm = np.zeros((m, 6), F)
    for j in range(6):
        acc = np.zeros(m, F)
        for i in range(25):
            acc = (acc + pmat[i, j] * patch[:, i]).astype(F)
        prm[:, j] = acc
    a1, a2, a3, a4, a5 = (prm[:, j] for j in range(5))
    fxx, fyy, fxy = (F(2.0) * a1).astype(F), (F(2.0) * a3).astype(F), a2
    d = ((fxx * fyy).astype(F) - (fxy * fxy).astype(F)).astype(F)
    A0, B0, R0, A1, B1, R1 = fxx, a2, -a4, a2, fyy, -a5
    with np.errstate(all="ignore"):
        if find_xy == "cramer":
            det = ((A0 * B1).astype(F) - (B0 * A1).astype(F)).astype(F)
            x0 = (((R0 * B1).astype(F) - (B0 * R1).astype(F)).astype(F) / det).astype(F)
            y0 = (((A0 * R1).astype(F) - (R0 * A1).astype(F)).astype(F) / det).astype(F)
        else:
            sw = (np.abs(A1) > np.abs(A0)) if find_xy != "nopivot" else np.zeros(m, bool)
            pa, pb, pr_ = np.where(sw, A1, A0), np.where(sw, B1, B0), np.where(sw, R1, R0)
            qa, qb, qr = np.where(sw, A0, A1), np.where(sw, B0, B1), np.where(sw, R0, R1)
            if find_xy == "recip":
                ipa = (F(1.0) / pa).astype(F)
                l = (qa * ipa).astype(F)
                u22 = (qb - (l * pb).astype(F)).astype(F)
                y0 = ((qr - (l * pr_).astype(F)).astype(F) * (F(1.0) / u22).astype(F)).astype(F)
                x0 = ((pr_ - (pb * y0).astype(F)).astype(F) * ipa).astype(F)
            else:
                l = (qa / pa).astype(F)
                u22 = (qb - (l * pb).astype(F)).astype(F)
                y0 = ((qr - (l * pr_).astype(F)).astype(F) / u22).astype(F)
                x0 = ((pr_ - (pb * y0).astype(F)).astype(F) / pa).astype(F)
        c5 = ((a1 + a3).astype(F) / F(2.0)).astype(F)
        c4 = ((a1 - a3).astype(F) / F(2.0)).astype(F)
        c3 = (a2 / F(2.0)).astype(F)
        k = np.sqrt(((c4 * c4).astype(F) + (c3 * c3).astype(F)).astype(F)).astype(F)
        phi = (np.arccos((-c5 / k).astype(F)).astype(F) / F(2.0) / F(np.pi) * F(180.0)).astype(F)
        theta = (np.arctan2(c3, c4).astype(F) / F(2.0) / F(np.pi) * F(180.0)).astype(F)
        m_d = d < 0
        m_move = (np.abs(x0) <= 1) & (np.abs(y0) <= 1)
        m_c5 = np.abs(c5) < k
        refined = m_d & m_move & m_c5
        max_k = k[refined].max() if refined.any() else F(0.0)
        m_k = k >= F(max_k / F(10.0))
        m_phi = (phi >= F(min_angle)) & (phi <= F(max_angle))
    x = (rx.astype(F) + x0).astype(F)
    y = (ry.astype(F) + y0).astype(F)
    return dict(x=x, y=y, k=k, theta=theta, phi=phi, d=d, x0=x0, y0=y0, c5=c5, max_k=max_k,
                masks=dict(d=m_d, move=m_move, c5=m_c5, k=m_k, phi=m_phi), refined=refined, kept=refined & m_k & m_phi)


def saddle_array(res, O):
    kept = res["kept"]
    out = np.zeros(int(kept.sum()), O.SADDLE_DTYPE)
    for f in ("x", "y", "k", "theta", "phi"):
        out[f] = res[f][kept]
    return out


# ---------------------------------------------------------------------------------------------- the study
def variants(exact_pmat):
    cone0 = cone_kernel("fold")
    v = {"baseline": (exact_pmat, cone0, "div"),
         "pmat:householder_f32": (pmat_householder_f32(False), cone0, "div"),
         "pmat:householder_f32_b": (pmat_householder_f32(True), cone0, "div"),
         "pmat:normal_eq_f32": (pmat_normal_eq_f32(), cone0, "div")}
    for n in (1, 2, 4):
        for seed in range(3):
            v["pmat:ulp%d_seed%d" % (n, seed)] = (pmat_perturbed(exact_pmat, n, 1000 * n + seed), cone0, "div")
    v["find_xy:recip"] = (exact_pmat, cone0, "recip")
    v["find_xy:nopivot"] = (exact_pmat, cone0, "nopivot")
    v["find_xy:cramer"] = (exact_pmat, cone0, "cramer")
    v["cone:pairwise"] = (exact_pmat, cone_kernel("pairwise"), "div")
    v["cone:f64"] = (exact_pmat, cone_kernel("f64"), "div")
    v["all:worst"] = (v["pmat:householder_f32"][0], cone_kernel("pairwise"), "recip")
    return v


def study_frame(args):
    """One image through the oracle (blur plane, centroids, its own lists) and every variant.  -> per-variant record."""
    name, img = args
    from oracle import oracle as O
    ref, dbg = O.refined_saddle_points(img, debug=True)
    blur = dbg["blur"]
    cx, cy = dbg["centers"][:, 0].copy(), dbg["centers"][:, 1].copy()
    exact_pmat, exact_cone = O.refine_constants(2)
    vs = variants(exact_pmat)
    assert np.array_equal(vs["baseline"][1].view(np.uint32), exact_cone.view(np.uint32)), "cone kernel restated wrongly"
    grey = O.luma_u8(img)
    cache = {}
    out = {}
    base = None
    for vname, (pmat, cone, fxy) in vs.items():
        ck = cone.tobytes()
        if ck not in cache:
            cache[ck] = patches(blur, cx, cy, cone)
        rx, ry, patch = cache[ck]
        res = solve(pmat, patch, rx, ry, fxy)
        sad = saddle_array(res, O)
        tags = O.detect_tail(grey, sad)
        if vname == "baseline":
            # the numpy re-evaluation IS the oracle's: same lists, bit for bit (angles: numpy's arccos / arctan2 vs glibc)
            assert len(sad) == len(ref), (name, len(sad), len(ref))
            for f in ("x", "y", "k"):
                assert np.array_equal(sad[f].view(np.uint32), ref[f].view(np.uint32)), (name, f)
            assert int(res["refined"].sum()) == len(dbg["refined"])
            base = (res, sad, tags)
            # how close the baseline's own candidates come to each threshold
            r = res
            with np.errstate(all="ignore"):
                margins = {
                    "move_px": float(np.min(np.abs(np.maximum(np.abs(r["x0"]), np.abs(r["y0"]))[r["masks"]["d"]] - 1.0))) if r["masks"]["d"].any() else None,
                    "phi_deg": float(np.min(np.minimum(np.abs(r["phi"][r["refined"]] - 30.0), np.abs(r["phi"][r["refined"]] - 60.0)))) if r["refined"].any() else None,
                    "k_rel": float(np.min(np.abs(r["k"][r["refined"]] / (r["max_k"] / 10.0) - 1.0))) if r["refined"].any() else None,
                }
            out[vname] = {"candidates": int(len(rx)), "refined": int(res["refined"].sum()), "saddles": int(len(sad)), "tags": len(tags),
                          "closest_to_threshold": margins}
            continue
        b, bsad, btags = base
        both = b["kept"] & res["kept"]
        rec = {"saddles": int(len(sad)), "gained": int((res["kept"] & ~b["kept"]).sum()), "lost": int((b["kept"] & ~res["kept"]).sum()),
               "flips": {s: int((res["masks"][s] != b["masks"][s])[b["masks"]["d"] | res["masks"]["d"]].sum()) if s != "d"
                         else int((res["masks"][s] != b["masks"][s]).sum()) for s in ("d", "move", "c5", "k", "phi")}}
        if both.any():
            rec["max_dx_px"] = float(np.max(np.abs(res["x"][both].astype(np.float64) - b["x"][both])))
            rec["max_dy_px"] = float(np.max(np.abs(res["y"][both].astype(np.float64) - b["y"][both])))
            rec["max_dx0_ulp"] = int(np.max(ulp_distance(res["x0"][both], b["x0"][both])))
            rec["max_dy0_ulp"] = int(np.max(ulp_distance(res["y0"][both], b["y0"][both])))
            rec["max_dk_rel"] = float(np.max(np.abs(res["k"][both].astype(np.float64) - b["k"][both]) / b["k"][both]))
            rec["max_dk_ulp"] = int(np.max(ulp_distance(res["k"][both], b["k"][both])))
            rec["max_dtheta_deg"] = float(np.max(np.abs(res["theta"][both].astype(np.float64) - b["theta"][both])))
            rec["max_dphi_deg"] = float(np.max(np.abs(res["phi"][both].astype(np.float64) - b["phi"][both])))
        ids_equal = sorted(tags) == sorted(btags)
        rec["tag_ids_equal"] = ids_equal
        rec["tags"] = len(tags)
        if ids_equal and tags:
            rec["max_corner_px"] = float(max(np.max(np.abs(tags[t].astype(np.float64) - btags[t])) for t in tags))
        out[vname] = rec
    return name, out


def pmat_table(exact):
    t = {}
    for n, (p, _, _) in variants(exact).items():
        if n.startswith("pmat:") or n == "all:worst":
            nz = exact != 0
            t[n] = {"max_ulp_vs_exact": int(ulp_distance(p[nz], exact[nz]).max()), "max_abs_where_exact_is_0": float(np.abs(p[~nz]).max()) if (~nz).any() else 0.0,
                    "max_rel": float(np.max(np.abs(p[nz].astype(np.float64) - exact[nz]) / np.abs(exact[nz])))}
    return t


def aggregate(per_frame):
    agg = {}
    for _, rec in per_frame:
        for v, r in rec.items():
            a = agg.setdefault(v, {})
            if v == "baseline":
                for key in ("candidates", "refined", "saddles", "tags"):
                    a[key] = a.get(key, 0) + r[key]
                for key, val in r["closest_to_threshold"].items():
                    if val is not None:
                        a.setdefault("closest_to_threshold", {})[key] = min(a.get("closest_to_threshold", {}).get(key, 1e30), val)
                continue
            for key in ("gained", "lost"):
                a[key] = a.get(key, 0) + r[key]
            for s, n in r["flips"].items():
                a.setdefault("flips", {})[s] = a.get("flips", {}).get(s, 0) + n
            for key in ("max_dx_px", "max_dy_px", "max_dx0_ulp", "max_dy0_ulp", "max_dk_rel", "max_dk_ulp", "max_dtheta_deg", "max_dphi_deg", "max_corner_px"):
                if key in r:
                    a[key] = max(a.get(key, 0), r[key])
            a["frames_with_different_tag_ids"] = a.get("frames_with_different_tag_ids", 0) + (0 if r["tag_ids_equal"] else 1)
            a["frames_with_different_saddle_count"] = a.get("frames_with_different_saddle_count", 0) + (1 if r["gained"] or r["lost"] else 0)
    return agg


def inputs(n_synth, names=None):
    from tests.util import ALL_IMAGES, load_image, synth_module
    items = [(n, load_image(n)) for n in (names if names is not None else ALL_IMAGES)]
    synth = synth_module()
    for i in range(n_synth):
        items.append(("synthetic_1280x800_frame_%d" % i, np.asarray(synth.render_frame(i, 1280, 800)[0])))
    return items


def run(n_synth=256, jobs=8, names=None):
    from oracle import oracle as O
    O.build()
    items = inputs(n_synth, names)
    if jobs > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(jobs) as pool:
            per_frame = pool.map(study_frame, items, chunksize=1)
    else:
        per_frame = [study_frame(it) for it in items]
    exact, _ = O.refine_constants(2)
    agg = aggregate(per_frame)
    worst_px = max(max(a.get("max_dx_px", 0), a.get("max_dy_px", 0)) for v, a in agg.items() if v != "baseline")
    plausible = [v for v in agg if v.startswith(("pmat:householder", "pmat:ulp", "find_xy:recip", "cone:", "all:worst"))]
    plausible_px = max(max(agg[v].get("max_dx_px", 0), agg[v].get("max_dy_px", 0)) for v in plausible)
    return {"images": len(items), "p_mat_variants_vs_exact": pmat_table(exact), "aggregate": agg,
            "max_px_any_variant": worst_px, "max_px_plausible_variants": plausible_px,
            "plausible_variants": plausible,
            "per_image": {n: r for n, r in per_frame if not n.startswith("synthetic_")}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256, help="synthetic 1280x800 frames of the bench's generator (indices 0..n-1)")
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    print(json.dumps(run(a.frames, a.jobs), indent=1))


if __name__ == "__main__":
    main()
