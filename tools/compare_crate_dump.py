"""Compare the output of bindings/rust/tests/parity_dump.rs (one JSON object per line and
implementation -- "impl": "crate" = the real aprilgrid-rs crate, "amd" = the same calls through bindings/rust; a line without
"impl" is the crate's) with the golden lists of tests/golden/saddles_<image>.json.  No GPU, no oracle.

    python tools/compare_crate_dump.py crate_dump.jsonl [--strict]

Pass criterion = the tolerance DESIGN.md section 3 states against the crate, derived in tests/sensitivity_study.py
(profiles/r5_sensitivity.json: the two places where the restatement knowingly differs from the crate -- faer's f32
Householder QR for p_mat, faer's 2x2 LU for find_xy -- moved every saddle of 265 images by at most 1.2e-4 px = 1 ulp of a
coordinate in [1024, 2048), k by 9e-6 relative, theta / phi by 2e-4 degrees, and flipped no decision):

    tag ids identical, saddle counts identical;
    |dx|, |dy| <= TOL_PX = 1e-3 px for saddles and tag corners;   |dk| <= TOL_K_REL = 1e-4 relative;
    |dtheta|, |dphi| <= 1e-3 degrees.

Differences in the last bits inside the tolerance are reported (how many records, the largest in ulp) but do not fail the
run; --strict fails on any differing bit of x / y / k / corners.  Exit status 0 = pass."""
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ANGLE_TOL_DEG = 1e-3
TOL_PX = 1e-3       # stated corner / saddle tolerance against the crate (8 x the largest deviation any plausible variant produced)
TOL_K_REL = 1e-4    # k: 10 x the largest relative deviation observed


def f32(h):
    return struct.unpack("<f", struct.pack("<I", int(h, 16)))[0]


def ulps(a, b):
    """Distance of two f32 bit patterns (hex strings) in units in the last place."""
    def key(h):
        u = int(h, 16)
        return -(u & 0x7fffffff) if u & 0x80000000 else u
    return abs(key(a) - key(b))


def compare(dump, gold, out, strict):
    bad = 0
    name = gold["image"] + (" [%s]" % dump["impl"] if "impl" in dump else "")
    ds, gs = dump["saddles"], gold["saddles"]
    if len(ds["x_bits"]) != len(gs["x_bits"]):
        out.append("FAIL %s: %d saddles in the crate's list, %d in the golden list" % (name, len(ds["x_bits"]), len(gs["x_bits"])))
        return 1
    for f, tol, rel in (("x_bits", TOL_PX, False), ("y_bits", TOL_PX, False), ("k_bits", TOL_K_REL, True)):
        diff = [(i, a, b) for i, (a, b) in enumerate(zip(ds[f], gs[f])) if a.lower() != b.lower()]
        if not diff:
            continue
        worst_ulp = max(ulps(a, b) for _, a, b in diff)
        worst = max(abs(f32(a) - f32(b)) / (abs(f32(b)) if rel else 1.0) for _, a, b in diff)
        i, a, b = diff[0]
        over = worst > tol
        out.append("%s %s: %s differs at %d of %d saddles (largest %d ulp = %.3g %s, tolerance %g); first at index %d: crate %s golden %s"
                   % ("FAIL" if over or strict else "note", name, f, len(diff), len(gs[f]), worst_ulp, worst, "relative" if rel else "px", tol, i, a, b))
        bad += 1 if over or strict else 0
    for f in ("theta_deg", "phi_deg"):
        worst = max((abs(float(a) - float(b)) for a, b in zip(ds[f], gs[f])), default=0.0)
        if worst > ANGLE_TOL_DEG:
            out.append("FAIL %s: %s differs by up to %.6f degrees (tolerance %g)" % (name, f, worst, ANGLE_TOL_DEG))
            bad += 1
    dt, gt = dump["tags"], gold["tags"]
    if sorted(dt, key=int) != sorted(gt, key=int):
        out.append("FAIL %s: tag ids differ: only in the crate's map %s, only in the golden map %s"
                   % (name, sorted(set(dt) - set(gt), key=int), sorted(set(gt) - set(dt), key=int)))
        bad += 1
    worst_corner, n_corner = 0.0, 0
    for i in sorted(set(dt) & set(gt), key=int):
        for pc, pg in zip(dt[i], gt[i]):
            for a, b in zip(pc, pg):
                if a.lower() != b.lower():
                    n_corner += 1
                    worst_corner = max(worst_corner, abs(f32(a) - f32(b)))
    if n_corner:
        over = worst_corner > TOL_PX
        out.append("%s %s: %d tag corner coordinates differ, by up to %.3g px (tolerance %g)" % ("FAIL" if over or strict else "note", name, n_corner, worst_corner, TOL_PX))
        bad += 1 if over or strict else 0
    return bad


def main(argv):
    strict = "--strict" in argv
    paths = [a for a in argv if not a.startswith("--")]
    out, bad, seen = [], 0, 0
    for line in open(paths[0]):
        line = line.strip()
        if not line.startswith("{"):
            continue
        dump = json.loads(line)
        gpath = os.path.join(GOLDEN, "saddles_%s.json" % os.path.splitext(dump["image"])[0])
        if not os.path.exists(gpath):
            out.append("FAIL %s: no golden list" % dump["image"])
            bad += 1
            continue
        bad += compare(dump, json.load(open(gpath)), out, strict)
        seen += 1
    if out:
        print("\n".join(out))
    if not bad and seen:
        print("%d images: tag ids identical, every field within the stated tolerance (%g px, k %g relative, angles %g degrees)%s"
              % (seen, TOL_PX, TOL_K_REL, ANGLE_TOL_DEG, "" if out else "; in fact bit for bit"))
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
