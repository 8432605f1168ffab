"""Compare the output of tests/parity_dump.rs (one JSON object per line, printed by the real
aprilgrid-rs crate) with the golden lists of tests/golden/saddles_<image>.json.  No GPU, no oracle.

    python tools/compare_crate_dump.py crate_dump.jsonl        exit status 0 = every field of every image agrees"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ANGLE_TOL_DEG = 1e-3


def ulps(a, b):
    """Distance of two f32 bit patterns (hex strings) in units in the last place."""
    def key(h):
        u = int(h, 16)
        return -(u & 0x7fffffff) if u & 0x80000000 else u
    return abs(key(a) - key(b))


def compare(dump, gold, out):
    bad = 0
    name = gold["image"]
    ds, gs = dump["saddles"], gold["saddles"]
    if len(ds["x_bits"]) != len(gs["x_bits"]):
        out.append("%s: %d saddles in the crate's list, %d in the golden list" % (name, len(ds["x_bits"]), len(gs["x_bits"])))
        return 1
    for f in ("x_bits", "y_bits", "k_bits"):
        diff = [(i, a, b) for i, (a, b) in enumerate(zip(ds[f], gs[f])) if a.lower() != b.lower()]
        if diff:
            i, a, b = diff[0]
            out.append("%s: %s differs at %d of %d saddles; first at index %d: crate %s golden %s (%d ulp)"
                       % (name, f, len(diff), len(gs[f]), i, a, b, ulps(a, b)))
            bad += 1
    for f in ("theta_deg", "phi_deg"):
        worst = max((abs(float(a) - float(b)) for a, b in zip(ds[f], gs[f])), default=0.0)
        if worst > ANGLE_TOL_DEG:
            out.append("%s: %s differs by up to %.6f degrees (tolerance %g)" % (name, f, worst, ANGLE_TOL_DEG))
            bad += 1
    dt, gt = dump["tags"], gold["tags"]
    if sorted(dt, key=int) != sorted(gt, key=int):
        out.append("%s: tag ids differ: only in the crate's map %s, only in the golden map %s"
                   % (name, sorted(set(dt) - set(gt), key=int), sorted(set(gt) - set(dt), key=int)))
        bad += 1
    for i in sorted(set(dt) & set(gt), key=int):
        if [[c.lower() for c in p] for p in dt[i]] != [[c.lower() for c in p] for p in gt[i]]:
            out.append("%s: corners of tag %s differ: crate %s golden %s" % (name, i, dt[i], gt[i]))
            bad += 1
            break
    return bad


def main(path):
    out, bad, seen = [], 0, 0
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        dump = json.loads(line)
        gpath = os.path.join(GOLDEN, "saddles_%s.json" % os.path.splitext(dump["image"])[0])
        if not os.path.exists(gpath):
            out.append("%s: no golden list" % dump["image"])
            bad += 1
            continue
        bad += compare(dump, json.load(open(gpath)), out)
        seen += 1
    print("\n".join(out) if out else "all fields of %d images agree with tests/golden" % seen)
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
