"""Short form of tests/sensitivity_study.py (the stated tolerance against the real crate): the numpy
re-evaluation of rochade_refine reproduces the oracle bit for bit, the plausible stand-ins for faer's QR / LU move no
saddle by more than the stated tolerance and flip no decision, and tools/compare_crate_dump.py passes a dump that
deviates like that and fails one that deviates by more."""
import json
import os
import subprocess
import sys

import numpy as np

from tests.util import GOLDEN, ROOT, load_image


def test_p_mat_stand_ins_are_what_they_claim():
    from oracle import oracle as O
    from tests import sensitivity_study as S
    exact, cone = O.refine_constants(2)
    t = S.pmat_table(exact)
    # a float32 Householder QR lands within a few dozen ulp of the exact pseudo-inverse -- the size of deviation to expect of faer's
    assert 1 <= t["pmat:householder_f32"]["max_ulp_vs_exact"] <= 200
    assert t["pmat:ulp4_seed0"]["max_ulp_vs_exact"] == 4 and t["pmat:ulp1_seed0"]["max_ulp_vs_exact"] == 1
    assert np.array_equal(S.cone_kernel("fold").view(np.uint32), cone.view(np.uint32))


def test_variants_stay_within_the_stated_tolerance_and_flip_nothing():
    from tests import sensitivity_study as S
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import compare_crate_dump as CMP
    r = S.run(n_synth=2, jobs=1, names=["EuRoC.png", "1520525725372653511.png"])  # (asserts baseline == oracle, bit for bit)
    agg = r["aggregate"]
    assert agg["baseline"]["saddles"] > 800 and agg["baseline"]["tags"] >= 36 * 4
    for v in r["plausible_variants"]:
        a = agg[v]
        assert a["gained"] == 0 and a["lost"] == 0 and not any(a["flips"].values()), (v, a)
        assert a["frames_with_different_tag_ids"] == 0, v
        assert max(a["max_dx_px"], a["max_dy_px"], a.get("max_corner_px", 0.0)) <= CMP.TOL_PX / 4, (v, a)
        assert a["max_dk_rel"] <= CMP.TOL_K_REL / 4 and max(a["max_dtheta_deg"], a["max_dphi_deg"]) <= CMP.ANGLE_TOL_DEG / 2, (v, a)
    # the committed full run (9 images + 256 bench frames) says the same
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_sensitivity.json")))
    assert full["images"] == 265 and full["max_px_plausible_variants"] <= CMP.TOL_PX / 4
    for v in full["plausible_variants"]:
        a = full["aggregate"][v]
        assert a["gained"] == 0 and a["lost"] == 0 and a["frames_with_different_tag_ids"] == 0


def _dump_line(g, dx_ulp):
    """A crate dump for golden record g whose x coordinates are dx_ulp units in the last place off."""
    d = json.loads(json.dumps(g))
    d["saddles"]["x_bits"] = ["%08x" % (int(h, 16) + dx_ulp) for h in g["saddles"]["x_bits"]]
    return json.dumps(d)


def test_compare_crate_dump_applies_the_stated_tolerance(tmp_path):
    g = json.load(open(os.path.join(GOLDEN, "saddles_EuRoC.json")))
    tool = os.path.join(ROOT, "tools", "compare_crate_dump.py")
    for ulp, strict, want_rc, word in ((0, False, 0, "bit for bit"), (1, False, 0, "note"), (1, True, 1, "FAIL"), (64, False, 1, "FAIL")):
        p = tmp_path / ("dump_%d_%d.jsonl" % (ulp, strict))
        p.write_text(_dump_line(g, ulp) + "\n")
        r = subprocess.run([sys.executable, tool, str(p)] + (["--strict"] if strict else []), capture_output=True, text=True)
        assert r.returncode == want_rc and word in r.stdout, (ulp, strict, r.stdout, r.stderr)
