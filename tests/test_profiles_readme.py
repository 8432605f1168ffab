"""Evidence hygiene: every file under profiles/ is named in profiles/README.md -- literally, or by a
backticked pattern with `*` / `{a,b}` / `…` (e.g. `r4_box1_*`, `rejected/r4_sparse_flow_experiment.{patch,txt}`)."""
import fnmatch
import os
import re

from tests.util import ROOT


def _patterns(text):
    pats = set()
    for tok in re.findall(r"`([^`]+)`", text):
        tok = tok.strip().replace("…", "*")
        if tok.startswith("../"):
            tok = tok[3:]
        if " " in tok or not re.search(r"[A-Za-z0-9]", tok):
            continue
        # brace expansion, one level
        m = re.search(r"\{([^{}]*)\}", tok)
        alts = [tok[:m.start()] + a + tok[m.end():] for a in m.group(1).split(",")] if m else [tok]
        pats.update(alts)
    return pats


def test_every_profile_file_is_named_in_the_readme():
    prof = os.path.join(ROOT, "profiles")
    pats = _patterns(open(os.path.join(prof, "README.md")).read())
    files = []
    for d, _, fs in os.walk(prof):
        for f in fs:
            rel = os.path.relpath(os.path.join(d, f), prof)
            if rel != "README.md":
                files.append(rel)
    assert len(files) > 150
    missing = [f for f in files if not any(fnmatch.fnmatch(f, p) or fnmatch.fnmatch(f, "*" + p) for p in pats)]
    assert not missing, "files under profiles/ that profiles/README.md does not name: %s" % missing
