"""The oracle against the reference's own scoring code on seeded random problems (CPU; build container only: it needs
oracle/_ref/ref_driver, the partial build of /root/reference's src/methods.h + src/gcre_paths.h).  Same draw as
tests/test_gpu_fuzz.py; compared per level: every score (bit pattern), the ids in the reference's heap order, the counts, every f32 null
maximum and the hash of the kept path rows, as `ref_driver --bin` prints them.  A handful of cases by default, GCRE_FUZZ_CASES=300 for a long run."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle
from geneticscre_amd.harness_io import write_problem_bin
from geneticscre_amd.synth import make_problem
from helpers import fnv_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
N_CASES = min(int(os.environ.get("GCRE_FUZZ_CASES", "4")), 2000)

pytestmark = pytest.mark.skipif(not (os.path.isdir("/root/reference/src") and os.path.exists(BINARY)),
                                reason="needs the reference tree and the partial reference build (build container only)")


@pytest.mark.parametrize("case", range(N_CASES))
def test_oracle_matches_reference_code_on_random_problems(case, tmp_path):
    from test_gpu_fuzz import draw, value_table
    cfg, _ = draw(400000 + case)
    perms = max(1, min(cfg["perms"], 600))   # the reference's inline mode is one thread (K = 0 is a golden of its own)
    if cfg["n_cases"] + cfg["n_ctrls"] > 3000:
        perms = min(perms, 100)
    if (cfg["n_cases"] + cfg["n_ctrls"]) % 64 == 0:
        # PathSet::load asserts patients < 64 * width (src/gcre_paths.h:63).  The reference pads the width to its SIMD
        # width (src/join_base.cpp:15-23), so it refuses cohorts of exactly k * gs_vec_width patients; the driver's width is
        # ceil(n / 64) (the padding is layout only), so it refuses every multiple of 64: an error there, nothing to compare
        cfg["n_ctrls"] += 1
    p = make_problem(cfg["genes"], cfg["edges"], cfg["n_cases"], cfg["n_ctrls"], perms, cfg["length"], method=cfg["method"],
                     top_k=cfg["top_k"], seed=cfg["seed"], threshold=cfg["threshold"],
                     table=value_table(cfg["table"], cfg["n_cases"], cfg["n_ctrls"], cfg["seed"]))
    dump = str(tmp_path / "case.gcrebin")
    write_problem_bin(dump, p, nthreads=0)
    r = subprocess.run([BINARY, "--bin", dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    ref = json.loads(r.stdout)
    got = oracle.process_paths(p, order="reference", nthreads=0)
    for lvl in range(1, cfg["length"] + 1):
        e, g = ref[f"lst{lvl}"], got[f"lst{lvl}"]
        assert [f"{int(b):016x}" for b in g.scores.view(np.uint64)] == e["scores"], (case, lvl, cfg)
        assert g.src.tolist() == e["src"] and g.trg.tolist() == e["trg"], (case, lvl, cfg)
        assert g.cases.tolist() == e["cases"] and g.ctrls.tolist() == e["ctrls"], (case, lvl, cfg)
        assert [f"{int(b):08x}" for b in g.null.view(np.uint32)] == e["null"], (case, lvl, cfg)
    for lvl, key in ((1, "lst1a"), (2, "lst2"), (3, "lst3")):
        if lvl < cfg["length"]:
            assert fnv_rows(got[f"paths{lvl}"]) == ref[key]["kept_hash"], (case, key, cfg)
