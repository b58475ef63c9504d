"""Shared comparison helpers for the parity tests."""
from __future__ import annotations

import numpy as np

from geneticscre_amd.synth import Problem, make_problem


def small_table(n_cases: int, n_ctrls: int, seed: int = 0) -> np.ndarray:
    """A cheap value table with many distinct f64 values whose f32 roundings differ from them."""
    rng = np.random.default_rng(seed)
    return rng.random((n_cases + 1, n_ctrls + 1)) * 20.0 + rng.random((n_cases + 1, n_ctrls + 1)) * 1e-7


def assert_same_result(got, want, check_ids: bool = True):
    """got: geneticscre_amd.api.JoinResult; want: oracle.OracleResult in canonical order.  Bit-exact."""
    assert got.scores.dtype == np.float64
    np.testing.assert_array_equal(got.scores.view(np.uint64), want.scores.view(np.uint64))
    np.testing.assert_array_equal(got.cases, want.cases)
    np.testing.assert_array_equal(got.ctrls, want.ctrls)
    if check_ids:
        np.testing.assert_array_equal(got.src, want.src)
        np.testing.assert_array_equal(got.trg, want.trg)
    assert got.null.dtype == np.float32
    np.testing.assert_array_equal(got.null.view(np.uint32), want.null.view(np.uint32))


def fnv_rows(rows: np.ndarray) -> str:
    """FNV-1a over the uint64 words of a kept path set, as oracle/ref_partial/ref_driver.cpp hashes them."""
    h = 1469598103934665603
    for w in np.ascontiguousarray(rows, dtype=np.uint64).ravel().tolist():
        h ^= w
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def load_ref_cases():
    """(name, Problem, expected dict) for every golden generated from the partial reference build."""
    import json
    import os
    from geneticscre_amd.harness_io import read_problem

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cases")
    out = []
    for name in json.load(open(os.path.join(d, "INDEX.json")))["cases"]:
        exp = json.load(open(os.path.join(d, name + ".json")))
        c = exp["_case"]
        p = read_problem(os.path.join(d, name + ".txt"), method=c["method"], iterations=c["iterations"],
                         top_k=c["top_k"], path_length=c["path_length"])
        out.append((name, p, exp))
    return out


# ---- goldens at BASELINE mask widths (tests/golden/ref_cases/w*.json) ------------------------------------------------
# The reference's own scoring code (oracle/ref_partial) run on problems too large to commit as text: 16, 79 and 157 mask
# words (BASELINE configs[1], [2], [3]), both methods, the hypergeometric value table.  Only the outputs are committed;
# the inputs are regenerated from the seed by the same generator call, and their SHA-256 is checked against the one
# recorded when the golden was cut, so a drifting generator fails loudly.
WIDE_CASES = {
    # name: method, genes, edges, cases, ctrls, permutations, length, top_k, seed
    "w16_m1": ("method1", 200, 700, 460, 540, 300, 4, 20, 211),
    "w16_m2": ("method2", 200, 700, 460, 540, 300, 4, 20, 212),
    "w79_m1": ("method1", 200, 700, 2400, 2600, 300, 4, 20, 213),
    "w79_m2": ("method2", 200, 700, 2400, 2600, 300, 4, 20, 214),
    "w157_m1": ("method1", 200, 700, 4300, 5700, 300, 4, 20, 215),
    "w157_m2": ("method2", 200, 700, 4300, 5700, 300, 4, 20, 216),
    # round 4: path length 5 (paths3 x paths3, src/wrapper.cpp:271-276) at the width of configs[2], both methods, and the
    # signed method at 313 words (20,000 patients: the sorted-prefix-sum form of the native table builder)
    "w79_m1_len5": ("method1", 110, 300, 2450, 2550, 300, 5, 20, 217),
    "w79_m2_len5": ("method2", 110, 300, 2450, 2550, 300, 5, 20, 218),
    "w313_m2": ("method2", 70, 200, 9400, 10600, 150, 4, 20, 219),
}
# goldens whose reference run takes minutes (the reference's signed method copies and symmetrises its padded (n+1)^2 table
# once per join, src/methods.h:128: 230 s at 20,000 patients): committed like the others, checked against the oracle and
# the GPU like the others, but cut again by test_committed_goldens_regenerate only under GCRE_SLOW_GOLDENS=1
SLOW_WIDE_CASES = {"w313_m2"}
_WIDE_TABLES: dict = {}
_WIDE_PROBLEMS: dict = {}


def wide_problem(name: str) -> Problem:
    """The inputs of a wide golden, rebuilt from its seed (cached per session; the 10,000-patient table takes ~15 s)."""
    if name not in _WIDE_PROBLEMS:
        from geneticscre_amd import api
        method, genes, edges, nc, nt, perms, length, top_k, seed = WIDE_CASES[name]
        if (nc, nt) not in _WIDE_TABLES:
            _WIDE_TABLES[(nc, nt)] = api.values_table(nc, nt)
        _WIDE_PROBLEMS[name] = make_problem(genes, edges, nc, nt, perms, length, method=method, top_k=top_k, seed=seed,
                                            table=_WIDE_TABLES[(nc, nt)])
    return _WIDE_PROBLEMS[name]


def load_wide_case(name: str):
    """(Problem, expected dict) of a wide golden; fails if the regenerated inputs are not the ones the golden was cut from."""
    import json
    import os
    from geneticscre_amd.harness_io import problem_digest
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cases")
    exp = json.load(open(os.path.join(d, name + ".json")))
    p = wide_problem(name)
    assert problem_digest(p) == exp["_case"]["input_sha256"], f"{name}: the generator no longer reproduces the golden's inputs"
    return p, exp
