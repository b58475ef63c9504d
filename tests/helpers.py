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
