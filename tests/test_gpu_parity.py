"""HIP path vs the CPU oracle through the C ABI, bit-exact (run on a real MI355X: pytest -m gpu)."""
import json
import os

import numpy as np
import pytest

import oracle
from geneticscre_amd import api
from geneticscre_amd.harness_io import read_problem
from geneticscre_amd.synth import make_problem
from helpers import assert_same_result, small_table

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_both(p):
    want = oracle.process_paths(p, order="canonical")
    got = api.process_paths(p)
    return got, want


def set_kernel(monkeypatch, kernel):
    """auto = inclusion-exclusion on count planes with pruned lookups (gcre_ie.hip), dense fallback by cost;
    ie-noprune looks every count up; sparse = bit-sliced delta streaming (gcre_sparse.hip); dense = AND+popcount."""
    if kernel == "ie-noprune":
        monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
        monkeypatch.setenv("GCRE_IE_PRUNE", "0")
    elif kernel in ("ie-quad", "ie-m1"):
        # the pruned method-1 launches in their four-paths-per-wave form (gcre_ieq.hip) wherever it can run, or never;
        # no warm-up slice, so that small joins reach the pruned kernels at all
        monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
        monkeypatch.setenv("GCRE_IE_QUAD", "2" if kernel == "ie-quad" else "0")
        monkeypatch.setenv("GCRE_IE_WARM", "0")
    else:
        monkeypatch.setenv("GCRE_NULL_KERNEL", kernel)


@pytest.mark.parametrize("case", json.load(open(os.path.join(GOLD, "appendix_b_expected.json")))["cases"],
                         ids=lambda c: c["name"])
def test_appendix_b_golden(case):
    """The reference's own printed outputs: score values and null maxima must match exactly;
    ids only where the score is not tied (App. A-9)."""
    p = read_problem(os.path.join(GOLD, "appendix_b_tiny.txt"), method=case["method"], iterations=case["iterations"],
                     top_k=case["top_k"], path_length=case["path_length"])
    got = api.process_paths(p)["lst4"]
    exp_scores = np.array([float(s) for s in case["scores"]])
    np.testing.assert_array_equal(got.scores, exp_scores)
    np.testing.assert_array_equal(got.null, np.array(case["null"], dtype=np.float32))
    ids = {(s, tuple(i)) for s, i in zip(exp_scores.tolist(), case["ids"])}
    for s, a, b in zip(got.scores.tolist(), got.src.tolist(), got.trg.tolist()):
        if (exp_scores == s).sum() == 1:
            assert (s, (a, b)) in ids


@pytest.mark.parametrize("kernel", ["auto", "ie", "ie-quad", "ie-m1", "ie-noprune", "sparse", "dense"])
@pytest.mark.parametrize("method", ["method1", "method2"])
@pytest.mark.parametrize("n_perm", [0, 3, 100, 130, 700])
def test_process_paths_matches_oracle(method, n_perm, kernel, monkeypatch):
    """Both forms of the null kernel (bit-sliced sparse: gcre_sparse.hip; dense AND+popcount: gcre_kernels.hip)."""
    set_kernel(monkeypatch, kernel)
    nc, nt = 37, 52   # patients not a multiple of 64, nCases != nControls
    p = make_problem(60, 150, nc, nt, n_perm, 5, method=method, top_k=9, seed=11 + n_perm,
                     table=small_table(nc, nt, 3))
    got, want = run_both(p)
    for lvl in range(1, 6):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_wide_masks_and_hypergeometric_table(method):
    """300 patients (5 words -> padded to 8), real -log hypergeometric table, top_k larger than some levels."""
    p = make_problem(40, 90, 140, 160, 257, 4, method=method, top_k=5000, seed=5)
    got, want = run_both(p)
    for lvl in range(1, 5):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


@pytest.mark.parametrize("kernel", ["auto", "ie", "ie-quad", "ie-m1", "ie-noprune", "sparse", "dense"])
@pytest.mark.parametrize("method,n_perm", [("method1", 1100), ("method2", 600), ("method1", 2500)])
def test_baseline_mask_width_full_parity(method, n_perm, kernel, monkeypatch):
    """BASELINE configs[2] geometry (5,000 patients = 79 mask words, real -log hypergeometric table, K not a
    multiple of the permutation tile) on a network small enough for the oracle: every level bit-exact."""
    set_kernel(monkeypatch, kernel)
    p = make_problem(220, 800, 2500, 2500, n_perm, 4, method=method, top_k=50, seed=77)
    want = oracle.process_paths(p, order="canonical", nthreads=8)
    got = api.process_paths(p)
    for lvl in range(1, 5):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])
    assert p.levels.n_paths["4"] > 5000


def test_full_size_properties_without_oracle():
    """At a size the oracle cannot finish in seconds, size-independent properties: (a) null maxima of the whole level
    equal the element-wise MAX over any partition of it, (b) top-k of the whole equals the merge of the parts,
    (c) p-values are monotone in the score, (d) a second run is bit-identical (no atomics-order dependence)."""
    from geneticscre_amd import dist
    p = make_problem(3000, 30000, 2500, 2500, 2000, 4, method="method1", top_k=64, seed=99)
    plan = api.ResidentPlan(p)
    a = plan.run()
    b = plan.run()
    for k in a:
        np.testing.assert_array_equal(a[k].null.view(np.uint32), b[k].null.view(np.uint32))
        np.testing.assert_array_equal(a[k].scores, b[k].scores)
        np.testing.assert_array_equal(a[k].src, b[k].src)
    parts = [plan.run(rank=r, world=4) for r in range(4)]
    for k in a:
        null = np.maximum.reduce([q[k].null for q in parts])
        np.testing.assert_array_equal(null.view(np.uint32), a[k].null.view(np.uint32))
        rows = np.vstack([np.stack([q[k].scores, q[k].src, q[k].trg, q[k].cases, q[k].ctrls], axis=1) for q in parts])
        best = dist.merge_topk(rows, p.top_k)
        np.testing.assert_array_equal(best[:, 0], a[k].scores)
        np.testing.assert_array_equal(best[:, 1], a[k].src)
        np.testing.assert_array_equal(best[:, 2], a[k].trg)
        pv = a[k].pvalues()
        assert (np.diff(pv) <= 0).all()        # ascending scores -> non-increasing p-values
    assert plan.uids["4"].total_paths > 200000


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_dense_rows_through_the_sparse_kernel(method, monkeypatch):
    """The sparse kernel must stay exact when the data is not sparse at all: half of all patients carry every gene
    (long bit lists, counter planes beyond 8, table diagonals too long for the LDS staging)."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "sparse")
    p = make_problem(30, 70, 700, 900, 300, 4, method=method, top_k=11, seed=88, threshold=0.6)
    rng = np.random.default_rng(5)
    p.data1 = (rng.random(p.data1.shape) < 0.45).astype(np.int32)
    p.data2 = p.data1[p.levels.uids["1b"].src]
    got, want = run_both(p)
    for lvl in range(1, 5):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


from helpers import fnv_rows, load_ref_cases  # noqa: E402

REF_CASES = load_ref_cases()


@pytest.mark.parametrize("kernel", ["auto", "ie", "ie-quad", "ie-m1", "ie-noprune", "sparse", "dense"])
@pytest.mark.parametrize("name,p,exp", REF_CASES, ids=[c[0] for c in REF_CASES])
def test_hip_matches_reference_scoring_code(name, p, exp, kernel, monkeypatch):
    """The HIP path against goldens printed by the reference's own scoring code (oracle/ref_partial): score values,
    counts and f32 null maxima bit-exact at every level; ids wherever the score is not tied (the reference's choice
    among ties is heap-order dependent)."""
    set_kernel(monkeypatch, kernel)
    got = api.process_paths(p)
    every = oracle.process_paths(p, order="canonical")       # only for "is this score tied among ALL paths?"
    for lvl in range(1, p.path_length + 1):
        e, r = exp[f"lst{lvl}"], got[f"lst{lvl}"]
        all_bits = [f"{int(b):016x}" for b in every[f"lst{lvl}"].all_scores.view(np.uint64)]
        assert [f"{int(b):016x}" for b in r.scores.view(np.uint64)] == e["scores"], (name, lvl)
        assert [f"{int(b):08x}" for b in r.null.view(np.uint32)] == e["null"], (name, lvl)
        assert sorted(zip(e["scores"], e["cases"], e["ctrls"])) == \
               sorted(zip([f"{int(b):016x}" for b in r.scores.view(np.uint64)], r.cases.tolist(), r.ctrls.tolist()))
        for k, s in enumerate(e["scores"]):
            if all_bits.count(s) == 1:      # a tie with a path outside the top-k also frees the choice
                assert (r.src[k], r.trg[k]) == (e["src"][k], e["trg"][k]), (name, lvl, k)


from helpers import WIDE_CASES, load_wide_case  # noqa: E402


@pytest.mark.parametrize("kernel", ["auto", "sparse"])
@pytest.mark.parametrize("name", list(WIDE_CASES))
def test_hip_matches_reference_scoring_code_at_baseline_widths(name, kernel, monkeypatch):
    """The HIP path against outputs of the reference's own scoring code at 16, 79 and 157 mask words (BASELINE
    configs[1]-[3]) with the hypergeometric table, both methods: reference code <-> GPU, not via the oracle."""
    set_kernel(monkeypatch, kernel)
    p, exp = load_wide_case(name)
    got = api.process_paths(p)
    every = oracle.process_paths(p, order="canonical", nthreads=8)       # only for "is this score tied among ALL paths?"
    for lvl in range(1, p.path_length + 1):
        e, r = exp[f"lst{lvl}"], got[f"lst{lvl}"]
        all_bits = [f"{int(b):016x}" for b in every[f"lst{lvl}"].all_scores.view(np.uint64)]
        assert [f"{int(b):016x}" for b in r.scores.view(np.uint64)] == e["scores"], (name, lvl)
        assert [f"{int(b):08x}" for b in r.null.view(np.uint32)] == e["null"], (name, lvl)
        # counts and ids of every score that is not tied among ALL paths of the level are the reference's; a tie -- also one
        # with a path outside the top-k (w313_m2, level 3: two paths with the same score and different counts at the cut) --
        # is cut by joined-path ordinal here and by heap arrival order there: those entries are the oracle's
        # under the canonical rule
        got_bits = [f"{int(b):016x}" for b in r.scores.view(np.uint64)]
        untied = [s for s in e["scores"] if all_bits.count(s) == 1]
        assert sorted((s, c, t) for s, c, t in zip(e["scores"], e["cases"], e["ctrls"]) if s in untied) == \
               sorted((s, c, t) for s, c, t in zip(got_bits, r.cases.tolist(), r.ctrls.tolist()) if s in untied), (name, lvl)
        for k, s in enumerate(e["scores"]):
            if all_bits.count(s) == 1:
                assert (r.src[k], r.trg[k]) == (e["src"][k], e["trg"][k]), (name, lvl, k)
        c = every[f"lst{lvl}"]
        assert r.src.tolist() == c.src.tolist() and r.trg.tolist() == c.trg.tolist(), (name, lvl)
        assert r.cases.tolist() == c.cases.tolist() and r.ctrls.tolist() == c.ctrls.tolist(), (name, lvl)


@pytest.mark.parametrize("name,p,exp", REF_CASES[:4], ids=[c[0] for c in REF_CASES[:4]])
def test_hip_kept_rows_match_reference(name, p, exp):
    """Rows written by the keep joins (levels 1a, 2, 3) hash to what the reference's PathSet held."""
    plan = api.ResidentPlan(p)
    plan.run()
    for lvl, key in ((1, "lst1a"), (2, "lst2"), (3, "lst3")):
        if lvl <= p.path_length:
            assert fnv_rows(plan.kept[str(lvl)].to_numpy()) == exp[key]["kept_hash"], (name, key)
