"""CPU tests of the oracle: the reference's own known-answer vectors, plus internal consistency checks.

The oracle (oracle/gcre_oracle.cpp) is the checker every GPU parity test leans on, so it is pinned first:
against the outputs the UNMODIFIED reference binary printed for SURVEY.md Appendix B (the only golden vectors
that exist for this path), and against an independent brute-force restatement written with Python sets.
"""
import json
import math
import os

import numpy as np
import pytest

import oracle
from geneticscre_amd.harness_io import read_problem, write_problem
from geneticscre_amd.synth import make_problem, masks_from_case_or_control
from helpers import small_table

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.load(open(os.path.join(GOLD, "appendix_b_expected.json")))["cases"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("nthreads", [0, 4])
def test_appendix_b_known_answers(case, nthreads):
    """Scores, ids (reference heap order, nthreads=0) and null maxima exactly as the reference printed them."""
    p = read_problem(os.path.join(GOLD, "appendix_b_tiny.txt"), method=case["method"], iterations=case["iterations"],
                     top_k=case["top_k"], path_length=case["path_length"])
    r = oracle.process_paths(p, order="reference", nthreads=nthreads)["lst4"]
    assert r.scores.tolist() == [float(s) for s in case["scores"]]
    assert r.null.tolist() == [float(v) for v in case["null"]]
    if nthreads == 0:   # ids of tied scores depend on thread timing
        assert list(zip(r.src.tolist(), r.trg.tolist())) == [tuple(i) for i in case["ids"]]


def brute_force_level(p, uids, rows0, rows1, method, n_cases, table, masks):
    """Independent restatement with Python sets: returns (scores, cases, ctrls, null) per joined path / permutation.

    rows are lists of (pos_set, neg_set) of patient indices."""
    T = np.full((p.n_cases + p.n_ctrls + 1,) * 2, -1.0)
    T[:table.shape[0], :table.shape[1]] = table
    Tmax = np.maximum(T, T.T)
    K = len(masks)
    null = np.zeros(K, dtype=np.float32)
    out = []
    for idx in range(len(uids)):
        for j in range(max(int(uids.count[idx]), 0)):
            loc = int(uids.location[idx]) + j
            pos0, neg0 = rows0[idx]
            pos1, neg1 = rows1[loc]
            if method == 2 and not uids.need_flip(idx, loc):
                pos1, neg1 = neg1, pos1
            pos, neg = pos0 | pos1, neg0 | neg1
            is_case = lambda c: c < n_cases
            if method == 1:
                cases = sum(1 for c in pos if is_case(c))
                ctrls = len(pos) - cases
                score = T[cases][ctrls]
                for r in range(K):
                    pc = len(pos & masks[r])
                    v = T[pc][len(pos) - pc]
                    if v > float(null[r]):
                        null[r] = np.float32(v)
            else:
                case_pos = sum(1 for c in pos if is_case(c))
                ctrl_neg = len(pos) - case_pos
                ctrl_pos = sum(1 for c in neg if is_case(c))
                case_neg = len(neg) - ctrl_pos
                score = T[case_pos][ctrl_neg] + T[case_neg][ctrl_pos]
                cases, ctrls = case_pos + case_neg, ctrl_pos + ctrl_neg
                for r in range(K):
                    a, b = len(pos & masks[r]), len(neg & masks[r])
                    v = Tmax[a][len(pos) - a] + Tmax[len(neg) - b][b]
                    if v > float(null[r]):
                        null[r] = np.float32(v)
            out.append((score, cases, ctrls, (pos, neg)))
    return out, null


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_against_set_based_brute_force(method):
    nc, nt, K = 9, 13, 11
    p = make_problem(14, 30, nc, nt, K, 4, method=method, top_k=7, seed=3, table=small_table(nc, nt, 1))
    M = 1 if method == "method1" else 2
    got = oracle.process_paths(p, order="canonical")
    packed = masks_from_case_or_control(p.perm_cases, nc)
    masks = [{c for c in range(nc + nt) if (int(packed[r][c // 64]) >> (c % 64)) & 1} for r in range(K)]
    data_rows = [({c for c in range(nc + nt) if p.data1[g][c]}, set()) for g in range(p.data1.shape[0])]
    lv = p.levels
    zero = [(set(), set())] * len(lv.uids["1a"])
    l1, _ = brute_force_level(p, lv.uids["1a"], zero, [data_rows[i] for i in lv.data_inds["1a"]], M, nc, p.value_table, masks)
    paths1 = [e[3] for e in l1]
    l2, n2 = brute_force_level(p, lv.uids["2"], paths1, [data_rows[i] for i in lv.data_inds["2"]], M, nc, p.value_table, masks)
    paths2 = [e[3] for e in l2]
    l3, n3 = brute_force_level(p, lv.uids["3"], paths2, [data_rows[i] for i in lv.data_inds["3"]], M, nc, p.value_table, masks)
    paths3 = [e[3] for e in l3]
    l4, n4 = brute_force_level(p, lv.uids["4"], paths3, paths2, M, nc, p.value_table, masks)
    for lst, (entries, null) in {"lst2": (l2, n2), "lst3": (l3, n3), "lst4": (l4, n4)}.items():
        r = got[lst]
        np.testing.assert_array_equal(r.all_scores, np.array([e[0] for e in entries]))
        np.testing.assert_array_equal(r.all_cases, np.array([e[1] for e in entries]))
        np.testing.assert_array_equal(r.all_ctrls, np.array([e[2] for e in entries]))
        np.testing.assert_array_equal(r.null.view(np.uint32), null.view(np.uint32))
        best = sorted((e[0] for e in entries), reverse=True)[:p.top_k]
        assert sorted(r.scores[r.src >= 0].tolist(), reverse=True) == best[:len(r.scores[r.src >= 0])]


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_orders_and_threads_agree_on_values(method):
    """Reference heap order vs canonical order: identical score multiset, null maxima and kept rows, any thread count."""
    p = make_problem(50, 140, 40, 33, 64, 5, method=method, top_k=15, seed=8, table=small_table(40, 33, 2))
    a = oracle.process_paths(p, order="reference", nthreads=0)
    b = oracle.process_paths(p, order="canonical", nthreads=3)
    for k in ("lst1", "lst2", "lst3", "lst4", "lst5"):
        np.testing.assert_array_equal(a[k].scores, b[k].scores)
        np.testing.assert_array_equal(a[k].null.view(np.uint32), b[k].null.view(np.uint32))
        assert sorted(zip(a[k].scores.tolist(), a[k].cases.tolist(), a[k].ctrls.tolist())) == \
               sorted(zip(b[k].scores.tolist(), b[k].cases.tolist(), b[k].ctrls.tolist()))
    for k in ("paths1", "paths2", "paths3"):
        np.testing.assert_array_equal(a[k], b[k])


def test_canonical_ties_prefer_smaller_path_ordinal():
    """A table with one value everywhere ties every path: the canonical top-k is the first k paths."""
    p = make_problem(20, 40, 10, 10, 4, 3, method="method1", top_k=5, seed=4, table=np.full((11, 11), 3.25))
    r = oracle.process_paths(p, order="canonical")["lst3"]
    u = p.levels.uids["3"]
    firsts = []
    for idx in range(len(u)):
        for j in range(int(u.count[idx])):
            firsts.append((idx, int(u.location[idx]) + j))
    want = firsts[:5][::-1]      # ascending output: best (smallest ordinal) last
    assert list(zip(r.src.tolist(), r.trg.tolist())) == want


def test_sentinel_when_fewer_paths_than_top_k():
    """a level with fewer than top_k scorable paths reports the {-inf,-1,-1,0,0} sentinel first."""
    p = make_problem(6, 8, 5, 5, 3, 2, method="method1", top_k=500, seed=1, table=small_table(5, 5))
    r = oracle.process_paths(p)["lst2"]
    assert r.scores[0] == -math.inf and r.src[0] == -1 and r.trg[0] == -1
    assert len(r.scores) == p.levels.n_paths["2"] + 1


def test_permutation_rows_reused_and_truncated():
    """join_base.cpp:89-90 (surplus rows dropped) and :116-123 (missing rows reused cyclically)."""
    rng = np.random.default_rng(0)
    rows = (rng.random((3, 20)) < 0.5).astype(np.int32)
    a = oracle.OracleJoinExec(1, 8, 12, 7)
    a.set_permuted_cases(rows)
    for r in range(7):
        np.testing.assert_array_equal(a.perm_mask(r), a.perm_mask(r % 3))
    b = oracle.OracleJoinExec(1, 8, 12, 2)
    b.set_permuted_cases(rows)
    np.testing.assert_array_equal(b.perm_mask(1), a.perm_mask(1))
    with pytest.raises(ValueError):
        oracle.OracleJoinExec(1, 8, 12, 2).set_permuted_cases(np.zeros((0, 0), np.int32))   # reference divides by zero
    with pytest.raises(ValueError):
        oracle.OracleJoinExec(1, 8, 12, 2).set_permuted_cases(rows[:, :5])                    # check_equal, :101


def test_join_assertions():
    """check_equal / check_index of join_base.cpp:196-200."""
    p = make_problem(10, 20, 6, 6, 2, 2, seed=2, table=small_table(6, 6))
    ex = oracle.OracleJoinExec(1, 6, 6, 2)
    ex.set_value_table(p.value_table)
    ex.set_permuted_cases(p.perm_cases)
    data = ex.load(p.data1)
    u = p.levels.uids["2"]
    with pytest.raises(ValueError):
        ex.join(u, data[:-1], data[p.levels.data_inds["2"]])          # uids.size() != paths0.size
    with pytest.raises(IndexError):
        ex.join(u, data, data[p.levels.data_inds["2"]][:-1])          # location + count - 1 >= paths1.size
    with pytest.raises(ValueError):
        oracle.OracleJoinExec(1, 0, 5, 1)                             # check_true(num_cases > 0 ...)


def test_harness_text_round_trip(tmp_path):
    p = make_problem(12, 25, 7, 6, 5, 5, method="method2", top_k=4, seed=6, table=small_table(7, 6))
    f = tmp_path / "dump.txt"
    write_problem(str(f), p)
    q = read_problem(str(f), method="method2", iterations=5, top_k=4)
    a, b = oracle.process_paths(p), oracle.process_paths(q)
    for k in ("lst1", "lst2", "lst3", "lst4", "lst5"):
        np.testing.assert_array_equal(a[k].scores, b[k].scores)
        np.testing.assert_array_equal(a[k].null, b[k].null)
        np.testing.assert_array_equal(a[k].src, b[k].src)


from helpers import fnv_rows, load_ref_cases  # noqa: E402

REF_CASES = load_ref_cases()


@pytest.mark.parametrize("name,p,exp", REF_CASES, ids=[c[0] for c in REF_CASES])
def test_oracle_matches_reference_scoring_code(name, p, exp):
    """Goldens printed by the reference's own score_permute / merge_scores / PathSet code (partial reference build,
    oracle/ref_partial): every level's scores (bit patterns), ids in the reference's heap order, counts, f32 null
    maxima and the kept path rows must match exactly."""
    got = oracle.process_paths(p, order="reference", nthreads=0)
    for lvl in range(1, p.path_length + 1):
        e, r = exp[f"lst{lvl}"], got[f"lst{lvl}"]
        assert [f"{int(b):016x}" for b in r.scores.view(np.uint64)] == e["scores"], (name, lvl)
        assert r.src.tolist() == e["src"] and r.trg.tolist() == e["trg"], (name, lvl)
        assert r.cases.tolist() == e["cases"] and r.ctrls.tolist() == e["ctrls"], (name, lvl)
        assert [f"{int(b):08x}" for b in r.null.view(np.uint32)] == e["null"], (name, lvl)
    for lvl, key in ((1, "lst1a"), (2, "lst2"), (3, "lst3")):
        if lvl <= p.path_length:
            assert fnv_rows(got[f"paths{lvl}"]) == exp[key]["kept_hash"], (name, key)
            assert len(got[f"paths{lvl}"]) == exp[key]["kept_rows"]


from helpers import WIDE_CASES, load_wide_case  # noqa: E402


@pytest.mark.parametrize("name", list(WIDE_CASES))
def test_oracle_matches_reference_scoring_code_at_baseline_widths(name):
    """The same, at the mask widths of BASELINE configs[1]-[3] (16, 79, 157 words) with the hypergeometric table, both
    methods: outputs of the reference's own scoring code on problems regenerated from their seed (only the outputs are
    committed; the inputs' SHA-256 is checked against the one recorded with the golden)."""
    p, exp = load_wide_case(name)
    got = oracle.process_paths(p, order="reference", nthreads=0)
    for lvl in range(1, p.path_length + 1):
        e, r = exp[f"lst{lvl}"], got[f"lst{lvl}"]
        assert [f"{int(b):016x}" for b in r.scores.view(np.uint64)] == e["scores"], (name, lvl)
        assert r.src.tolist() == e["src"] and r.trg.tolist() == e["trg"], (name, lvl)
        assert r.cases.tolist() == e["cases"] and r.ctrls.tolist() == e["ctrls"], (name, lvl)
        assert [f"{int(b):08x}" for b in r.null.view(np.uint32)] == e["null"], (name, lvl)
    for lvl, key in ((1, "lst1a"), (2, "lst2"), (3, "lst3")):
        assert fnv_rows(got[f"paths{lvl}"]) == exp[key]["kept_hash"], (name, key)


@pytest.mark.skipif(not (os.path.isdir("/root/reference/src") and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_driver"))),
                    reason="needs the reference tree and the partial reference build (build container only)")
def test_committed_goldens_regenerate(tmp_path):
    """tests/golden/make_ref_goldens.py into a temporary directory gives the committed files byte for byte: the fixtures
    cannot drift from their generator (round 2: a table builder changed after the goldens were cut and nothing noticed)."""
    import filecmp
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_goldens", os.path.join(ROOT, "tests", "golden", "make_ref_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from helpers import SLOW_WIDE_CASES
    slow = os.environ.get("GCRE_SLOW_GOLDENS") == "1"     # w313_m2: 230 s of reference time (helpers.SLOW_WIDE_CASES)
    mod.main(str(tmp_path), slow=slow)
    committed = os.path.join(ROOT, "tests", "golden", "ref_cases")
    names = sorted(n for n in os.listdir(committed) if slow or n[:-5] not in SLOW_WIDE_CASES)
    assert names == sorted(os.listdir(tmp_path))
    match, mismatch, errors = filecmp.cmpfiles(committed, str(tmp_path), names, shallow=False)
    assert not mismatch and not errors, (mismatch, errors)
