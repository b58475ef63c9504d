"""The inspection cache (gcre_set_inspect_cache): what a join computes before its permutation kernel -- the real-label half
of JoinExec::join, src/join_base.cpp:236-262 + methods.h:90-99 -- is computed for the first permutation window only;
later windows (and, when asked, later passes) start at the null kernel.  Results must be those of the oracle whether a
join was replayed or not, in every kernel form, sharded or not (pytest -m gpu)."""
import numpy as np
import pytest

import oracle
from geneticscre_amd import api, dist
from geneticscre_amd.synth import make_problem
from helpers import assert_same_result

pytestmark = pytest.mark.gpu

LEVELS = (("1b", "lst1"), ("2", "lst2"), ("3", "lst3"), ("4", "lst4"), ("5", "lst5"))


def problem(method, seed, K=5000, L=5):
    return make_problem(40, 110, 310, 335, K, L, method=method, top_k=15, seed=seed, threshold=0.05)


def check(got, want, L):
    for name, lst in LEVELS[:L]:
        assert_same_result(got[name], want[lst])


@pytest.fixture(autouse=True)
def _no_chain_unless_asked(monkeypatch):
    """The replay counts below are those of the cache alone: the launch-ahead chain (on by default for plans of this size)
    replays the first window's joins too.  The chain's own tests take the variable out again."""
    monkeypatch.setenv("GCRE_AHEAD", "0")


@pytest.mark.parametrize("kernel", ["auto", "ie", "sparse", "dense"])
@pytest.mark.parametrize("method", ["method1", "method2"])
def test_later_windows_start_at_the_null_kernel(method, kernel, monkeypatch):
    """Three one-tile windows over 5000 permutations: 6 joins, each inspected once; windows 2 and 3 replay.  A second
    pass forgets the inspections by default (it does all the work again) and keeps them when asked to."""
    if kernel != "auto":
        monkeypatch.setenv("GCRE_NULL_KERNEL", kernel)
    monkeypatch.setenv("GCRE_WINDOW_TILES", "1")
    p = problem(method, 21)
    want = oracle.process_paths(p, order="canonical")
    plan = api.ResidentPlan(p)
    try:
        joins = len(plan.names)
        plan.set_window(2048)
        got = plan.run()
        check(got, want, 5)
        assert plan.last_profile["inspect_replays"] == 2 * joins, plan.last_profile
        first_stats = plan.last_profile["stats_kernel_ms"]
        assert first_stats > 0
        # default: a new pass inspects again (a bench step never reuses the previous step's work)
        got = plan.run()
        check(got, want, 5)
        assert plan.last_profile["inspect_replays"] == 2 * joins
        # steady state over resident inputs: nothing but null kernels
        got = plan.run(keep_inspections=True)
        check(got, want, 5)
        assert plan.last_profile["inspect_replays"] == 2 * joins      # this pass filled the cache it keeps
        got = plan.run(keep_inspections=True)
        check(got, want, 5)
        assert plan.last_profile["inspect_replays"] == 3 * joins, plan.last_profile
        assert plan.last_profile["stats_kernel_ms"] == 0
        # and a pass without the flag starts from scratch again
        got = plan.run()
        check(got, want, 5)
        assert plan.last_profile["inspect_replays"] == 2 * joins
    finally:
        plan.close()


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_single_window_pass_can_be_kept_too(method, monkeypatch):
    """One window: the cache is off unless asked for; with keep_inspections the second pass replays every join."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    p = problem(method, 22, K=2300, L=4)
    want = oracle.process_paths(p, order="canonical")
    plan = api.ResidentPlan(p)
    try:
        got = plan.run()
        check(got, want, 4)
        assert plan.last_profile["inspect_replays"] == 0
        got = plan.run(keep_inspections=True)
        check(got, want, 4)
        assert plan.last_profile["inspect_replays"] == 0
        got = plan.run(keep_inspections=True)
        check(got, want, 4)
        assert plan.last_profile["inspect_replays"] == len(plan.names)
        assert plan.last_profile["stats_kernel_ms"] == 0
        got = plan.run()       # cache off again: buffers released, everything recomputed
        check(got, want, 4)
        assert plan.last_profile["inspect_replays"] == 0
    finally:
        plan.close()


def test_a_new_value_table_or_top_k_invalidates_the_cache(monkeypatch):
    """Observed scores and the top-k depend on the table and on k: a kept inspection is not reused across them."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    p = problem("method1", 23, K=2100, L=3)
    plan = api.ResidentPlan(p)
    try:
        plan.run(keep_inspections=True)
        plan.run(keep_inspections=True)
        assert plan.last_profile["inspect_replays"] == len(plan.names)
        table = np.array(p.value_table, dtype=np.float64, copy=True)
        table[table > 0] *= 0.5
        plan.ex.set_value_table(table)
        p2 = problem("method1", 23, K=2100, L=3)
        p2.value_table = table
        want = oracle.process_paths(p2, order="canonical")
        got = plan.run(keep_inspections=True)
        assert plan.last_profile["inspect_replays"] == 0
        check(got, want, 3)
        plan.ex.top_k = 7
        p2.top_k = 7
        want = oracle.process_paths(p2, order="canonical")
        got = plan.run(keep_inspections=True)
        assert plan.last_profile["inspect_replays"] == 0
        check(got, want, 3)
    finally:
        plan.close()


@pytest.mark.parametrize("world", [1, 2])
def test_a_short_last_window_changes_the_path_tile_not_the_cached_rows(world, monkeypatch):
    """K = 2300 in one-tile windows: the second window has 252 permutations, for which the dense kernel's path tile -- and
    with it the padded size of every per-chunk buffer -- is larger.  A replayed chunk's buffers must grow with their
    contents (they were once simply re-allocated: stale row numbers, a memory fault)."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_PLANES_OUT_MAX_MB", "0")
    monkeypatch.setenv("GCRE_WINDOW_TILES", "1")
    p = make_problem(70, 260, 310, 335, 2300, 5, method="method1", top_k=15, seed=14, threshold=0.05)
    want = oracle.process_paths(p, order="canonical")
    parts = []
    for rank in range(world):
        plan = api.ResidentPlan(p)
        parts.append(plan.run(rank=rank, world=world))
        assert plan.last_profile["inspect_replays"] == len(plan.names)
        plan.close()
    for name, lst in LEVELS:
        null = np.maximum.reduce([r[name].null for r in parts])
        np.testing.assert_array_equal(null.view(np.uint32), want[lst].null.view(np.uint32), err_msg=name)


@pytest.mark.parametrize("chunk", ["256", "0"])
def test_windows_of_sharded_chunked_joins_replay(chunk, monkeypatch):
    """Two ranks, joins cut into several chunks (GCRE_CHUNK_PATHS) and two-tile windows: every chunk of every join is
    replayed in the second window, merged results are the oracle's."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_WINDOW_TILES", "2")
    if chunk != "0":
        monkeypatch.setenv("GCRE_CHUNK_PATHS", chunk)
    p = problem("method1", 24, K=5000, L=4)
    want = oracle.process_paths(p, order="canonical")
    parts = []
    for rank in range(2):
        plan = api.ResidentPlan(p)
        parts.append(plan.run(rank=rank, world=2))
        assert plan.last_profile["inspect_replays"] >= len(plan.names)
        plan.close()
    for name, lst in LEVELS[:4]:
        null = np.maximum(parts[0][name].null, parts[1][name].null)
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[lst]
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_process_paths_windows_inspect_once(method, monkeypatch):
    """The one-shot call with forced one-tile windows: oracle results, operands made once."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_WINDOW_TILES", "1")
    p = problem(method, 25, K=4500, L=5)
    want = oracle.process_paths(p, order="canonical")
    got = api.process_paths(p)
    for lvl in range(1, 6):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_kept_inspections_without_permutations_and_with_a_sentinel(method):
    """iterations = 0 (no null kernel at all, App. A-3) and top_k above the path count (the sentinel, App. A-8): a replayed
    pass returns the same lists as the inspected one and as the oracle."""
    p = make_problem(12, 20, 9, 9, 0, 3, method=method, top_k=400, seed=51)
    want = oracle.process_paths(p, order="canonical")
    plan = api.ResidentPlan(p)
    try:
        for rep in range(3):
            got = plan.run(keep_inspections=True)
            check(got, want, 3)
            assert plan.last_profile["inspect_replays"] == (0 if rep == 0 else len(plan.names))
            assert all(len(got[name].null) == 0 for name, _ in LEVELS[:3])
    finally:
        plan.close()


@pytest.mark.parametrize("method", ["method1", "method2"])
@pytest.mark.parametrize("kernel,chunk,window", [("", "", ""), ("ie", "900", ""), ("ie", "", "1"), ("sparse", "", ""), ("dense", "", "")])
def test_inspect_ahead_gives_the_same_results(method, kernel, chunk, window, monkeypatch):
    """gcre_join_ahead: every join's inspector runs on a stream of its own beside the permutation kernel of the join before it,
    into the next join's inspection cache; the next join starts at its null kernel.  The default for plans of up to 64 M joined
    paths (GCRE_AHEAD=0 turns it off, =1 on for any size), exact like everything else: single and several chunks per join, two
    permutation windows, the kernel forms that do not run the inclusion-exclusion inspector, a sharded pass."""
    monkeypatch.delenv("GCRE_AHEAD", raising=False)
    for k, v in (("GCRE_NULL_KERNEL", kernel), ("GCRE_CHUNK_PATHS", chunk), ("GCRE_WINDOW_TILES", window)):
        if v:
            monkeypatch.setenv(k, v)
    p = make_problem(90, 330, 61, 70, 2300, 5, method=method, top_k=13, seed=77)
    want = oracle.process_paths(p, order="canonical", nthreads=4)
    plan = api.ResidentPlan(p)
    try:
        assert plan.ahead
        for _ in range(2):                     # the second pass starts from forgotten inspections again
            got = plan.run()
            for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4), ("5", 5)):
                assert_same_result(got[name], want[f"lst{lvl}"])
        if not chunk and kernel in ("", "ie"):
            assert plan.last_profile["inspect_replays"] >= 4     # levels 1b..5 started at their null kernels
        parts = [plan.run(rank, 3) for rank in range(3)]         # shards: the ahead inspection carries the shard's key
        for name, lvl in (("1b", 1), ("4", 4), ("5", 5)):
            null = np.maximum.reduce([q[name].null for q in parts])
            assert np.array_equal(null.view(np.uint32), want[f"lst{lvl}"].null.view(np.uint32)), (name,)
    finally:
        plan.close()


def test_inspect_ahead_switch(monkeypatch):
    """GCRE_AHEAD=0: no chain is registered, every join runs whole -- same results."""
    monkeypatch.setenv("GCRE_AHEAD", "0")
    p = make_problem(90, 330, 61, 70, 300, 4, method="method1", top_k=13, seed=78)
    want = oracle.process_paths(p, order="canonical", nthreads=4)
    plan = api.ResidentPlan(p)
    try:
        assert not plan.ahead
        got = plan.run()
        for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4)):
            assert_same_result(got[name], want[f"lst{lvl}"])
        assert plan.last_profile["inspect_replays"] == 0
    finally:
        plan.close()
