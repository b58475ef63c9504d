"""The inclusion-exclusion null kernel (gcre_ie.hip): count planes, reduced-operand hints and pruned lookups must
leave every result bit-identical to the oracle, whatever combination of them a join ends up using."""
import numpy as np
import pytest

import oracle
from geneticscre_amd import api, dist
from geneticscre_amd.synth import make_problem
from helpers import assert_same_result, small_table

pytestmark = pytest.mark.gpu


def sparse_problem(method, seed, K=300, L=5, nc=310, nt=335, genes=70, edges=260):
    """Rare-variant shaped: 645 patients (11 words, padded to 12), every gene in <= 5 % of them."""
    return make_problem(genes, edges, nc, nt, K, L, method=method, top_k=15, seed=seed, threshold=0.05)


def check_levels(got, want, levels):
    for lvl in levels:
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


@pytest.mark.parametrize("method", ["method1", "method2"])
@pytest.mark.parametrize("prune", ["1", "0"])
def test_process_paths_runs_on_planes_with_verified_hints(method, prune, monkeypatch):
    """All six joins go through k_null_ie: hints verified, paths0 planes resident from level 2 on (by-product of
    the kept joins), and most joined paths scored as N0 + Nz - overlap."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_IE_PRUNE", prune)
    p = sparse_problem(method, 5)
    got = api.process_paths(p)
    want = oracle.process_paths(p, order="canonical")
    check_levels(got, want, range(1, 6))
    prof = got["profile"]
    assert prof["ie_launches"] >= 6
    assert prof["ie_hinted_joins"] == 6
    assert prof["ie_plane_joins"] == 6          # zero sets build theirs, kept sets inherit theirs
    total_lists = sum(p.levels.n_paths[k] for k in ("1a", "1b", "2", "3", "4", "5")) * (2 if method == "method2" else 1)
    # (signed method: one half of a gene row is empty -- an empty delta list, nothing to subtract)
    assert prof["ie_overlap_lists"] > (0.5 if method == "method1" else 0.25) * total_lists


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_auto_is_the_ie_kernel_on_rare_variant_data(method):
    p = sparse_problem(method, 6, K=2100, L=4)      # two permutation tiles, the second one partial
    got = api.process_paths(p)
    want = oracle.process_paths(p, order="canonical")
    check_levels(got, want, range(1, 5))
    assert got["profile"]["ie_launches"] >= 5 and got["profile"]["ie_overlap_lists"] > 0


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_wrong_hint_is_detected_and_ignored(method, monkeypatch):
    """A reduced operand that does not reproduce the joined rows must change nothing: the device check fails and the
    join runs on paths1 itself."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_CHUNK_PATHS", "256")       # the failure shows up in the first of several chunks
    p = sparse_problem(method, 7, K=200, L=4)
    full = oracle.process_paths(p, order="canonical")
    ex = api.JoinExec(method, p.n_cases, p.n_ctrls, p.iterations)
    ex.top_k = p.top_k
    ex.set_value_table(p.value_table)
    ex.set_permuted_cases(p.perm_cases)
    p3, p2 = ex.from_words(full["paths3"]), ex.from_words(full["paths2"])
    genes = ex.load(p.data1)
    du = api.DeviceUids(ex, p.levels.uids["4"])
    rng = np.random.default_rng(1)
    du.set_reduced(genes, rng.integers(0, genes.size, p2.size))     # garbage
    assert_same_result(ex.join(du, p3, p2), full["lst4"])
    assert ex.profile()["ie_hinted_joins"] == 0 and ex.profile()["ie_launches"] > 0
    real = np.asarray(p.levels.data_inds["3"], np.int64)             # the real one: the gene level 2 added at loc,
    if method == "method2":                                         # in the (-) half when its relation is negative
        real = real | ((np.asarray(p.levels.uids["2"].signs) != 1).astype(np.int64) << 31)
    du.set_reduced(genes, real)
    assert_same_result(ex.join(du, p3, p2), full["lst4"])
    assert ex.profile()["ie_hinted_joins"] == 1
    du.set_reduced(None)
    assert_same_result(ex.join(du, p3, p2), full["lst4"])
    assert ex.profile()["ie_hinted_joins"] == 0
    with pytest.raises(IndexError):
        du.set_reduced(genes, np.full(p2.size, genes.size))         # out of range
    ex.close()


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_wrong_hint_on_a_kept_level_changes_nothing(method, monkeypatch):
    """A garbage reduced operand on level 2 -- a join that keeps its rows, their count planes and (unsigned method)
    their recipe for level 3: detected on the device, the join runs on paths1 itself, the levels after it are built on
    what it really produced."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_CHUNK_PATHS", "512")
    p = sparse_problem(method, 23, K=140, L=4)
    want = oracle.process_paths(p, order="canonical")
    plan = api.ResidentPlan(p)
    rng = np.random.default_rng(2)
    n1 = plan.parsed[0].size
    plan.uids["2"].set_reduced(plan.parsed[0], rng.integers(0, n1, len(p.levels.data_inds["3"])))
    got = plan.run()
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4)):
        assert_same_result(got[name], want[f"lst{lvl}"])
    np.testing.assert_array_equal(plan.kept["2"].to_numpy(), want["paths2"])
    plan.close()


def test_planes_are_rebuilt_when_the_masks_change(monkeypatch):
    """Count planes belong to one set of permutation masks: new masks on the same context, same path sets."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    a = sparse_problem("method1", 8, K=150, L=3)
    b = sparse_problem("method1", 8, K=150, L=3)
    b.perm_cases = np.ascontiguousarray(a.perm_cases[::-1] ^ 1)     # different permutations, same data
    b.perm_cases[:, 0] = 1
    plan = api.ResidentPlan(a)
    first = plan.run()
    want_a = oracle.process_paths(a, order="canonical")
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3)):
        assert_same_result(first[name], want_a[f"lst{lvl}"])
    plan.ex.set_permuted_cases(b.perm_cases)
    second = plan.run()
    want_b = oracle.process_paths(b, order="canonical")
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3)):
        assert_same_result(second[name], want_b[f"lst{lvl}"])
    assert not np.array_equal(first["3"].null, second["3"].null)
    plan.close()


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_sharded_plan_keeps_planes_for_every_row(method, monkeypatch):
    """Two ranks' worth of shards on one GPU: kept levels are scored per shard but their count planes cover every
    row, so the next level still runs on resident planes; merged results equal the one-shot run."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_CHUNK_PATHS", "512")
    p = sparse_problem(method, 9, K=120, L=5)
    want = oracle.process_paths(p, order="canonical")
    parts = []
    for rank in range(2):
        plan = api.ResidentPlan(p)
        parts.append(plan.run(rank=rank, world=2))
        assert plan.last_profile["ie_plane_joins"] == 6
        plan.close()
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4), ("5", 5)):
        null = np.maximum(parts[0][name].null, parts[1][name].null)
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[f"lst{lvl}"]
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)


@pytest.mark.parametrize("method", ["method1", "method2"])
@pytest.mark.parametrize("world", [2, 5])
def test_ranks_keep_planes_only_for_the_rows_they_read(method, world, monkeypatch):
    """gcre_join_opts.keep_ranged = 2 at levels 1 and 2: every row is written (later joins read them as paths1 and as
    reduced operands) but a rank only computes count planes for the rows its own work on the next level reads as
    paths0.  Every join still runs on resident planes, the merged results equal the one-shot run."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_CHUNK_PATHS", "1024")
    p = sparse_problem(method, 11, K=150, L=4)
    want = oracle.process_paths(p, order="canonical")
    parts = []
    for rank in range(world):
        plan = api.ResidentPlan(p)
        n1, n2 = plan.needed_rows("1a", rank, world), plan.needed_rows("2", rank, world)
        assert n1 is not None and n2 is not None and plan.keep_mode("2") == 2 and plan.keep_mode("3") == 1
        assert n2[1] - n2[0] < plan.uids["2"].total_paths          # a strict part of level 2 gets planes
        parts.append(plan.run(rank=rank, world=world))
        assert plan.last_profile["ie_plane_joins"] == 5             # no join fell back to bit lists
        for lvl in ("1", "2"):                                      # ... and all rows of levels 1 and 2 are there
            np.testing.assert_array_equal(plan.kept[lvl].to_numpy(), want[f"paths{lvl}"])
        plan.close()
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4)):
        null = np.maximum.reduce([r[name].null for r in parts])
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[f"lst{lvl}"]
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)


def test_a_partial_set_of_planes_is_never_trusted_beyond_its_rows(monkeypatch):
    """A set whose planes cover a range only (keep_ranged = 2) and is then read in full -- as paths0 of an unsharded
    join -- gets its planes rebuilt: results equal the oracle's."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    p = sparse_problem("method1", 13, K=130, L=3)
    want = oracle.process_paths(p, order="canonical")
    plan = api.ResidentPlan(p)
    ex = plan.ex
    p0, p1, res = plan.operands("1a")
    total = plan.uids["1a"].total_paths
    ex.join(plan.uids["1a"], p0, p1, res, shard=(0, total // 3), keep=(0, total // 2), keep_mode=2)
    p0, p1, res = plan.operands("2")
    r = ex.join(plan.uids["2"], p0, p1, res)        # reads every row of level 1 as paths0
    np.testing.assert_array_equal(r.null.view(np.uint32), want["lst2"].null.view(np.uint32))
    np.testing.assert_array_equal(r.scores, want["lst2"].scores)
    plan.close()


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_many_permutation_tiles_few_segments(method, monkeypatch):
    """Ten 2048-permutation tiles (the last one partial) over joins of a few hundred segments: the work queues hand
    whole tiles to some XCDs and fractions to others, most waves steal, every wave changes tile several times."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    p = sparse_problem(method, 21, K=18500, L=4, genes=40, edges=110)
    got = api.process_paths(p)
    want = oracle.process_paths(p, order="canonical", nthreads=8)
    check_levels(got, want, range(1, 5))
    assert got["profile"]["ie_launches"] >= 5


def test_arbitrary_table_prunes_exactly(monkeypatch):
    """The pruning ladder makes no assumption about the table's shape: a table with random cells (not valley-shaped,
    with zeros, huge values and -1 padding) gives the oracle's maxima."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    nc, nt = 150, 170
    rng = np.random.default_rng(3)
    table = rng.random((nc + 1, nt + 1)) * rng.choice([0.0, 1.0, 40.0, 1e6], size=(nc + 1, nt + 1))
    table[rng.random(table.shape) < 0.05] = -1.0
    p = make_problem(50, 170, nc, nt, 400, 4, method="method1", top_k=7, seed=12, threshold=0.06, table=table)
    got = api.process_paths(p)
    want = oracle.process_paths(p, order="canonical")
    for lvl in range(1, 5):
        np.testing.assert_array_equal(got[f"lst{lvl}"].null.view(np.uint32), want[f"lst{lvl}"].null.view(np.uint32))
        np.testing.assert_array_equal(got[f"lst{lvl}"].scores, want[f"lst{lvl}"].scores)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_only_keep_the_rows_they_need(world, monkeypatch):
    """gcre_join_opts.keep_ranged: at level 3 a rank writes the rows its own level-4/5 shards read (plus its scored
    shard) and nothing else -- the merged results are unchanged and the rows outside stay untouched."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    p = sparse_problem("method1", 10, K=100, L=4)
    assert api.ResidentPlan(sparse_problem("method1", 10, K=10, L=5)).needed_rows("3", 0, 2) is None   # level 5 reads all of level 3
    want = oracle.process_paths(p, order="canonical")
    parts = []
    for rank in range(world):
        plan = api.ResidentPlan(p)
        need = plan.needed_rows("3", rank, world)
        b, e = plan.shard("3", rank, world)
        total = plan.uids["3"].total_paths
        assert 0 <= need[0] <= need[1] <= total and (need[1] - need[0]) < total     # a strict part of the level
        parts.append(plan.run(rank=rank, world=world))
        rows = plan.kept["3"].to_numpy()
        full = want["paths3"]
        inside = np.zeros(total, bool)
        inside[need[0]:need[1]] = True
        inside[b:e] = True
        np.testing.assert_array_equal(rows[inside], full[inside])
        assert not rows[~inside].any()                              # never written: still the zeros the set was created with
        plan.close()
    for name, lvl in (("3", 3), ("4", 4)):
        null = np.maximum.reduce([r[name].null for r in parts])
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[f"lst{lvl}"]
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)


@pytest.mark.parametrize("method,L", [("method1", 4), ("method1", 5), ("method2", 4)])
@pytest.mark.parametrize("world", [2, 5])
def test_pivot_group_shards_reproduce_the_full_join(method, L, world, monkeypatch):
    """GCRE_PIVOT_SHARDS=1: the last levels sharded by pivot group -- a join index with count = 0 outside the rank's groups
    (the reference skips such uids, src/join_base.cpp:236) -- instead of by ordinal range.  Every uid is scored by exactly
    one rank, the merged results equal the oracle's, ids included (a rank's paths stay in the level's order)."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_PIVOT_SHARDS", "1")
    p = sparse_problem(method, 12, K=120, L=L)
    want = oracle.process_paths(p, order="canonical")
    parts, covered = [], {}
    for rank in range(world):
        plan = api.ResidentPlan(p)
        assert plan.pivot_sharded(str(L), world) and plan.needed_rows("3", rank, world) is None
        parts.append(plan.run(rank=rank, world=world))
        for name in ("4", "5")[: L - 3]:
            covered[name] = covered.get(name, 0) + plan.pivot_uids(name, rank, world).total_paths
        plan.close()
    for name, total in covered.items():
        assert total == int(np.maximum(np.asarray(p.levels.uids[name].count), 0).sum())
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4), ("5", 5))[:L]:
        null = np.maximum.reduce([r[name].null for r in parts])
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[f"lst{lvl}"]
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)
        distinct = np.r_[True, w.scores[1:] != w.scores[:-1]] & np.r_[w.scores[1:] != w.scores[:-1], True]
        np.testing.assert_array_equal(best[distinct, 1].astype(np.int64), w.src[distinct], err_msg=name)
        np.testing.assert_array_equal(best[distinct, 2].astype(np.int64), w.trg[distinct], err_msg=name)


@pytest.mark.parametrize("method", ["method1", "method2"])
@pytest.mark.parametrize("tiles", ["1", "2"])
def test_permutation_windows_change_nothing(method, tiles, monkeypatch):
    """Large permutation counts run in windows of whole tiles (gcre_set_perm_window; the count planes are per tile).
    Forced down to 1- and 2-tile windows over 5000 permutations (3 tiles, the last one partial): gcre_process_paths
    and the resident plan -- sharded over two ranks as well -- return what a single window returns."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_WINDOW_TILES", tiles)
    p = sparse_problem(method, 13, K=5000, L=5, genes=40, edges=110)
    want = oracle.process_paths(p, order="canonical")
    got = api.process_paths(p)
    check_levels(got, want, range(1, 6))
    parts = []
    for rank in range(2):
        plan = api.ResidentPlan(p)
        assert plan.ex.plan_perm_window(10) == 2048 * int(tiles)
        parts.append(plan.run(rank=rank, world=2))
        plan.close()
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4), ("5", 5)):
        null = np.maximum(parts[0][name].null, parts[1][name].null)
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[f"lst{lvl}"]
        assert len(null) == 5000
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_windows_with_ranged_planes(method, monkeypatch):
    """One-tile permutation windows and three ranks at path length 4: every window rebuilds the count planes of levels
    1 and 2 for the rows the rank reads only (keep_ranged = 2), level 3 keeps a range of rows (keep_ranged = 1)."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_WINDOW_TILES", "1")
    p = sparse_problem(method, 17, K=4500, L=4, genes=40, edges=110)
    want = oracle.process_paths(p, order="canonical")
    parts = []
    for rank in range(3):
        plan = api.ResidentPlan(p)
        parts.append(plan.run(rank=rank, world=3))
        assert plan.last_profile["ie_plane_joins"] == 5 * 3        # five joins, three windows, all on resident planes
        plan.close()
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4)):
        null = np.maximum.reduce([r[name].null for r in parts])
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[f"lst{lvl}"]
        assert len(null) == 4500
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)


def test_perm_window_arguments():
    ex = api.JoinExec("method1", 40, 40, 5000)
    ex.set_perm_window(2048, 4096)
    ex.set_perm_window(4096, 5000)          # the last window may end at iterations
    for bad in ((100, 2048), (0, 3000), (4096, 2048), (0, 6000)):
        with pytest.raises(api.GcreError):
            ex.set_perm_window(*bad)
    ex.close()


@pytest.mark.parametrize("recipes", [False, True])
@pytest.mark.parametrize("method,density,n", [("method1", 0.45, 1600), ("method2", 0.45, 1600), ("method1", 0.3, 9000),
                                               ("method2", 0.3, 9000), ("method1", 0.02, 9000)])
def test_wide_counters_and_dense_rows(method, density, n, recipes, monkeypatch):
    """Every counter width of the IE kernels: dense rows over 1,600 patients (12 planes, gene planes of 3 groups, long
    overlap and delta lists through the overflow area) and over 9,000 patients (16 planes, up to 4 groups), plus rare
    variants over 9,000 patients (141 words per row: the inspector's 10-pass row loop).  With ``recipes`` no kept
    method-1 set whose paths0 had planes stores its own: the next level rebuilds every base row from the recipe,
    long lists included; small chunks make the recipe span several inspector passes."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    if recipes:
        if method == "method2":
            pytest.skip("the signed method stores planes")
        monkeypatch.setenv("GCRE_PLANES_OUT_MAX_MB", "0")
        monkeypatch.setenv("GCRE_CHUNK_PATHS", "192")
    nc = n // 2 - 37
    p = make_problem(24, 55, nc, n - nc, 130, 4, method=method, top_k=9, seed=21, threshold=0.9,
                     table=small_table(nc, n - nc, 4))
    rng = np.random.default_rng(6)
    p.data1 = (rng.random(p.data1.shape) < density).astype(np.int32)
    p.data2 = p.data1[p.levels.uids["1b"].src]
    got = api.process_paths(p)
    want = oracle.process_paths(p, order="canonical")
    check_levels(got, want, range(1, 5))
    assert got["profile"]["ie_launches"] >= 5 and got["profile"]["ie_hinted_joins"] == 5


@pytest.mark.parametrize("limit_mb", ["0", "100000"])
@pytest.mark.parametrize("world", [1, 2])
def test_recipes_replace_stored_planes(limit_mb, world, monkeypatch):
    """A kept method-1 set leaves the recipe of the join that made it; whether its planes are also stored (small sets)
    or rebuilt inside the next level's kernel from the recipe (GCRE_PLANES_OUT_MAX_MB=0: every set whose paths0 had
    stored planes) must not show in any result -- one device or two shards, with permutation windows on top."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_PLANES_OUT_MAX_MB", limit_mb)
    monkeypatch.setenv("GCRE_WINDOW_TILES", "1")
    p = sparse_problem("method1", 14, K=2300, L=5)
    want = oracle.process_paths(p, order="canonical")
    if world == 1:
        check_levels(api.process_paths(p), want, range(1, 6))
    parts = []
    for rank in range(world):
        plan = api.ResidentPlan(p)
        parts.append(plan.run(rank=rank, world=world))
        plan.close()
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4), ("5", 5)):
        null = np.maximum.reduce([r[name].null for r in parts])
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[f"lst{lvl}"]
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)


@pytest.mark.parametrize("method,seed", [("method1", 101), ("method1", 102), ("method2", 103)])
def test_three_kernel_forms_agree_at_scale(method, seed, monkeypatch):
    """Beyond what the CPU oracle finishes in seconds: 10^6 joined paths x 2,500 permutations, a hub gene
    included, scored by the IE kernels (with recipes), the delta-streaming kernel and the dense AND+popcount kernel.
    Three independent formulations of the same counts must return bit-identical results."""
    rng = np.random.default_rng(seed)
    p = make_problem(400, 3000, 330, 350, 2500, 5, method=method, top_k=40, seed=seed, threshold=0.05)
    hub = int(np.bincount(p.levels.uids["2"].count.astype(np.int64) > 0).argmax())   # any gene; its row becomes dense-ish
    p.data1[hub] = (rng.random(p.data1.shape[1]) < 0.04).astype(np.int32)
    p.data2 = p.data1[p.levels.uids["1b"].src]
    results = {}
    for form, env in (("ie", {"GCRE_NULL_KERNEL": "ie", "GCRE_PLANES_OUT_MAX_MB": "0"}),
                      ("sparse", {"GCRE_NULL_KERNEL": "sparse"}), ("dense", {"GCRE_NULL_KERNEL": "dense"})):
        for k in ("GCRE_NULL_KERNEL", "GCRE_PLANES_OUT_MAX_MB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        results[form] = api.process_paths(p)
    assert sum(p.levels.n_paths[k] for k in ("4", "5")) > 500000
    for lvl in range(1, 6):
        a = results["dense"][f"lst{lvl}"]
        for form in ("ie", "sparse"):
            b = results[form][f"lst{lvl}"]
            np.testing.assert_array_equal(a.null.view(np.uint32), b.null.view(np.uint32), err_msg=f"{form} L{lvl} null")
            np.testing.assert_array_equal(a.scores.view(np.uint64), b.scores.view(np.uint64), err_msg=f"{form} L{lvl}")
            np.testing.assert_array_equal(a.src, b.src)
            np.testing.assert_array_equal(a.trg, b.trg)


def test_freed_reduced_operand_is_not_used(monkeypatch):
    """A hint whose operand was freed in the meantime is dropped (the set is looked up by id, never dereferenced)."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    p = sparse_problem("method1", 15, K=150, L=4)
    full = oracle.process_paths(p, order="canonical")
    ex = api.JoinExec("method1", p.n_cases, p.n_ctrls, p.iterations)
    ex.top_k = p.top_k
    ex.set_value_table(p.value_table)
    ex.set_permuted_cases(p.perm_cases)
    p3, p2 = ex.from_words(full["paths3"]), ex.from_words(full["paths2"])
    genes = ex.load(p.data1)
    du = api.DeviceUids(ex, p.levels.uids["4"])
    du.set_reduced(genes, p.levels.data_inds["3"])
    assert_same_result(ex.join(du, p3, p2), full["lst4"])
    assert ex.profile()["ie_hinted_joins"] == 1
    genes.free()
    du._reduced = None
    junk = ex.create_path_set(genes.size)          # may well land where the freed set lived
    assert_same_result(ex.join(du, p3, p2), full["lst4"])
    assert ex.profile()["ie_hinted_joins"] == 0
    junk.free()
    ex.close()


@pytest.mark.parametrize("patients", [(2050, 2060), (3400, 3650), (4900, 4995)])
def test_inspector_row_widths_up_to_160_words(patients, monkeypatch):
    """The block-staged inspector handles rows of 3, 4 and 5 register blocks (65, 111 and 155 mask words here; 4 and 5
    run two waves per SIMD): oracle results for a kept join and the joins behind it."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    nc, nt = patients
    p = make_problem(45, 150, nc, nt, 120, 4, method="method1", top_k=10, seed=nc, table=small_table(nc, nt, 3))
    want = oracle.process_paths(p, order="canonical", nthreads=8)
    check_levels(api.process_paths(p), want, range(1, 5))
    plan = api.ResidentPlan(p)
    try:
        got = plan.run()
        for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4)):
            assert_same_result(got[name], want[f"lst{lvl}"])
        np.testing.assert_array_equal(plan.kept["3"].to_numpy(), want["paths3"])     # the rows the inspector wrote
        np.testing.assert_array_equal(plan.kept["2"].to_numpy(), want["paths2"])
    finally:
        plan.close()


@pytest.mark.parametrize("quad", ["0", "2"])
def test_bound_sum_past_the_counter_planes_is_not_inside(quad, monkeypatch):
    """The bound filter adds the base counts and the added row's counts, overlap counted twice: with dense rows that sum
    passes 2^L although no count of the joined path does (250 + 45 patients, carriers up to 40 %: 9 counter planes, sums up
    to 590).  The carry out of the top plane means "above hi"; dropped, the wrapped sum looked inside the interval and two
    null maxima of a level came out too low (found by tests/test_gpu_fuzz.py, case 5328)."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_IE_QUAD", quad)
    monkeypatch.setenv("GCRE_IE_WARM", "256")
    p = make_problem(80, 240, 250, 45, 257, 5, method="method1", top_k=40, seed=6328, threshold=0.4)
    want = oracle.process_paths(p, order="canonical")
    check_levels(api.process_paths(p), want, range(1, 6))
