"""CPU tests of the host side: join-index construction, ABI surface, sharding and the rank merge."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle
from geneticscre_amd import api, dist, synth
from geneticscre_amd.uids import assemble_uids, build_level_tables, count_locations
from helpers import small_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_level_tables_match_reference_construction():
    """build_level_tables (vectorised) == getUidsCountsLocations + assemble_uids (R/PathMethods.R:133-152,
    src/wrapper.cpp:99-140) on the Appendix B network: 0->1(+), 0->2(-), 1->2(+), 2->0(+)."""
    src, trg, sign = np.array([0, 0, 1, 2]), np.array([1, 2, 2, 0]), np.array([1, -1, 1, 1])
    lv = build_level_tables(3, src, trg, sign)
    # level 2 / 3 / 4 / 5 rows exactly as recorded in SURVEY.md App. B (src:trg:count:location)
    want = {
        "2": [(0, 0, 2, 0), (1, 1, 1, 2), (2, 2, 1, 3)],
        "3": [(0, 1, 1, 2), (0, 2, 1, 3), (1, 2, 1, 3), (2, 0, 2, 0)],
        "4": [(0, 2, 1, 3), (0, 0, 2, 0), (1, 0, 2, 0), (2, 1, 1, 2), (2, 2, 1, 3)],
        "5": [(0, 2, 2, 3), (0, 0, 2, 0), (1, 0, 2, 0), (2, 1, 1, 2), (2, 2, 2, 3)],
    }
    for k, rows in want.items():
        u = lv.uids[k]
        assert list(zip(u.src.tolist(), u.trg.tolist(), u.count.tolist(), u.location.tolist())) == rows
    assert lv.uids["4"].signs.tolist() == [1, -1, 1, 1, -1]
    # the dictionary route used by the R code gives the same resolution
    cl = count_locations(trg, src)
    u3 = assemble_uids(3, src, trg, cl, sign)
    assert u3.count.tolist() == lv.uids["3"].count.tolist()
    assert u3.location.tolist() == lv.uids["3"].location.tolist()


def test_count_locations_missing_targets():
    """Targets without outgoing relations get (0, -1) (PathMethods.R:145-148); unknown keys resolve to (0, 0)."""
    cl = count_locations(np.array([5, 7, 9]), np.array([5, 5, 9]))
    assert cl == {5: (2, 0), 9: (1, 2), 7: (0, -1)}
    u = assemble_uids(2, [1, 1, 1, 1], [5, 7, 9, 11], cl, [1, 1, 1])
    assert u.count.tolist() == [2, 0, 1, 0] and u.location.tolist() == [0, -1, 2, 0]
    assert u.path_idx.tolist() == [0, 2, 2, 3, 3]


def test_values_table_properties():
    """-log two-sided hypergeometric p (R/Utils.R:137-159): the diagonal extremes are the most significant cells,
    p = 1 (value 0) at the mode, symmetric when nCases == nControls, no infinities left."""
    t = synth.values_table(12, 12)
    assert t.shape == (13, 13) and np.isfinite(t).all() and (t >= -1e-12).all()
    np.testing.assert_allclose(t, t.T, rtol=1e-12, atol=1e-12)
    assert t[0, 0] == pytest.approx(0.0, abs=1e-12)
    assert t[6, 6] == pytest.approx(0.0, abs=1e-9)          # mode of a symmetric hypergeometric
    assert t[12, 0] == t[0, 12] and t[12, 0] > t[6, 0] > t[3, 3]
    t2 = synth.values_table(5, 9)
    assert t2.shape == (6, 10) and np.isfinite(t2).all()


def test_case_or_control_and_packed_masks():
    rng = np.random.default_rng(5)
    pc = synth.case_or_control(11, 20, 9, rng)
    assert pc.shape == (9, 31) and set(np.unique(pc)) <= {0, 1}
    m = synth.masks_from_case_or_control(pc, 11)
    ex = oracle.OracleJoinExec(1, 11, 20, 9)
    ex.set_permuted_cases(pc)
    for r in range(9):
        np.testing.assert_array_equal(ex.perm_mask(r), m[r])
        assert sum(bin(int(w)).count("1") for w in m[r]) == 11     # every permutation keeps nCases cases
    strat = synth.case_or_control(11, 20, 4, rng, strata=np.arange(31) % 3)
    assert strat.shape == (4, 31)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports exactly what include/gcre_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "gcre_hip.h")).read()
    declared = set(re.findall(r"\b(gcre_[a-z0-9_]+)\s*\(", header))
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    lib = api.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gcre_abi_version() == 4
    out = subprocess.run(["nm", "-D", "--defined-only", api.lib_path()], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and "gcre_" in l}
    assert declared <= exported


def test_product_never_imports_the_oracle():
    """The product path must not route through the CPU oracle: no module of geneticscre_amd mentions it."""
    pkg = os.path.join(ROOT, "geneticscre_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "gcre_oracle" not in text, f


def test_resolve_count_locs_abi():
    oc, ol = api.resolve_count_locs([5, 7, 9, 11], [5, 9, 7], [2, 1, 0], [0, 2, -1])
    assert oc.tolist() == [2, 0, 1, 0] and ol.tolist() == [0, -1, 2, 0]


@pytest.mark.skipif(__import__("importlib").util.find_spec("torch") is None, reason="torch missing")
def test_no_gpu_means_loud_failure():
    """Without a device the product refuses to run instead of falling back to anything."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.GcreError, match="no HIP device"):
        api.JoinExec("method1", 5, 5, 3)
    with pytest.raises(ValueError):
        api.JoinExec("method1", 0, 5, 3)     # check_true(num_cases > 0 ...), join_base.cpp:47


def test_shard_uids_partitions_the_level():
    p = synth.make_problem(30, 90, 12, 10, 5, 4, seed=9, table=small_table(12, 10))
    u = p.levels.uids["4"]
    total = u.count_total_paths()
    seen = 0
    for r in range(3):
        b, e = dist.shard_bounds(total, r, 3)
        s = dist.shard_uids(u, b, e)
        assert s.count_total_paths() == e - b and len(s) == len(u)
        seen += e - b
    assert seen == total


def test_merge_topk_sentinel_and_ties():
    rows = np.array([[5.0, 3, 1, 1, 1], [5.0, 2, 9, 1, 1], [7.0, 8, 0, 2, 2], [-np.inf, -1, -1, 0, 0],
                     [-np.inf, -1, -1, 0, 0]])
    best = dist.merge_topk(rows, 2)
    assert best[:, 0].tolist() == [5.0, 7.0] and best[0, 1:3].tolist() == [2, 9]      # tie -> smaller (src, trg)
    best = dist.merge_topk(rows, 10)
    assert best[0].tolist() == [-np.inf, -1, -1, 0, 0] and len(best) == 4


def _rank_main(rank, world, port, method, tmp):
    import torch
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    p = synth.make_problem(40, 120, 21, 30, 37, 4, method=method, top_k=8, seed=13, table=small_table(21, 30, 4))
    ex = oracle.OracleJoinExec(method, p.n_cases, p.n_ctrls, p.iterations)
    ex.top_k = p.top_k
    ex.set_value_table(p.value_table)
    ex.set_permuted_cases(p.perm_cases)
    full = oracle.process_paths(p, order="canonical")
    lv = p.levels
    ops = {"2": (full["paths1"], ex.load(p.data1)[lv.data_inds["2"]]), "3": (full["paths2"], ex.load(p.data1)[lv.data_inds["3"]]),
           "4": (full["paths3"], full["paths2"])}
    ok = True
    for name, (p0, p1) in ops.items():
        u = lv.uids[name]
        b, e = dist.shard_bounds(u.count_total_paths(), rank, world)
        r = ex.join(dist.shard_uids(u, b, e), p0, p1, keep=False, order="canonical")
        null = torch.from_numpy(r.null.copy())
        best = dist.exchange_level(r.scores, r.src, r.trg, r.cases, r.ctrls, null, p.top_k, world)
        want = full[f"lst{name}"]
        ok &= np.array_equal(best[:, 0], want.scores) and np.array_equal(best[:, 1], want.src)
        ok &= np.array_equal(best[:, 2], want.trg) and np.array_equal(best[:, 3], want.cases)
        ok &= np.array_equal(null.numpy().view(np.uint32), want.null.view(np.uint32))
    open(os.path.join(tmp, f"rank{rank}.ok"), "w").write("1" if ok else "0")
    tdist.destroy_process_group()


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_two_rank_exchange_over_gloo(method, tmp_path):
    """world_size 2 on CPU: shard every level, score each shard (oracle), MAX all-reduce + top-k all-gather/merge
    exactly as bench.py does over RCCL -- the merged result must equal the unsharded one, ids included."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, port, method, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "rank0.ok").read() == "1" and open(tmp_path / "rank1.ok").read() == "1"


def _window_main(rank, world, port, tmp):
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank planned a different window from "its" free memory
    local = [6 * 2048, 3 * 2048][rank]
    w = dist.agree_window(local, world)
    K = 20_000
    wins = [(k0, min(K, k0 + w)) for k0 in range(0, K, w)]
    open(os.path.join(tmp, f"win{rank}.txt"), "w").write(repr(wins))
    tdist.destroy_process_group()


def test_ranks_with_different_plans_walk_the_same_windows(tmp_path):
    """Every (level, window) is one round of collectives (bench.py on_level): ranks whose gcre_plan_perm_window results
    differ must still walk identical window lists -- the smallest plan wins (dist.agree_window, MIN all-reduce)."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_window_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = open(tmp_path / "win0.txt").read(), open(tmp_path / "win1.txt").read()
    assert a == b and a.startswith("[(0, 6144), (6144, 12288)")


def test_r_shim_is_valid_c_against_the_r_api_declarations():
    """R is not in this image: the .Call shim is syntax- and type-checked against declarations of the R API functions it
    uses (tests/r_api_decls, signatures from "Writing R Extensions") and the real C ABI header."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
                        "-Werror=int-conversion", "-I", os.path.join(root, "tests", "r_api_decls"), "-I", os.path.join(root, "include"),
                        os.path.join(root, "geneticscre_amd", "csrc", "r_shim.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_native_level_tables_equal_python_restatement(seed):
    """gcre_build_levels (C++) == uids.build_level_tables (numpy) == the R construction, on random signed networks."""
    rng = np.random.default_rng(seed)
    g, src, trg, sign = synth.signed_network(60 + 40 * seed, 200 + 150 * seed, rng)
    a = build_level_tables(g, src, trg, sign)
    b = api.build_levels(g, src, trg, sign)
    for k in ("1a", "1b", "2", "3", "4", "5"):
        for f in ("src", "trg", "count", "location", "signs"):
            np.testing.assert_array_equal(getattr(a.uids[k], f), getattr(b.uids[k], f), err_msg=f"{k}.{f}")
        assert a.n_paths[k] == b.n_paths[k]
    for k in ("1a", "1b", "2", "3"):
        np.testing.assert_array_equal(a.data_inds[k], b.data_inds[k])
    for k in a.rels3:
        np.testing.assert_array_equal(a.rels3[k], b.rels3[k])
    with pytest.raises(ValueError):
        api.build_levels(g, src[::-1].copy(), trg[::-1].copy(), sign)          # not sorted


def test_native_values_table_is_the_published_algorithm():
    """gcre_values_table against the independent Python restatement of getValuesTable (R/Utils.R:137-159)
    over R's published dhyper (dbinom_raw / stirlerr / bd0) with the exact `<=` and R's long double sum: bit for bit."""
    from oracle import values_table as ovt
    for nc, nt in ((12, 12), (5, 9), (40, 33), (70, 60), (0, 4), (1, 1)):
        a, b = ovt.values_table(nc, nt), api.values_table(nc, nt)
        assert a.shape == b.shape and np.isfinite(b).all()
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (nc, nt)
    # dhyper itself against exact rational arithmetic: a handful of ulps, as Loader's algorithm promises
    from fractions import Fraction
    from math import comb
    for (x, r, b, n) in ((3, 12, 12, 7), (0, 70, 60, 9), (25, 70, 60, 65), (40, 70, 60, 40), (130, 200, 300, 260)):
        exact = Fraction(comb(r, x) * comb(b, n - x), comb(r + b, n))
        assert abs(ovt.dhyper(x, r, b, n) / float(exact) - 1.0) < 1e-13


def test_value_table_cells_that_hang_on_the_last_place_are_the_listed_ones():
    """With R's exact `<=` some cells depend on how outcomes that tie in exact arithmetic come out of dhyper; the header
    of oracle/values_table.py lists them for the 70 x 60 table.  (The earlier builder counted such pairs as ties with a
    1e-12 slack: those cells differed from R's by the mass of one outcome.)"""
    from oracle import values_table as ovt
    assert ovt.tie_sensitive_cells(70, 60) == ovt.TIE_SENSITIVE_70_60
    assert ovt.tie_sensitive_cells(12, 12) == []


def test_large_tables_sum_in_sorted_order_to_the_same_doubles_up_to_an_ulp(monkeypatch):
    """Past 2.5e11 inner steps the native builder sums the qualifying outcomes through a sorted long double prefix sum
    instead of R's index order: same set, same accumulator width; the rounded double may move in its last place."""
    exact = api.values_table(150, 130)
    monkeypatch.setenv("GCRE_VT_EXACT_WORK", "0")
    fast = api.values_table(150, 130)
    ulps = np.abs(exact.view(np.int64) - fast.view(np.int64))
    assert ulps.max() <= 4 and (ulps > 0).mean() < 0.05
