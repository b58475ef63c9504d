"""JoinExec / PathSet level tests of the HIP library against the oracle (pytest -m gpu, real MI355X)."""
import math

import numpy as np
import pytest

import oracle
from geneticscre_amd import api, dist, synth
from geneticscre_amd.synth import make_problem
from helpers import assert_same_result, small_table

pytestmark = pytest.mark.gpu


def make_pair(p):
    ex = api.JoinExec(p.method, p.n_cases, p.n_ctrls, p.iterations)
    ex.top_k = p.top_k
    ex.set_value_table(p.value_table)
    ex.set_permuted_cases(p.perm_cases)
    ox = oracle.OracleJoinExec(p.method, p.n_cases, p.n_ctrls, p.iterations)
    ox.top_k = p.top_k
    ox.set_value_table(p.value_table)
    ox.set_permuted_cases(p.perm_cases)
    return ex, ox


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_load_select_and_kept_rows_round_trip(method):
    """PathSet::load / select (gcre_paths.h:56-92) and the rows a keep-join writes (join_base.cpp:243-249)."""
    p = make_problem(35, 80, 70, 61, 20, 3, method=method, top_k=6, seed=21, table=small_table(70, 61, 5))
    ex, ox = make_pair(p)
    d = ex.load(p.data1)
    want = ox.load(p.data1)
    np.testing.assert_array_equal(d.to_numpy(), want)
    idx = p.levels.data_inds["3"]
    np.testing.assert_array_equal(d.select(idx).to_numpy(), want[idx])
    np.testing.assert_array_equal(ex.from_words(want).to_numpy(), want)
    # keep joins of levels 1a -> 2 -> 3: rows must be bit-identical, including the (+)/(-) halves of method 2
    lv = p.levels
    full = oracle.process_paths(p, order="canonical")
    z = ex.create_path_set(len(lv.data_inds["1a"]))
    k1 = ex.create_path_set(lv.n_paths["1a"])
    ex.join(lv.uids["1a"], z, d.select(lv.data_inds["1a"]), k1)
    np.testing.assert_array_equal(k1.to_numpy(), full["paths1"])
    k2 = ex.create_path_set(lv.n_paths["2"])
    r2 = ex.join(lv.uids["2"], k1, d.select(lv.data_inds["2"]), k2)
    np.testing.assert_array_equal(k2.to_numpy(), full["paths2"])
    assert_same_result(r2, full["lst2"])
    k3 = ex.create_path_set(lv.n_paths["3"])
    r3 = ex.join(lv.uids["3"], k2, d.select(lv.data_inds["3"]), k3)
    np.testing.assert_array_equal(k3.to_numpy(), full["paths3"])
    assert_same_result(r3, full["lst3"])


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_shards_and_chunks_reproduce_the_full_join(method, monkeypatch):
    """Scoring a level in 3 shards, with the device chunk forced down to 64 paths, then merging the way bench.py
    does (MAX of null maxima, top-k merge) equals the one-shot result -- ids included."""
    monkeypatch.setenv("GCRE_CHUNK_PATHS", "64")
    p = make_problem(45, 130, 33, 48, 150, 4, method=method, top_k=10, seed=31, table=small_table(33, 48, 6))
    ex, ox = make_pair(p)
    full = oracle.process_paths(p, order="canonical")
    p3, p2 = ex.from_words(full["paths3"]), ex.from_words(full["paths2"])
    u = p.levels.uids["4"]
    du = api.DeviceUids(ex, u)
    whole = ex.join(du, p3, p2)
    assert_same_result(whole, full["lst4"])
    total = du.total_paths
    null = np.zeros(p.iterations, np.float32)
    rows = []
    for r in range(3):
        b, e = dist.shard_bounds(total, r, 3)
        part = ex.join(du, p3, p2, shard=(b, e))
        null = np.maximum(null, part.null)
        rows.append(np.stack([part.scores, part.src, part.trg, part.cases, part.ctrls], axis=1))
    best = dist.merge_topk(np.vstack(rows), p.top_k)
    np.testing.assert_array_equal(best[:, 0], full["lst4"].scores)
    np.testing.assert_array_equal(best[:, 1], full["lst4"].src)
    np.testing.assert_array_equal(best[:, 2], full["lst4"].trg)
    np.testing.assert_array_equal(null.view(np.uint32), full["lst4"].null.view(np.uint32))
    empty = ex.join(du, p3, p2, shard=(5, 5))     # an empty shard is legal: sentinel only, null maxima all zero
    assert empty.scores.tolist() == [-math.inf] and not empty.null.any()


def test_massive_ties_cut_in_path_order():
    """Every path scores the same: the selected top-k must be the k smallest joined-path ordinals."""
    p = make_problem(60, 200, 16, 16, 10, 4, method="method1", top_k=37, seed=41, table=np.full((17, 17), 2.5))
    got = api.process_paths(p)
    want = oracle.process_paths(p, order="canonical")
    for lvl in (2, 3, 4):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])
    assert p.levels.n_paths["4"] > 2000      # the tie cut really spans several 1024-entry chunks


def test_sentinel_zero_permutations_and_row_reuse():
    """top_k above the path count -> sentinel (App. A-8); iterations = 0 -> empty TestScores (App. A-3);
    fewer permutation rows than iterations -> cyclic reuse, more -> truncation (join_base.cpp:89-123)."""
    p = make_problem(12, 20, 9, 9, 0, 3, method="method1", top_k=400, seed=51, table=small_table(9, 9))
    got, want = api.process_paths(p), oracle.process_paths(p, order="canonical")
    for lvl in (1, 2, 3):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])
        assert got[f"lst{lvl}"].scores[0] == -math.inf and len(got[f"lst{lvl}"].null) == 0
        assert np.isnan(got[f"lst{lvl}"].pvalues()).all()        # 0/0 in R, ProcessPaths.R:316
    q = make_problem(12, 20, 9, 9, 4, 3, method="method2", top_k=5, seed=52, table=small_table(9, 9))
    for iters in (9, 2):
        q.iterations = iters
        got, want = api.process_paths(q), oracle.process_paths(q, order="canonical")
        for lvl in (1, 2, 3):
            assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


def test_packed_masks_equal_int_matrix():
    p = make_problem(25, 60, 40, 45, 90, 3, method="method1", top_k=4, seed=61, table=small_table(40, 45))
    a = api.ResidentPlan(p).run()
    b = api.ResidentPlan(p, packed_masks=synth.masks_from_case_or_control(p.perm_cases, p.n_cases)).run()
    for k in a:
        np.testing.assert_array_equal(a[k].null.view(np.uint32), b[k].null.view(np.uint32))
        np.testing.assert_array_equal(a[k].scores, b[k].scores)


def test_reference_assertions_surface_as_errors():
    """check_equal / check_index / check_true of the reference (gcre_types.h:58-76) become ValueError / IndexError."""
    p = make_problem(10, 20, 6, 6, 2, 2, seed=2, table=small_table(6, 6))
    ex, _ = make_pair(p)
    data = ex.load(p.data1)
    u = p.levels.uids["2"]
    with pytest.raises(ValueError):
        ex.join(u, data.select(np.arange(data.size - 1)), data.select(p.levels.data_inds["2"]))   # uids.size() != paths0.size
    with pytest.raises(IndexError):
        ex.join(u, data, data.select(p.levels.data_inds["2"][:-1]))                                # location out of range
    with pytest.raises(ValueError):
        ex.join(u, data, data.select(p.levels.data_inds["2"]), ex.create_path_set(3))              # paths_res.size mismatch
    with pytest.raises(IndexError):
        data.select([0, data.size])                                                                # PathSet::select check_index
    with pytest.raises(ValueError):
        api.JoinExec("method1", 4, 0, 1)                                                           # check_true(num_ctrls > 0)
    with pytest.raises(ValueError):
        ex.set_permuted_cases(p.perm_cases[:, :5])                                                 # check_equal on columns
    fresh = api.JoinExec("method1", 6, 6, 2)
    with pytest.raises(ValueError):
        fresh.join(u, fresh.create_path_set(len(u)), fresh.create_path_set(30))                    # table / masks not set


def test_r_list_shape_and_pvalues():
    """make_score_list (wrapper.cpp:142-174): ids are 1-based (idx+1, loc+1); p-value = #(TestScores >= score)/K with
    the f64 score compared against f32-rounded maxima (ProcessPaths.R:316)."""
    p = make_problem(30, 70, 20, 20, 50, 3, method="method1", top_k=5, seed=71)
    r = api.process_paths(p)["lst3"]
    lst = r.as_r_list()
    assert lst["ids"].shape == (5, 2) and (lst["ids"][:, 0] == r.src + 1).all()
    assert lst["debug"][0] == f"[debug] {r.src[0]}:{r.trg[0]} {r.cases[0]}/{r.ctrls[0]}"
    pv = r.pvalues()
    want = [(r.null.astype(np.float64) >= s).mean() for s in r.scores]
    np.testing.assert_array_equal(pv, np.array(want))
    assert (np.diff(r.scores) >= 0).all()


def test_bench_two_ranks_reproduce_one_rank(tmp_path):
    """bench.py's N > 1 path rehearsed on the one GPU of the test box: two ranks (gloo collectives, both on cuda:0)
    shard every level, exchange null maxima and top-k tables, and must reproduce the one-rank results bit for bit."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # GCRE_EXCHANGE_UNIT: the ranks also share their running maxima inside the joins (every join above ~2000 path-tiles)
    env = dict(os.environ, GCRE_BENCH_DUMP="1", GCRE_QUIET="1", GCRE_EXCHANGE_UNIT="1000")
    common = ["--no-cpu-baseline", "--no-steady-state", "--steps", "1", "--warmup", "0", "--config", "subgraph", "--edges", "20000",
              "--perms", "2500", "--top-k", "25"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *common], capture_output=True, text=True,
                         env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--scaling", "strong", *common], capture_output=True, text=True,
                         env=env, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert a["result_sha256"] == b["result_sha256"]
    assert a["config"]["scores_per_step"] == b["config"]["scores_per_step"]


def _mix64(z):
    M = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def _masks_restated(seed, K, n, n_cases, strata):
    """CPU restatement of k_generate_masks: selection sampling per stratum on splitmix64(seed, r, patient)."""
    M = (1 << 64) - 1
    strata = np.zeros(n, np.int64) if strata is None else np.asarray(strata)
    S = int(strata.max()) + 1
    out = []
    for r in range(K):
        need = [int(((strata == s) & (np.arange(n) < n_cases)).sum()) for s in range(S)]
        rem = [int((strata == s).sum()) for s in range(S)]
        base = _mix64(seed ^ ((0x51ED270B7F3C9A1D * (r + 1)) & M))
        bits = 0
        for c in range(n):
            s = int(strata[c])
            u = _mix64((base + c) & M)
            if ((u * rem[s]) >> 64) < need[s]:
                bits |= 1 << c
                need[s] -= 1
            rem[s] -= 1
        out.append(bits)
    return out


@pytest.mark.parametrize("stratified", [False, True])
def test_device_permutation_masks(stratified):
    """gcre_generate_perm_masks: every mask is reproduced bit for bit by the CPU restatement, keeps exactly nCases
    cases (inside every stratum when stratified, R/Utils.R:8-13), and feeds the scorer like uploaded masks do."""
    nc, nt, K = 41, 59, 37
    n = nc + nt
    strata = (np.arange(n) * 7 % 3).astype(np.int32) if stratified else None
    p = make_problem(25, 60, nc, nt, K, 3, method="method1", top_k=5, seed=91, table=small_table(nc, nt))
    ex = api.JoinExec("method1", nc, nt, K)
    ex.generate_permutations(1234567, strata)
    want = _masks_restated(1234567, K, n, nc, strata)
    got = []
    for r in range(K):
        words = ex.perm_mask(r)
        got.append(sum(int(w) << (64 * k) for k, w in enumerate(words)))
    assert got == want
    for bits in got:
        assert bin(bits).count("1") == nc
        if stratified:
            for s in range(3):
                sel = [c for c in range(n) if strata[c] == s]
                assert sum((bits >> c) & 1 for c in sel) == sum(1 for c in sel if c < nc)
    assert len(set(got)) > K // 2                                   # they are not all the same permutation
    # scoring with generated masks == scoring with the same masks uploaded packed
    packed = np.array([[(b >> (64 * k)) & (2**64 - 1) for k in range((n + 63) // 64)] for b in got], dtype=np.uint64)
    ex.top_k = 5
    ex.set_value_table(p.value_table)
    ex2 = api.JoinExec("method1", nc, nt, K)
    ex2.top_k = 5
    ex2.set_value_table(p.value_table)
    ex2.set_permuted_masks(packed)
    u = p.levels.uids["2"]
    d1, d2 = ex.load(p.data1), ex2.load(p.data1)
    a = ex.join(u, d1, d1.select(p.levels.data_inds["2"]))
    b = ex2.join(u, d2, d2.select(p.levels.data_inds["2"]))
    np.testing.assert_array_equal(a.null.view(np.uint32), b.null.view(np.uint32))


@pytest.mark.parametrize("stratified", [False, True])
def test_device_permutation_masks_are_uniform(stratified):
    """A biased sampler would pass the reproduction test above: p-values are only as good as these permutations.  Over
    K = 20,000 device-drawn masks every patient must be a case with the frequency of its stratum (chi-square over the
    patients), and pairs of patients must be cases together as often as sampling without replacement says
    (R/Utils.R:22-46: a uniform permutation of the labels inside each stratum)."""
    nc, nt, K = 70, 110, 20000
    n = nc + nt
    strata = (np.arange(n) * 5 % 4).astype(np.int32) if stratified else np.zeros(n, np.int32)
    ex = api.JoinExec("method1", nc, nt, K)
    ex.generate_permutations(20261004, strata if stratified else None)
    W = (n + 63) // 64
    words = np.stack([ex.perm_mask(r) for r in range(K)]).astype(np.uint64)                    # K x W
    bits = np.unpackbits(words.view(np.uint8).reshape(K, W * 8), axis=1, bitorder="little")[:, :n].astype(np.int64)
    assert (bits.sum(axis=1) == nc).all()
    is_case = np.arange(n) < nc
    chi, dof = 0.0, 0
    for s_ in np.unique(strata):
        sel = np.flatnonzero(strata == s_)
        m, k = len(sel), int(is_case[sel].sum())
        assert (bits[:, sel].sum(axis=1) == k).all()                # the stratum keeps its number of cases
        if k == 0 or k == m:
            continue
        pr = k / m
        obs = bits[:, sel].sum(axis=0)
        # indicators inside one permutation are negatively correlated (their sum is fixed): the statistic is chi-square with
        # m - 1 degrees of freedom after the finite-population factor (m - 1) / m
        chi += float((((obs - K * pr) ** 2) / (K * pr * (1 - pr))).sum()) * (m - 1) / m
        dof += m - 1
        # pairs: both cases with probability k (k - 1) / (m (m - 1))
        both = (bits[:, sel[0]] & bits[:, sel[1]]).sum() + (bits[:, sel[2]] & bits[:, sel[-1]]).sum()
        p2 = k * (k - 1) / (m * (m - 1))
        assert abs(both - 2 * K * p2) < 6 * np.sqrt(2 * K * p2 * (1 - p2)), (s_, both, 2 * K * p2)
    assert abs(chi - dof) < 5 * np.sqrt(2 * dof), (chi, dof)
    # consecutive permutations are not correlated either: the number of patients that are cases in both r and r + 1
    same = (bits[:-1] & bits[1:]).sum(axis=1)
    if not stratified:
        mean = nc * nc / n
        assert abs(same.mean() - mean) < 6 * np.sqrt(mean) / np.sqrt(K - 1) * 2


def _gwaspa_case(seed, n_genes=60, n_edges=200, nc=48, nt=52):
    """A dataset + knowledge base with extra genes on either side and one planted 3-gene pathway whose union of
    carriers is (almost) all cases."""
    rng = np.random.default_rng(seed)
    g, src, trg, sign = synth.signed_network(n_genes, n_edges, rng)
    uid = np.arange(g) * 5 + 100
    symbols = [f"G{u}" for u in uid]
    n = nc + nt
    data = (rng.random((g, n)) < 0.02).astype(np.int32)
    # plant: walk a -> b -> c; each gene carries a disjoint third of the first 15 cases (each stays under the threshold)
    a = int(src[0]); b = int(trg[0])
    nxt = np.flatnonzero(src == b)
    c = int(trg[nxt[0]])
    for k, gene in enumerate((a, b, c)):
        data[gene] = 0
        data[gene, 5 * k:5 * k + 5] = 1
    genes = symbols + ["ORPHAN"]                                    # a gene the knowledge base does not know
    data = np.vstack([data, np.zeros((1, n), np.int32)])
    ents_uid = np.concatenate([uid, [9999]])                        # an entity without data
    ents_sym = symbols + ["NODATA"]
    rs = np.concatenate([uid[src], [uid[0]]])
    rt = np.concatenate([uid[trg], [9999]])                         # a relation that points outside the dataset
    rg = np.concatenate([sign, [1]])
    return genes, data, (ents_uid, ents_sym, rs, rt, rg), (f"G{uid[a]}", f"G{uid[b]}", f"G{uid[c]}")


@pytest.mark.parametrize("signed", [False, True])
def test_gwaspa_front_end_matches_oracle_and_finds_planted_pathway(signed):
    """report.gwaspa: native table + native level tables + device permutations + device scoring + getPaths decoding.
    Every level's lists equal the oracle's on the same masks (read back from the device); the planted pathway
    is the best length-3 row with p == 0."""
    from geneticscre_amd import report
    nc, nt, K = 48, 52, 64
    genes, data, network, planted = _gwaspa_case(5)
    out = report.gwaspa(genes, data, nc, nt, network, signed=signed, threshold=0.2, top_k=8, path_length=4,
                        n_permutations=K, seed=77)
    df = out["GWASPA.Results"]
    assert list(df.columns) == report.COLUMNS and len(df) == 8 * 4
    assert (np.diff(df["Pvalues"].to_numpy()) >= 0).all()
    best3 = df[df["Lengths"] == 3].iloc[0]
    assert best3["Paths"] == " -> ".join(planted) and best3["Pvalues"] == 0.0
    if not signed:                       # method2 splits carriers by the sign of the gene they sit on
        assert best3["Cases"] == 15 and best3["Controls"] <= 3

    # same masks through the oracle: label kept <=> (patient is a case) == (mask bit)
    prep = out["prepared"]
    ex = api.JoinExec("method2" if signed else "method1", nc, nt, K)
    ex.generate_permutations(77)
    bits = np.array([[(int(ex.perm_mask(r)[c // 64]) >> (c % 64)) & 1 for c in range(nc + nt)] for r in range(K)])
    ex.close()
    kept = (bits == (np.arange(nc + nt) < nc)[None, :]).astype(np.int32)
    lv = api.build_levels(len(prep.ents_uid), prep.src, prep.trg, prep.sign)
    n2 = len(prep.ents2_uid)
    from geneticscre_amd.uids import UidRelSet
    ids2 = np.arange(n2, dtype=np.int32)
    lv.uids["1b"] = UidRelSet(1, ids2, ids2, np.ones(n2, np.int32), np.arange(n2, dtype=np.int64), np.ones(n2, np.int32))
    lv.data_inds["1b"], lv.n_paths["1b"] = ids2, n2
    p = synth.Problem("method2" if signed else "method1", nc, nt, 4, 8, K, lv, prep.data1, prep.data2,
                      api.values_table(nc, nt), kept, 0)
    want = oracle.process_paths(p, order="canonical")
    for L in range(1, 5):
        g, w = out["levels"][f"lst{L}"], want[f"lst{L}"]
        np.testing.assert_array_equal(g.null.view(np.uint32), w.null.view(np.uint32), err_msg=f"null L{L}")
        np.testing.assert_array_equal(g.scores.view(np.uint64), w.scores.view(np.uint64))
        np.testing.assert_array_equal(g.cases, w.cases)
        np.testing.assert_array_equal(g.ctrls, w.ctrls)


@pytest.mark.parametrize("method", ["method1", "method2"])
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("window_tiles", ["", "1"])
def test_process_paths_on_several_devices_from_one_process(method, devices, window_tiles, monkeypatch):
    """gcre_process_paths_devices (what the .Call shim calls): one context and host thread per listed device, joined paths
    sharded, maxima and top-k tables merged on the host -- rehearsed here with the one GPU of the box listed up to three
    times.  Bit-identical to the oracle (and hence to the single-context call) for every device list.  The device threads
    also MAX-merge their running maxima during the larger joins (ExchangeHub; GCRE_EXCHANGE_UNIT makes these joins large)."""
    monkeypatch.setenv("GCRE_EXCHANGE_UNIT", "50")
    if window_tiles:   # two permutation windows (2048 + 252): every device replays its inspections in the second one
        monkeypatch.setenv("GCRE_WINDOW_TILES", window_tiles)
    p = make_problem(70, 260, 33, 41, 2300, 5, method=method, top_k=11, seed=21, table=small_table(33, 41, 5))
    want = oracle.process_paths(p, order="canonical", nthreads=4)
    got = api.process_paths_devices(p, devices)
    for lvl in range(1, 6):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


def test_rccl_selftest_through_the_librarys_own_binding():
    """libgcre_hip.so dlopens librccl.so itself (no torch in between): a one-rank communicator on device 0 and one
    ncclAllReduce(ncclFloat32, ncclMax), checked -- what a one-GPU box can run of the multi-GPU merge."""
    before = api.rccl_collectives()
    api.rccl_selftest(0)
    assert api.rccl_collectives() == before + 1


@pytest.mark.parametrize("method", ["method1", "method2"])
def test_process_paths_devices_merges_the_maxima_over_rccl(method, monkeypatch):
    """GCRE_RCCL=force: gcre_process_paths_devices creates a communicator even for ONE device and merges every level's null
    maxima with ncclAllReduce(MAX) on the device (two permutation windows: one collective per level and window).  With
    several distinct devices the same code runs with N ranks; on this box it is the one-rank case.  Results: the oracle's."""
    monkeypatch.setenv("GCRE_RCCL", "force")
    monkeypatch.setenv("GCRE_WINDOW_TILES", "1")
    p = make_problem(60, 200, 33, 41, 2300, 4, method=method, top_k=11, seed=23, table=small_table(33, 41, 6))
    want = oracle.process_paths(p, order="canonical", nthreads=4)
    before = api.rccl_collectives()
    got = api.process_paths_devices(p, [0])
    assert api.rccl_collectives() - before == 4 * 2      # lst1..lst4 (the 1a join's result is discarded, its merge is not needed) x 2 windows
    for lvl in range(1, 5):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


def test_native_harness_reproduces_the_reference_harness_output():
    """tools/harness/gcre_harness (the counterpart of the reference's test/harness.cpp) on SURVEY.md Appendix B's dump
    with the flags of the four recorded runs of the unmodified reference binary: the printed level-4 scores and null
    maxima are the reference's, ids wherever the score is not tied (App. A-9)."""
    import json
    import os
    import re
    import subprocess
    from geneticscre_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = hip_build.build_harness()
    gold = json.load(open(os.path.join(root, "tests", "golden", "appendix_b_expected.json")))
    for case in gold["cases"]:
        out = subprocess.run([exe, "-f", os.path.join(root, "tests", "golden", "appendix_b_tiny.txt"), "-p", str(case["iterations"]),
                              "-m", case["method"], "-l", str(case["path_length"]), "-k", str(case["top_k"]), "--lib", api.lib_path()],
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        res = re.search(r"results : (\d+) \|(.*)", out.stdout)
        got = [(float(m.group(1)), int(m.group(2)), int(m.group(3))) for m in re.finditer(r"(-?inf|-?[0-9.]+)\[(-?\d+):(-?\d+)\]", res.group(2))]
        want_scores = [float(s) for s in case["scores"]]
        assert [g[0] for g in got] == want_scores and int(res.group(1)) == len(want_scores)
        perms = [float(v) for v in re.search(r"perms :(.*)", out.stdout).group(1).split()]
        assert perms == [float(v) for v in case["null"]]
        for (s, a, b), (ea, eb) in zip(got, case["ids"]):
            if want_scores.count(s) == 1:
                assert (a, b) == (ea, eb)
        assert out.stdout.rstrip().endswith("done")
