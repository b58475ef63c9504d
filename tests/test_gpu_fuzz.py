"""Seeded random problems through the C ABI against the CPU oracle, bit-exact: sizes, carrier thresholds, permutation
counts, methods, path lengths, kernel forms and tuning knobs drawn per case (pytest -m gpu).  A handful of cases by
default; GCRE_FUZZ_CASES=5000 [GCRE_FUZZ_BASE=1000000] for a long run after a kernel change (the draw only depends on the
case number)."""
import os

import numpy as np
import pytest

import oracle
from geneticscre_amd import api
from geneticscre_amd.synth import make_problem
from helpers import assert_same_result
from test_gpu_exchange import check_merged, run_ranks

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("GCRE_FUZZ_CASES", "8"))
LIGHT = os.environ.get("GCRE_FUZZ_LIGHT") == "1"       # bulk runs: no cohorts past 6,000 patients (their oracle runs take seconds)
BASE = int(os.environ.get("GCRE_FUZZ_BASE", "0"))      # another stretch of the case numbers
KNOBS = {
    "GCRE_NULL_KERNEL": ["", "ie", "ie", "ie", "sparse", "dense"],
    "GCRE_IE_QUAD": ["", "0", "2", "2"],
    "GCRE_IE_WARM": ["", "0", "64", "256"],
    "GCRE_IE_PRUNE": ["", "", "", "0"],
    "GCRE_IEQ_BATCH": ["", "1", "3"],
    "GCRE_IE_BATCH": ["", "1", "5"],
    "GCRE_WINDOW_TILES": ["", "", "1", "2"],
    "GCRE_PLANES_OUT_MAX_MB": ["", "", "0"],
    "GCRE_CHUNK_PATHS": ["", "", "700", "5000"],        # several chunks per join: shard and recipe boundaries inside a join
    "GCRE_SELECT_STREAM": ["", "", "0"],
    "GCRE_PREFETCH_TABLES": ["", "0"],
    "GCRE_SPARSE_WAVES_PER_CU": ["", "", "4", "16"],
    "GCRE_IE_SJT": ["", "1", "64"],
    "GCRE_AHEAD": ["", "0", "1"],                       # inspect- and launch-ahead (ResidentPlan passes only; default: plans up to 64 M paths)
}


def entered(test: str, case: int) -> None:
    """`-x -q` shows dots only: the case in flight goes to stderr and to a progress file that survives a run cut off by the
    box's time limit (GCRE_FUZZ_LOG, default gpurun_out/fuzz_progress.log when that directory exists)."""
    line = f"[fuzz] {test} case {case} (GCRE_FUZZ_BASE={BASE})"
    import sys
    print(line, file=sys.stderr, flush=True)
    path = os.environ.get("GCRE_FUZZ_LOG")
    if path is None and os.path.isdir("gpurun_out"):
        path = os.path.join("gpurun_out", "fuzz_progress.log")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def value_table(kind: str, n_cases: int, n_ctrls: int, seed: int):
    """None = the hypergeometric table; otherwise an arbitrary one (the pruning must be exact for any table)."""
    if kind == "hyper":
        return None
    rng = np.random.default_rng(seed)
    t = rng.random((n_cases + 1, n_ctrls + 1)) * 12.0
    if kind == "ties":
        t = np.round(t)            # thirteen distinct values: ties everywhere
    elif kind == "flat":
        t[:] = 3.25
        t[rng.integers(0, n_cases + 1, 40), rng.integers(0, n_ctrls + 1, 40)] = 9.5   # a few spikes
    return t


def draw(case: int):
    rng = np.random.default_rng(9000 + case)
    n_cases = int(rng.integers(20, 420))
    n_ctrls = int(rng.integers(20, 420))
    if case % 7 == 3:      # lopsided cohorts, a handful of controls (or cases)
        if rng.random() < 0.5:
            n_ctrls = int(rng.integers(2, 12))
        else:
            n_cases = int(rng.integers(2, 12))
    if case % 11 == 5:     # wider rows
        n_cases, n_ctrls = int(rng.integers(400, 1100)), int(rng.integers(400, 1100))
    wide = case % 13 == 6     # BASELINE configs[2] width: ~5,000 patients, 79-82 mask words
    if wide:
        n_cases, n_ctrls = int(rng.integers(2300, 2700)), int(rng.integers(2300, 2700))
    huge = case % 29 == 7 and not LIGHT     # 100-160 mask words: the inspector's 4- and 5-block forms (two waves per SIMD)
    if huge:
        n_cases, n_ctrls = int(rng.integers(3200, 5100)), int(rng.integers(3200, 5100))
    beyond = case % 31 == 9 and not LIGHT   # past 10,240 patients: the per-path inspector, wider planes
    if beyond:
        n_cases, n_ctrls = int(rng.integers(5200, 6000)), int(rng.integers(5100, 6000))
    genes = int(rng.integers(25, 90))
    edges = int(rng.integers(genes * 2, genes * 5))
    if huge or beyond:
        genes = int(rng.integers(25, 45))
        edges = int(rng.integers(genes * 2, genes * 4))
    length = int(rng.choice([3, 4, 4, 5]))
    if case % 5 == 4:      # a larger network: thousands of segments per join, quads, several chunks of work per wave
        genes = int(rng.integers(150, 320))
        edges = int(rng.integers(genes * 3, genes * 6))
        length = int(rng.choice([3, 4, 4]))
    if length == 5:
        edges = min(edges, genes * 3)
    perms = int(rng.choice([0, 1, 31, 100, 257, 2048, 2300, 4500]))
    if wide:
        perms = min(perms, 2300)
    if huge or beyond:
        perms = min(perms, 257)
    method = str(rng.choice(["method1", "method1", "method2"]))
    threshold = float(rng.choice([0.02, 0.05, 0.05, 0.15, 0.4, 0.9]))
    top_k = int(rng.choice([1, 7, 15, 40, 3000]))
    table = str(rng.choice(["hyper", "hyper", "random", "ties", "flat"]))
    env = {k: str(rng.choice(v)) for k, v in KNOBS.items()}
    return dict(n_cases=n_cases, n_ctrls=n_ctrls, genes=genes, edges=edges, length=length, perms=perms, method=method,
                threshold=threshold, top_k=top_k, seed=1000 + case, table=table), env


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_problem_matches_oracle(case, monkeypatch):
    entered("problem_matches_oracle", BASE + case)
    cfg, env = draw(BASE + case)
    for k, v in env.items():
        if v:
            monkeypatch.setenv(k, v)
    p = make_problem(cfg["genes"], cfg["edges"], cfg["n_cases"], cfg["n_ctrls"], cfg["perms"], cfg["length"],
                     method=cfg["method"], top_k=cfg["top_k"], seed=cfg["seed"], threshold=cfg["threshold"],
                     table=value_table(cfg["table"], cfg["n_cases"], cfg["n_ctrls"], cfg["seed"]))
    want = oracle.process_paths(p, order="canonical")
    got = api.process_paths(p)
    for lvl in range(1, cfg["length"] + 1):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])


@pytest.mark.parametrize("case", range(max(1, N_CASES // 4)))
def test_random_sharded_plan_matches_oracle(case, monkeypatch):
    """The same draw through ResidentPlan, sharded over 2..5 ranks with the thresholds exchanged inside the joins: the
    merged null maxima and top-k are the oracle's."""
    entered("sharded_plan", BASE + 100000 + case)
    cfg, env = draw(BASE + 100000 + case)
    env["GCRE_WINDOW_TILES"] = ""
    env["GCRE_EXCHANGE_UNIT"] = str([5, 50, 2000][case % 3])
    env["GCRE_PIVOT_SHARDS"] = "1" if case % 2 else ""
    for k, v in env.items():
        if v:
            monkeypatch.setenv(k, v)
    K = max(cfg["perms"], 1)
    p = make_problem(cfg["genes"], cfg["edges"], cfg["n_cases"], cfg["n_ctrls"], K, min(cfg["length"], 4),
                     method=cfg["method"], top_k=cfg["top_k"], seed=cfg["seed"], threshold=cfg["threshold"],
                     table=value_table(cfg["table"], cfg["n_cases"], cfg["n_ctrls"], cfg["seed"]))
    want = oracle.process_paths(p, order="canonical")
    parts, _, _ = run_ranks(p, 2 + case % 4, K)
    check_merged(parts, want, p, min(cfg["length"], 4))


def test_launch_ahead_keeps_the_recipe_lists_when_a_chunk_is_split_again(monkeypatch):
    """Fuzz case 6100095 of the sharded-plan test under GCRE_AHEAD=1, kept as a regression: rank 3 of 5 inspects its two-chunk
    level-3 join ahead, the launch pass splits the second chunk again at the shard's end, and the re-inspected pieces must
    not reserve their long lists over those of the first chunk (the recipe's overflow counter belongs to the join)."""
    cfg, env = draw(6100095)
    env.update({"GCRE_WINDOW_TILES": "", "GCRE_EXCHANGE_UNIT": "2000", "GCRE_PIVOT_SHARDS": "1", "GCRE_AHEAD": "1"})
    for k, v in env.items():
        if v:
            monkeypatch.setenv(k, v)
        else:
            monkeypatch.delenv(k, raising=False)
    K = max(cfg["perms"], 1)
    p = make_problem(cfg["genes"], cfg["edges"], cfg["n_cases"], cfg["n_ctrls"], K, min(cfg["length"], 4),
                     method=cfg["method"], top_k=cfg["top_k"], seed=cfg["seed"], threshold=cfg["threshold"],
                     table=value_table(cfg["table"], cfg["n_cases"], cfg["n_ctrls"], cfg["seed"]))
    want = oracle.process_paths(p, order="canonical")
    parts, _, _ = run_ranks(p, 5, K)
    check_merged(parts, want, p, min(cfg["length"], 4))


@pytest.mark.parametrize("case", range(max(1, N_CASES // 4)))
def test_random_resident_plan_with_kept_inspections(case, monkeypatch):
    """ResidentPlan over the same draw: a pass, a pass that keeps its inspections, a pass that replays them under another
    permutation window -- each the oracle's."""
    entered("kept_inspections", BASE + 200000 + case)
    cfg, env = draw(BASE + 200000 + case)
    env["GCRE_WINDOW_TILES"] = ""
    for k, v in env.items():
        if v:
            monkeypatch.setenv(k, v)
    K = max(cfg["perms"], 1)
    L = min(cfg["length"], 4)
    p = make_problem(cfg["genes"], cfg["edges"], cfg["n_cases"], cfg["n_ctrls"], K, L,
                     method=cfg["method"], top_k=cfg["top_k"], seed=cfg["seed"], threshold=cfg["threshold"],
                     table=value_table(cfg["table"], cfg["n_cases"], cfg["n_ctrls"], cfg["seed"]))
    want = oracle.process_paths(p, order="canonical")
    names = (("1b", 1), ("2", 2), ("3", 3), ("4", 4))[:L]
    plan = api.ResidentPlan(p)
    try:
        for keep, window in ((False, None), (True, None), (True, 2048), (False, 2048)):
            if window is not None:
                plan.set_window(window)
            got = plan.run(keep_inspections=keep)
            for name, lvl in names:
                assert_same_result(got[name], want[f"lst{lvl}"])
    finally:
        plan.close()


@pytest.mark.parametrize("case", range(max(1, N_CASES // 4)))
def test_random_problem_on_several_device_threads(case, monkeypatch):
    """gcre_process_paths_devices with device 0 listed 2..4 times (one context and host thread each, shards, the host-side
    threshold hub inside the joins): the oracle's results."""
    entered("several_device_threads", BASE + 300000 + case)
    cfg, env = draw(BASE + 300000 + case)
    env["GCRE_EXCHANGE_UNIT"] = str([5, 50, 2000][case % 3])
    for k, v in env.items():
        if v:
            monkeypatch.setenv(k, v)
    p = make_problem(cfg["genes"], cfg["edges"], cfg["n_cases"], cfg["n_ctrls"], cfg["perms"], cfg["length"],
                     method=cfg["method"], top_k=cfg["top_k"], seed=cfg["seed"], threshold=cfg["threshold"],
                     table=value_table(cfg["table"], cfg["n_cases"], cfg["n_ctrls"], cfg["seed"]))
    want = oracle.process_paths(p, order="canonical")
    got = api.process_paths_devices(p, devices=[0] * (2 + case % 3))
    for lvl in range(1, cfg["length"] + 1):
        assert_same_result(got[f"lst{lvl}"], want[f"lst{lvl}"])
