"""BASELINE.json configs[1], [3] and [4] under -m gpu (configs[0] and [2] are covered by test_gpu_parity.py and the bench).

configs[1] runs at its stated size against the oracle.  configs[3] and [4] are too large for the CPU oracle at full size
(the reference itself cannot run configs[4]), so each is tested twice: its GEOMETRY (mask width,
permutation windows, method, path length) on an oracle-sized network, bit-exact against the oracle in every kernel form;
and its FULL SIZE through a property the domain offers -- two independent algorithms (inclusion-exclusion on count
planes vs delta streaming of bit lists, gcre_ie*.hip vs gcre_sparse.hip) must produce identical top-k tables and null
maxima for every level.  Reference semantics: src/methods.h:58-232, limits src/gcre_paths.h:26,38,46."""
import hashlib
import os
import sys

import numpy as np
import pytest

import oracle
from geneticscre_amd import api
from geneticscre_amd.synth import make_problem
from helpers import assert_same_result

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pytestmark = pytest.mark.gpu
LST = {"1b": "lst1", "2": "lst2", "3": "lst3", "4": "lst4", "5": "lst5"}
MODES = ["auto", "ie-quad", "ie-m1", "ie-noprune", "sparse", "dense"]


def set_mode(monkeypatch, kernel):
    if kernel == "ie-noprune":
        monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
        monkeypatch.setenv("GCRE_IE_PRUNE", "0")
    elif kernel in ("ie-quad", "ie-m1"):
        monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
        monkeypatch.setenv("GCRE_IE_QUAD", "2" if kernel == "ie-quad" else "0")
        monkeypatch.setenv("GCRE_IE_WARM", "0")
    else:
        monkeypatch.setenv("GCRE_NULL_KERNEL", kernel)


def run_plan(p, masks=None, seed=None):
    plan = api.ResidentPlan(p, device=0, packed_masks=masks, mask_seed=seed)
    try:
        out = plan.run()
        prof = dict(plan.last_profile)
        windows = max(1, -(-p.iterations // plan._window)) if plan._window else 1
    finally:
        plan.close()
    return out, prof, windows


def digest(out):
    h = hashlib.sha256()
    for name in sorted(out):
        r = out[name]
        for arr in (r.scores, r.src, r.trg, r.cases, r.ctrls, r.null):
            h.update(np.ascontiguousarray(arr).tobytes())
    return h.hexdigest()


# ---------------------------------------------------------------------------------------------------------------------
# configs[1]: STRINGdb 5k-gene subgraph, 1k patients (16 mask words), 1k permutations, length 3, method 1 -- as stated
# ---------------------------------------------------------------------------------------------------------------------
def test_config1_subgraph_at_full_size_matches_oracle():
    cfg = bench.CONFIGS["subgraph"]
    prob, masks = bench.build_inputs(cfg, 20261003, 100)
    assert (prob.n_cases + prob.n_ctrls + 63) // 64 == 16 and prob.iterations == 1000 and prob.path_length == 3
    want = oracle.process_paths(prob, order="canonical", nthreads=8, packed_masks=masks)
    got, prof, _ = run_plan(prob, masks)
    assert sum(int(np.maximum(np.asarray(prob.levels.uids[k].count), 0).sum()) for k in ("3",)) > 500_000
    for name, lst in LST.items():
        if name in got:
            assert_same_result(got[name], want[lst])
    assert prof["ie_launches"] > 0


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: 5,000 + 5,000 patients (157 mask words), 100,000 permutations in windows, length 4, method 1
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config3_small():
    nc = nt = 5000
    p = make_problem(150, 650, nc, nt, 4500, 4, method="method1", top_k=40, seed=33, table=bench.fast_table(nc, nt))
    return p, oracle.process_paths(p, order="canonical", nthreads=8)


@pytest.mark.parametrize("kernel", MODES)
def test_config3_geometry_matches_oracle(config3_small, kernel, monkeypatch):
    """157 words, K = 4,500 (not a multiple of the 2048-permutation tile), two permutation windows (2 tiles + 1)."""
    p, want = config3_small
    assert (p.n_cases + p.n_ctrls + 63) // 64 == 157
    set_mode(monkeypatch, kernel)
    monkeypatch.setenv("GCRE_WINDOW_TILES", "2")
    got, prof, windows = run_plan(p)
    assert windows == 2 or kernel == "dense"      # the dense form keeps no count planes: nothing to window
    for name, lst in LST.items():
        if name in got:
            assert_same_result(got[name], want[lst])


@pytest.fixture(scope="module")
def config3_small_hyper():
    nc = nt = 5000
    p = make_problem(150, 650, nc, nt, 4500, 4, method="method1", top_k=40, seed=33, table=api.values_table(nc, nt))
    return p, oracle.process_paths(p, order="canonical", nthreads=8)


@pytest.mark.parametrize("kernel", ["auto", "ie-quad", "sparse"])
def test_config3_geometry_with_the_hypergeometric_table(config3_small_hyper, kernel, monkeypatch):
    """The same geometry on the real -log hypergeometric table (gcre_values_table): the pruning ladder's intervals at 157
    words are checked on the table the product uses, not only on the smooth chi-square stand-in."""
    p, want = config3_small_hyper
    set_mode(monkeypatch, kernel)
    monkeypatch.setenv("GCRE_WINDOW_TILES", "2")
    got, prof, windows = run_plan(p)
    for name, lst in LST.items():
        if name in got:
            assert_same_result(got[name], want[lst])


def test_config3_full_size_two_algorithms_agree(monkeypatch):
    """BASELINE configs[3] as bench.py builds it (17,000 genes / 200,000 relations, 10,000 patients, 100,000
    permutations drawn on the device, length 4: 2.85e12 scores per pass): inclusion-exclusion (default) and delta
    streaming produce the same SHA-256 over every level's top-k table and null maxima.  The value table is the cheap
    chi-square stand-in (building the hypergeometric one takes a minute and does not matter to this property)."""
    cfg = bench.CONFIGS["sharded"]
    prob, masks = bench.build_inputs(cfg, 20261003, 100, table_fn=bench.fast_table)
    assert masks is None and prob.iterations == 100_000
    res = {}
    for kernel in ("auto", "sparse"):
        monkeypatch.setenv("GCRE_NULL_KERNEL", kernel)
        out, prof, windows = run_plan(prob, seed=20261003)
        res[kernel] = digest(out)
        if kernel == "auto":
            assert prof["ie_launches"] > 0 and prof["scores"] == prob.iterations * sum(
                int(np.maximum(np.asarray(prob.levels.uids[k].count), 0).sum()) for k in ("1a", "1b", "2", "3", "4"))
    assert res["auto"] == res["sparse"]


# ---------------------------------------------------------------------------------------------------------------------
# configs[4]: method 2 (signed), length 5, 50,000 patients (782 mask words), 100,000 permutations
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config4_small():
    nc, nt = 2000, 48000       # 50,000 patients = 782 words; the unbalanced split keeps the value table at 96 M cells
    p = make_problem(45, 150, nc, nt, 2300, 5, method="method2", top_k=30, seed=44, table=bench.fast_table(nc, nt))
    return p, oracle.process_paths(p, order="canonical", nthreads=8)


@pytest.mark.parametrize("kernel", ["auto", "ie", "ie-noprune", "sparse", "dense"])
def test_config4_geometry_matches_oracle(config4_small, kernel, monkeypatch):
    """782 words per half, signed method, level 5 (paths3 x paths3), K = 2,300 = two tiles."""
    p, want = config4_small
    assert (p.n_cases + p.n_ctrls + 63) // 64 == 782
    set_mode(monkeypatch, kernel)
    got, prof, _ = run_plan(p)
    for name, lst in LST.items():
        assert_same_result(got[name], want[lst])


@pytest.fixture(scope="module")
def config4_small_hyper():
    nc, nt = 2000, 48000
    p = make_problem(45, 150, nc, nt, 2300, 5, method="method2", top_k=30, seed=44, table=api.values_table(nc, nt))
    return p, oracle.process_paths(p, order="canonical", nthreads=8)


@pytest.mark.parametrize("kernel", ["auto", "sparse"])
def test_config4_geometry_with_the_hypergeometric_table(config4_small_hyper, kernel, monkeypatch):
    """782 words per half, signed method, on the real hypergeometric table (2,000 + 48,000 patients)."""
    p, want = config4_small_hyper
    set_mode(monkeypatch, kernel)
    got, prof, _ = run_plan(p)
    for name, lst in LST.items():
        assert_same_result(got[name], want[lst])


def test_config4_full_size_two_algorithms_agree(monkeypatch):
    """BASELINE configs[4] at its full cohort and permutation count on ONE GPU: 25,000 + 25,000 patients, 100,000
    permutations, length 5, method 2, on the 17,000-gene network with 60,000 relations (2.0 M level-5 paths, 2.95e11
    scores per pass).  With configs[2]'s 200,000 relations level 5 has ~3e8 joined paths = 3e13 scores per pass: about
    half a minute per pass for the inclusion-exclusion form on one GPU and several minutes for the delta-streaming
    cross-check -- BASELINE names eight GPUs for it -- so the single-GPU check runs the smaller network."""
    cfg = bench.CONFIGS["signed"]
    prob, masks = bench.build_inputs(cfg, 20261003, 100, table_fn=bench.fast_table)
    assert masks is None and (prob.n_cases + prob.n_ctrls + 63) // 64 == 782
    res = {}
    for kernel in ("auto", "sparse"):
        monkeypatch.setenv("GCRE_NULL_KERNEL", kernel)
        out, prof, windows = run_plan(prob, seed=20261003)
        res[kernel] = digest(out)
    assert res["auto"] == res["sparse"]
