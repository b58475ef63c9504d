"""Thresholds shared between devices during a join (gcre_join_opts.exchange): a shard prunes against the whole level's
running per-permutation maxima (`perm_scores[r] = max(...)`, src/methods.h:101-102), handed in by a caller-side MAX
all-reduce.  Rehearsed on one GPU: pass A records what every rank would send at each exchange, pass B hands every rank the
MAX over all ranks' records.  Merged results must be the oracle's in every kernel form; the number of calls is fixed by
the caller, whatever road the join takes (pytest -m gpu)."""
import numpy as np
import pytest
import torch

import oracle
from geneticscre_amd import api, dist
from geneticscre_amd.synth import make_problem

pytestmark = pytest.mark.gpu

LEVELS = (("1b", "lst1"), ("2", "lst2"), ("3", "lst3"), ("4", "lst4"), ("5", "lst5"))


def run_ranks(p, world, K, windows=None):
    """-> per-rank results of pass B, number of exchange calls per (rank, level, window)."""
    torch.zeros(1, device="cuda")
    d_null = torch.zeros(max(K, 1), dtype=torch.float32, device="cuda")
    plans = [api.ResidentPlan(p) for _ in range(world)]
    try:
        if windows:
            for pl in plans:
                pl.set_window(windows)
        rec, calls = {}, {}
        for rank, pl in enumerate(plans):
            def record(name, k0, k1, rank=rank):
                rec.setdefault((name, k0), {}).setdefault(rank, []).append(d_null[k0:k1].clone())
            pl.run(rank, world, d_null_out=d_null.data_ptr(), exchange=record)
        parts = []
        for rank, pl in enumerate(plans):
            seen = {}
            def merged(name, k0, k1, rank=rank, seen=seen):
                e = seen.get((name, k0), 0)
                seen[(name, k0)] = e + 1
                calls[(rank, name, k0)] = e + 1
                m = d_null[k0:k1]
                for lst in rec[(name, k0)].values():
                    torch.maximum(m, lst[e], out=m)
                torch.cuda.synchronize()
            parts.append(pl.run(rank, world, d_null_out=d_null.data_ptr(), exchange=merged))
        counts = {name: plans[0].exchange_count(name, world) for name in plans[0].names}
        return parts, calls, counts
    finally:
        for pl in plans:
            pl.close()


def check_merged(parts, want, p, L):
    for name, lst in LEVELS[:L]:
        null = np.maximum.reduce([r[name].null for r in parts])
        rows = [np.stack([r[name].scores, r[name].src, r[name].trg, r[name].cases, r[name].ctrls], axis=1) for r in parts]
        best = dist.merge_topk(np.vstack(rows), p.top_k)
        w = want[lst]
        np.testing.assert_array_equal(null.view(np.uint32), w.null.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(best[:, 0], w.scores, err_msg=name)


@pytest.mark.parametrize("kernel", ["auto", "ie", "sparse", "dense"])
@pytest.mark.parametrize("method", ["method1", "method2"])
def test_shared_thresholds_change_no_result(method, kernel, monkeypatch):
    """Three ranks, every join above 40 path-tiles per rank exchanges: the merged maxima and top-k are the oracle's, and
    every rank made exactly the calls the plan announced -- also where the join never ran the pruned kernel."""
    if kernel != "auto":
        monkeypatch.setenv("GCRE_NULL_KERNEL", kernel)
    monkeypatch.setenv("GCRE_EXCHANGE_UNIT", "20")
    monkeypatch.setenv("GCRE_IE_WARM", "64")
    K = 2300
    p = make_problem(60, 200, 310, 335, K, 4, method=method, top_k=15, seed=31, threshold=0.05)
    want = oracle.process_paths(p, order="canonical")
    parts, calls, counts = run_ranks(p, 3, K)
    check_merged(parts, want, p, 4)
    assert max(counts.values()) >= 3
    for rank in range(3):
        for name, n in counts.items():
            assert calls.get((rank, name, 0), 0) == n, (rank, name, calls, counts)


def test_shared_thresholds_with_windows_and_quads(monkeypatch):
    """Two-tile windows, the quad form forced on, two ranks: exchanges per (level, window)."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_IE_QUAD", "2")
    monkeypatch.setenv("GCRE_IE_WARM", "0")
    monkeypatch.setenv("GCRE_EXCHANGE_UNIT", "40")
    K = 5000
    p = make_problem(70, 260, 310, 335, K, 4, method="method1", top_k=15, seed=32, threshold=0.05)
    want = oracle.process_paths(p, order="canonical")
    parts, calls, counts = run_ranks(p, 2, K, windows=4096)
    check_merged(parts, want, p, 4)
    assert counts["4"] >= 4
    for rank in range(2):
        for k0 in (0, 4096):
            assert calls[(rank, "4", k0)] == counts["4"]


@pytest.mark.parametrize("tail", ["-1", "3", "40"])
def test_exchange_schedules_change_no_result(tail, monkeypatch):
    """GCRE_EXCHANGE_TAIL / GCRE_EXCHANGE_MAX only move the points at which a join hands its maxima over (the last slices
    equal steps instead of doubling ones, up to 12 of them): same calls as announced, same merged results."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_EXCHANGE_UNIT", "2")
    monkeypatch.setenv("GCRE_EXCHANGE_MAX", "12")
    monkeypatch.setenv("GCRE_EXCHANGE_TAIL", tail)
    monkeypatch.setenv("GCRE_IE_WARM", "64")
    K = 2300
    p = make_problem(60, 200, 310, 335, K, 4, method="method1", top_k=15, seed=35, threshold=0.05)
    want = oracle.process_paths(p, order="canonical")
    parts, calls, counts = run_ranks(p, 2, K)
    check_merged(parts, want, p, 4)
    assert max(counts.values()) >= 9
    for rank in range(2):
        for name, n in counts.items():
            assert calls.get((rank, name, 0), 0) == n, (rank, name, calls, counts)


def test_exchange_is_ignored_without_a_device_buffer():
    """No d_null_out: nothing to hand over -- the join runs as if no exchange had been asked for."""
    p = make_problem(40, 110, 100, 120, 300, 3, method="method1", top_k=10, seed=33)
    want = oracle.process_paths(p, order="canonical")
    plan = api.ResidentPlan(p)
    try:
        hit = []
        out = plan.run(0, 2, exchange=lambda *a: hit.append(a))
        assert not hit
        assert (out["3"].null <= want["lst3"].null).all()
    finally:
        plan.close()
