"""The quad form of the pruned method-1 null kernel (gcre_ieq.hip: up to four segments per wave that share the added
rows' planes) against the CPU oracle and against k_null_ie_m1, through the C ABI (pytest -m gpu).  Reference
semantics: src/methods.h:58-105."""
import numpy as np
import pytest

import oracle
from geneticscre_amd import api
from geneticscre_amd.synth import make_problem
from helpers import assert_same_result, small_table

pytestmark = pytest.mark.gpu


def run_plan(p, masks=None):
    plan = api.ResidentPlan(p, device=0, packed_masks=masks)
    try:
        out = plan.run()
        prof = dict(plan.last_profile)
    finally:
        plan.close()
    return out, prof


@pytest.mark.parametrize("warm", ["0", "64"])
@pytest.mark.parametrize("length,n_perm,patients", [(4, 2500, (450, 550)), (5, 700, (90, 110)), (3, 4100, (1300, 1200))])
def test_quad_kernel_matches_oracle_and_runs(length, n_perm, patients, warm, monkeypatch):
    """Every level bit-exact against the oracle with the quad form forced on, and the profile says it really ran.
    warm = 0: thresholds start from zero inside the pruned kernel; 64: seeded by a warm-up slice."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_IE_QUAD", "2")
    monkeypatch.setenv("GCRE_IE_WARM", warm)
    nc, nt = patients
    p = make_problem(120, 700, nc, nt, n_perm, length, method="method1", top_k=25, seed=length * 100 + n_perm)
    want = oracle.process_paths(p, order="canonical", nthreads=8)
    got, prof = run_plan(p)
    names = {"1b": "lst1", "2": "lst2", "3": "lst3", "4": "lst4", "5": "lst5"}
    for name, lst in names.items():
        if name in got:
            assert_same_result(got[name], want[lst])
    if warm == "0" or length >= 4:   # a short last level may fit the warm-up slice whole
        assert prof["ie_quad_launches"] > 0, prof


@pytest.mark.parametrize("table_seed", [1, 2])
def test_quad_kernel_arbitrary_table_prunes_exactly(table_seed, monkeypatch):
    """A table that is not valley-shaped (random cells): the ladder's intervals are valid for any table, so pruned and
    unpruned launches of the quad kernel give the oracle's maxima."""
    nc, nt = 300, 340
    p = make_problem(100, 600, nc, nt, 2300, 4, method="method1", top_k=10, seed=5, table=small_table(nc, nt, table_seed))
    want = oracle.process_paths(p, order="canonical", nthreads=8)
    for prune in ("1", "0"):
        monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
        monkeypatch.setenv("GCRE_IE_QUAD", "2")
        monkeypatch.setenv("GCRE_IE_WARM", "0")
        monkeypatch.setenv("GCRE_IE_PRUNE", prune)
        got, prof = run_plan(p)
        for name, lst in (("2", "lst2"), ("3", "lst3"), ("4", "lst4")):
            assert_same_result(got[name], want[lst])
        assert prof["ie_quad_launches"] > 0


@pytest.mark.parametrize("top_rate", [0.08, 0.25])
def test_quad_and_single_segment_forms_agree_on_dense_rows(top_rate, monkeypatch):
    """Carrier rates up to 8 % / 25 %: overlap lists of several 8-entry blocks, delta lists, 12-16 counter planes, bound
    filters that leave most lanes uncertain.  Same results as the oracle with the quad form and without it."""
    rng = np.random.default_rng(3)
    nc, nt = 260, 250
    p = make_problem(70, 420, nc, nt, 2100, 4, method="method1", top_k=15, seed=9)
    dense = (rng.random(p.data1.shape) < rng.uniform(0.02, top_rate, size=(p.data1.shape[0], 1))).astype(np.int32)
    p.data1[:] = dense
    p.data2[:] = dense[p.levels.uids["1b"].src]
    want = oracle.process_paths(p, order="canonical", nthreads=8)
    for quad in ("2", "0"):
        monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
        monkeypatch.setenv("GCRE_IE_QUAD", quad)
        monkeypatch.setenv("GCRE_IE_WARM", "0")
        got, prof = run_plan(p)
        for name, lst in (("1b", "lst1"), ("2", "lst2"), ("3", "lst3"), ("4", "lst4")):
            assert_same_result(got[name], want[lst])
        assert (prof["ie_quad_launches"] > 0) == (quad == "2"), prof


def test_eleven_counter_planes(monkeypatch):
    """Carrier totals between 1024 and 2047 need 11 counter bits: the quad kernel has variants of its own for them
    (configs[3] runs them at full size).  1,500 patients, genes carried by 30-42 % of them: oracle results with the quad
    form forced on, and the level-3 rows really carry that many."""
    monkeypatch.setenv("GCRE_NULL_KERNEL", "ie")
    monkeypatch.setenv("GCRE_IE_QUAD", "2")
    monkeypatch.setenv("GCRE_IE_WARM", "64")
    rng = np.random.default_rng(11)
    nc, nt = 760, 740
    p = make_problem(60, 330, nc, nt, 2100, 4, method="method1", top_k=12, seed=17)
    dense = (rng.random(p.data1.shape) < rng.uniform(0.30, 0.42, size=(p.data1.shape[0], 1))).astype(np.int32)
    p.data1[:] = dense
    p.data2[:] = dense[p.levels.uids["1b"].src]
    want = oracle.process_paths(p, order="canonical", nthreads=8)
    most = int(np.unpackbits(want["paths3"].view(np.uint8), axis=1).sum(axis=1).max())
    assert 1024 <= most < 2048, most
    got, prof = run_plan(p)
    for name, lst in (("2", "lst2"), ("3", "lst3"), ("4", "lst4")):
        assert_same_result(got[name], want[lst])
    assert prof["ie_quad_launches"] > 0
