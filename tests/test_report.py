"""getPaths / p-value / result-table post-processing and the dataset + network filtering of GWASPA."""
import numpy as np
import pytest

from geneticscre_amd import api, report, synth
from geneticscre_amd.uids import build_level_tables

# uids 10..50; relations sorted by (src, trg)
ENTS = (np.array([10, 20, 30, 40, 50]), ["A", "B", "C", "D", "E"])
RELS = {"srcuid": np.array([10, 20, 20, 30, 40]), "trguid": np.array([20, 30, 40, 40, 50]),
        "sign": np.array([1, -1, 1, -1, 1])}


def _rels3():
    # getRels3: every 2-edge walk a->b->c grouped by first edge
    rows = []
    for s, t, g in zip(RELS["srcuid"], RELS["trguid"], RELS["sign"]):
        for s2, t2, g2 in zip(RELS["srcuid"], RELS["trguid"], RELS["sign"]):
            if s2 == t:
                rows.append((s, t, g, t2, g2))
    a = np.array(rows)
    return {"srcuid": a[:, 0], "trguid": a[:, 1], "sign": a[:, 2], "trguid2": a[:, 3], "sign2": a[:, 4]}


def test_get_paths_hand_worked_examples():
    """Each expectation is worked by hand from PathMethods.R:2-131 (ids are R's 1-based)."""
    rd = {"srcuid": ENTS[0]}
    r3 = _rels3()   # rows: 10-20-30, 10-20-40, 20-30-40, 20-40-50, 30-40-50
    assert r3["srcuid"].tolist() == [10, 10, 20, 20, 30]
    # length 1: gene of row ids[,2]; ids beyond the table are the (-) copy
    assert report.get_paths([[3, 3], [1, 7]], 1, rd, rd) == (["30", "20"], ["30 (+)", "20 (-)"])
    # length 2: relation ids[,2]:  20 -| 30  and  20 -> 40
    assert report.get_paths([[2, 2], [2, 3]], 2, rd, RELS) == (
        ["20 -> 30", "20 -> 40"], ["20 (+) -> 30 (-)", "20 (+) -> 40 (+)"])
    # length 3: relation ids[,1] then relation ids[,2]:  10 -> 20 -| 30,   20 -| 30 -| 40 (two inhibitions cancel)
    assert report.get_paths([[1, 2], [2, 4]], 3, RELS, RELS) == (
        ["10 -> 20 -> 30", "20 -> 30 -> 40"], ["10 (+) -> 20 (+) -> 30 (-)", "20 (+) -> 30 (-) -> 40 (+)"])
    # length 4: Rels3 row ids[,1] then relation ids[,2]:  10 -> 20 -| 30 -| 40
    assert report.get_paths([[1, 4]], 4, r3, RELS) == (
        ["10 -> 20 -> 30 -> 40"], ["10 (+) -> 20 (+) -> 30 (-) -> 40 (+)"])
    # length 5: Rels3 row ids[,1] then Rels3 row ids[,2]:  10 -> 20 -| 30 , 30 -| 40 -> 50
    assert report.get_paths([[1, 5]], 5, r3, r3) == (
        ["10 -> 20 -> 30 -> 40 -> 50"], ["10 (+) -> 20 (+) -> 30 (-) -> 40 (+) -> 50 (+)"])
    # the sentinel row (ids (0,0), App. A-8) indexes nothing
    assert report.get_paths([[0, 0]], 2, rd, RELS) == (["NA -> NA"], ["NA (+) -> NA (+)"])
    assert report.uid_to_symbol(*ENTS, ["10 -> 20 -> 30", "NA -> 99"]) == ["A -> B -> C", "NA -> NA"]
    assert report.uid_to_symbol(*ENTS, ["10 (+) -> 20 (-)"], signed=True) == ["A (+) -> B (-)"]


@pytest.mark.parametrize("seed", [3, 4])
def test_decoded_ids_are_walks_of_the_network(seed):
    """Property over EVERY joined path of every level: the (idx, loc) pair the scorer reports decodes to a walk whose
    consecutive genes are relations of the network and whose (+)/(-) tags multiply out the relation signs."""
    rng = np.random.default_rng(seed)
    g, src, trg, sign = synth.signed_network(40, 120, rng)
    lv = build_level_tables(g, src, trg, sign)
    uid = np.arange(g) * 3 + 7                                  # uid != rank, to catch rank/uid mix-ups
    prep = report.Prepared(uid, [f"G{u}" for u in uid], uid, [f"G{u}" for u in uid], src, trg, sign, None, None)
    fr = report.frames_of(prep, lv)
    edge = {(int(uid[s]), int(uid[t])): int(x) for s, t, x in zip(src, trg, sign)}
    per_level = {2: ("rels_data", "rels"), 3: ("rels", "rels"), 4: ("rels3", "rels"), 5: ("rels3", "rels3")}
    for L, (f1, f2) in per_level.items():
        u = lv.uids[str(L)]
        ids = [(i + 1, l + 1) for i in range(len(u.count)) for l in range(int(u.location[i]), int(u.location[i]) + max(int(u.count[i]), 0))]
        assert len(ids) == lv.n_paths[str(L)] and len(ids) > 0
        paths, signpaths = report.get_paths(ids, L, fr[f1], fr[f2])
        assert len(set(paths)) == len(paths)                     # distinct joins are distinct walks
        for p, sp in zip(paths[:: max(1, len(paths) // 400)], signpaths[:: max(1, len(paths) // 400)]):
            hops = [int(x) for x in p.split(" -> ")]
            tags = [h.split(" ")[1] for h in sp.split(" -> ")]
            assert len(hops) == L and tags[0] == "(+)"
            cur = 1
            for a, b, tag in zip(hops, hops[1:], tags[1:]):
                assert (a, b) in edge, (L, p)
                cur *= edge[(a, b)]
                assert tag == ("(+)" if cur == 1 else "(-)"), (L, p, sp)


def test_preprocess_table_rules():
    sym = ["g1", "g2", "g1", "NA", "g3", "g4"]
    d = np.array([[1, 0, 0, 0, 0, 0, 0, 0],
                  [2, 1, 0, 0, 0, 0, 0, 0],        # 2 -> 1
                  [1, 1, 1, 1, 1, 1, 1, 1],        # duplicated symbol: dropped
                  [1, 0, 0, 0, 0, 0, 0, 0],        # NA symbol: dropped
                  [1, 1, 1, 0, 0, 0, 0, 0],        # 3 carriers > 0.25 * (8 + 1) = 2.25: filtered
                  [0, 0, 0, 0, 0, 0, 0, 0]])       # no carriers: kept (freq <= target), as in Utils.R:187-189
    genes, data = report.preprocess_table(sym, d, 0.25, 4, 4)
    assert genes == ["g1", "g2", "g4"]
    assert data.tolist() == [[1, 0, 0, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0, 0], [0] * 8]
    with pytest.raises(ValueError, match="nCases \\+ nControls"):
        report.preprocess_table(sym, d, 0.25, 4, 5)
    bad = d.copy()
    bad[0, 0] = 3
    with pytest.raises(ValueError, match="0,1 or 2"):
        report.preprocess_table(sym, bad, 0.25, 4, 4)


def test_prepare_inputs_follows_gwaspa_filtering():
    genes = ["A", "B", "C", "D", "Z"]                 # Z has data but is not in the knowledge base
    data = np.arange(5 * 4).reshape(5, 4) % 2
    ents_uid = [40, 10, 20, 30, 60, 70, 11]
    ents_sym = ["D", "A", "B", "C", "-1", "Q", "A"]    # "-1" dropped; second "A" is a duplicated symbol; Q has no data
    #           A->B      B->C       C->C loop  A->Q (target without data)  D->Q        Q->A (source without data)  dup
    rel = [(10, 20, 1), (20, 30, -1), (30, 30, 1), (10, 70, 1),            (40, 70, -1), (70, 10, 1),                (10, 20, 1)]
    p = report.prepare_inputs(genes, data, ents_uid, ents_sym, *zip(*rel))
    assert p.ents_uid.tolist() == [10, 20, 30] and p.ents_symbol == ["A", "B", "C"]        # D only points outside
    assert p.ents2_uid.tolist() == [10, 20, 30, 40] and p.ents2_symbol == ["A", "B", "C", "D"]
    assert (p.src.tolist(), p.trg.tolist(), p.sign.tolist()) == ([0, 1], [1, 2], [1, -1])
    assert p.data1.tolist() == data[[0, 1, 2]].tolist() and p.data2.tolist() == data[[0, 1, 2, 3]].tolist()
    with pytest.raises(ValueError, match="two different signs"):
        report.prepare_inputs(genes, data, ents_uid, ents_sym, *zip(*(rel + [(10, 20, -1)])))


def test_results_table_order_and_pvalues():
    """order(Pvalues, -Scores), p = #(TestScores >= score)/K against the f32 maxima (ProcessPaths.R:316,324)."""
    null = np.array([0.5, 1.5, 2.5, 3.5], dtype=np.float32)
    rd = {"srcuid": ENTS[0]}
    lst1 = api.JoinResult(np.array([1.0, 3.0]), np.array([0, 1], np.int32), np.array([0, 1], np.int32),
                          np.array([2, 3], np.int32), np.array([1, 0], np.int32), null)
    lst2 = api.JoinResult(np.array([2.0, 3.0, 9.0]), np.array([0, 1, 2], np.int32), np.array([0, 1, 3], np.int32),
                          np.array([2, 3, 4], np.int32), np.array([1, 0, 0], np.int32), null)
    df = report.results_table({"lst1": lst1, "lst2": lst2}, 2, {"rels_data": rd, "rels_data2": rd, "rels": RELS}, ENTS, ENTS)
    assert list(df.columns) == report.COLUMNS
    assert df["Pvalues"].tolist() == [0.0, 0.25, 0.25, 0.5, 0.75]
    assert df["Scores"].tolist() == [9.0, 3.0, 3.0, 2.0, 1.0]
    assert df["Lengths"].tolist() == [2, 1, 2, 2, 1]                    # stable: level 1's 3.0 precedes level 2's
    assert df["Paths"].tolist() == ["C -> D", "B", "B -> C", "A -> B", "A"]
    assert df["SignedPaths"].tolist()[0] == "C (+) -> D (-)"
    assert df["Cases"].tolist() == [4.0, 3.0, 3.0, 2.0, 2.0]
    # K = 0: TestScores empty -> NaN p-values (App. A-3), rows then ordered by score alone
    e = np.zeros(0, np.float32)
    lst1k = api.JoinResult(lst1.scores, lst1.src, lst1.trg, lst1.cases, lst1.ctrls, e)
    df0 = report.results_table({"lst1": lst1k}, 1, {"rels_data2": rd}, ENTS, ENTS)
    assert np.isnan(df0["Pvalues"]).all() and df0["Scores"].tolist() == [3.0, 1.0]


def test_check_input_messages():
    ok = dict(n_cases=5, n_ctrls=5, method="method1", threshold=0.05, top_k=10, path_length=5, iterations=100)
    report.check_input(**ok)
    for key, val, msg in (("n_cases", 1, "nCases"), ("n_ctrls", 2.5, "nControls"), ("top_k", 10001, "K must"),
                          ("method", "enrich", "method must"), ("threshold", 0, "threshold_percent"),
                          ("path_length", 6, "pathLength"), ("iterations", -1, "iterations"), ("n_cases", 65535, "65536")):
        with pytest.raises(ValueError, match=msg):
            report.check_input(**{**ok, key: val})


def test_read_dataset(tmp_path):
    f = tmp_path / "d.txt"
    f.write_text('symbols p1 p2 p3\n"TP53" 1 0 2\nBRCA1 0 0 1\n')
    sym, pats, m = report.read_dataset(str(f))
    assert sym == ["TP53", "BRCA1"] and pats == ["p1", "p2", "p3"] and m.tolist() == [[1, 0, 2], [0, 0, 1]]
    f.write_text("Gene p1\nX 1\n")
    with pytest.raises(ValueError, match="symbols"):
        report.read_dataset(str(f))
