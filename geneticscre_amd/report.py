"""What GWASPA does with the lists ProcessPaths returns, and the non-R front end around the device path.

  * ``get_paths``        id pairs -> "a -> b -> c" / "a (+) -> b (-) -> c (-)" strings  (R/PathMethods.R:2-131)
  * ``uid_to_symbol``    uid paths -> gene-symbol paths                                  (R/Utils.R:50-95)
  * ``results_table``    p-values, the GWASPA.Results columns and their order            (R/ProcessPaths.R:272-326)
  * ``preprocess_table`` / ``prepare_inputs``  the dataset and network filtering         (R/Utils.R:162-199, ProcessPaths.R:131-176)
  * ``gwaspa``           the whole call, with the table / level tables / permutations built natively

Host-side post-processing of <= top_k x 5 rows: string work, nothing here touches the scored path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

POS, NEG = "(+)", "(-)"


def _next_dirs(prev: np.ndarray, sign: np.ndarray) -> np.ndarray:
    """dirs[j+1] from dirs[j] and the edge sign: (-) iff (sign==1 & prev (-)) | (sign==-1 & prev (+)); any other sign
    value leaves (+) (PathMethods.R:28-31 and its copies)."""
    neg = ((sign == 1) & prev) | ((sign == -1) & ~prev)
    return neg


def get_paths(ids: np.ndarray, path_length: int, rels1: Dict[str, np.ndarray], rels2: Dict[str, np.ndarray]
              ) -> Tuple[List[str], List[str]]:
    """getPaths (R/PathMethods.R:2-131).  ``ids`` is the 1-based [n][2] matrix of ``JoinResult.as_r_list()["ids"]``.

    rels1 / rels2 are the frames GWASPA passes per level (ProcessPaths.R:278-291): dicts with "srcuid" and, where the
    level needs them, "trguid", "sign", "trguid2", "sign2".  An id above nrow(rels2) names the sign-flipped copy
    of a path (the reference never emits one -- methods.h:245-247 is commented out -- but getPaths decodes it).
    Sentinel rows (ids (0,0), App. A-8) decode to "NA" the way R's out-of-range indexing does.
    """
    ids = np.asarray(ids, dtype=np.int64).reshape(-1, 2)
    i1, i2 = ids[:, 0] - 1, ids[:, 1] - 1
    n2 = len(rels2["srcuid"])
    pivot = len(rels1["srcuid"]) if path_length == 3 else n2           # PathMethods.R:43 uses nrow(Rels1)
    first_neg = ids[:, 1] > pivot
    i3 = np.where(ids[:, 1] > n2, i2 - n2, i2)

    def col(frame, name, idx):
        a = np.asarray(frame[name])
        ok = (idx >= 0) & (idx < len(a))
        out = np.where(ok, a[np.clip(idx, 0, max(len(a) - 1, 0))] if len(a) else 0, 0)
        return out, ok

    if path_length == 1:
        genes = [col(rels2, "srcuid", np.where(i2 >= n2, i2 - n2, i2))]
        signs = []
        first_neg = i2 >= n2
    elif path_length == 2:
        genes = [col(rels2, "srcuid", i3), col(rels2, "trguid", i3)]
        signs = [col(rels2, "sign", i3)]
    elif path_length == 3:
        genes = [col(rels1, "srcuid", i1), col(rels1, "trguid", i1), col(rels2, "trguid", i3)]
        signs = [col(rels1, "sign", i1), col(rels2, "sign", i3)]
    elif path_length == 4:
        genes = [col(rels1, "srcuid", i1), col(rels1, "trguid", i1), col(rels1, "trguid2", i1), col(rels2, "trguid", i3)]
        signs = [col(rels1, "sign", i1), col(rels1, "sign2", i1), col(rels2, "sign", i3)]
    elif path_length == 5:
        genes = [col(rels1, "srcuid", i1), col(rels1, "trguid", i1), col(rels1, "trguid2", i1),
                 col(rels2, "trguid", i3), col(rels2, "trguid2", i3)]
        signs = [col(rels1, "sign", i1), col(rels1, "sign2", i1), col(rels2, "sign", i3), col(rels2, "sign2", i3)]
    else:
        raise ValueError("path_length must be 1..5")

    dirs = [np.asarray(first_neg, dtype=bool)]
    for s, _ok in signs:
        dirs.append(_next_dirs(dirs[-1], s))

    def name(g, ok, r):
        return str(int(g[r])) if ok[r] else "NA"

    paths, signpaths = [], []
    for r in range(len(ids)):
        paths.append(" -> ".join(name(g, ok, r) for g, ok in genes))
        signpaths.append(" -> ".join(f"{name(g, ok, r)} {NEG if d[r] else POS}" for (g, ok), d in zip(genes, dirs)))
    return paths, signpaths


def uid_to_symbol(ents_uid: Sequence[int], ents_symbol: Sequence[str], paths: Sequence[str], signed: bool = False
                  ) -> List[str]:
    """Uid2Symbol / SignUid2SignSymbol (R/Utils.R:50-95): replace every uid by its symbol, "NA" when unknown."""
    table = {int(u): s for u, s in zip(ents_uid, ents_symbol)}
    out = []
    for p in paths:
        hops = []
        for hop in p.split(" -> "):
            if signed:
                uid, _, tag = hop.partition(" ")
                hops.append(f"{table.get(int(uid), 'NA') if uid != 'NA' else 'NA'} {tag}")
            else:
                hops.append(table.get(int(hop), "NA") if hop != "NA" else "NA")
        out.append(" -> ".join(hops))
    return out


COLUMNS = ["SignedPaths", "Paths", "Lengths", "Scores", "Pvalues", "Cases", "Controls"]


def results_table(level_results: Dict[str, object], path_length: int, frames: Dict[str, Dict[str, np.ndarray]],
                  ents: Tuple[Sequence[int], Sequence[str]], ents2: Tuple[Sequence[int], Sequence[str]]):
    """GWASPA.Results (R/ProcessPaths.R:272-326) as a pandas DataFrame.

    ``level_results`` is what ``api.process_paths`` returns ("lst1".."lst5" -> JoinResult); ``frames`` holds
    "rels_data", "rels_data2", "rels", "rels3" (uid-valued).  Rows are ordered by p-value ascending then score
    descending with R's stable ``order``; p-values compare the f64 score with the f32-rounded maxima (App. A-7).
    """
    import pandas as pd

    per_level = {1: ("rels_data2", "rels_data2"), 2: ("rels_data", "rels"), 3: ("rels", "rels"),
                 4: ("rels3", "rels"), 5: ("rels3", "rels3")}
    cols: Dict[str, list] = {c: [] for c in COLUMNS}
    for L in range(1, path_length + 1):
        lst = level_results[f"lst{L}"]
        r = lst.as_r_list()
        f1, f2 = per_level[L]
        paths, signpaths = get_paths(r["ids"], L, frames[f1], frames[f2])
        who = ents2 if L == 1 else ents
        cols["SignedPaths"] += uid_to_symbol(who[0], who[1], signpaths, signed=True)
        cols["Paths"] += uid_to_symbol(who[0], who[1], paths)
        cols["Lengths"] += [L] * len(paths)
        cols["Scores"] += r["scores"].tolist()
        cols["Pvalues"] += lst.pvalues().tolist()
        cols["Cases"] += r["cases"].tolist()
        cols["Controls"] += r["controls"].tolist()
    df = pd.DataFrame(cols, columns=COLUMNS)
    p = df["Pvalues"].to_numpy()
    order = np.lexsort((-df["Scores"].to_numpy(), np.where(np.isnan(p), np.inf, p)))   # order(): NA last, stable
    return df.iloc[order].reset_index(drop=True)


# ---------------------------------------------------------------------------------------------------------------
# inputs


def check_input(n_cases, n_ctrls, method, threshold, top_k, path_length, iterations):
    """check_input (R/Utils.R:203-241); same conditions, ValueError instead of stop()."""
    def integer(v):
        return isinstance(v, (int, np.integer)) and not isinstance(v, bool)
    if not integer(n_cases) or n_cases < 2:
        raise ValueError("nCases must be an integer >= 2!")
    if not integer(n_ctrls) or n_ctrls < 2:
        raise ValueError("nControls must be an integer >= 2!")
    if not integer(top_k) or top_k < 1 or top_k > 10000:
        raise ValueError("K must be an integer >= 1 and <= 10000!")
    if not (isinstance(method, str) and method[:7] in ("method1", "method2")):
        raise ValueError("method must be one of: 'method1', 'method2'!")
    if not isinstance(threshold, (int, float)) or threshold > 1 or threshold <= 0:
        raise ValueError("threshold_percent must be a real number > 0 and <= 1!")
    if not integer(path_length) or path_length > 5 or path_length < 1:
        raise ValueError("pathLength must be an integer >= 1 and <= 5!")
    if not integer(iterations) or iterations < 0:
        raise ValueError("iterations must be an integer greater than or equal to 0!")
    if n_cases + n_ctrls > 65536:
        raise ValueError("Package cannot process data set with more than 65536 columns!")


def read_dataset(path: str) -> Tuple[List[str], List[str], np.ndarray]:
    """The whitespace table PreprocessTable reads (Utils.R:164-168): header ``symbols p1 p2 ...``, one gene per row."""
    with open(path) as fh:
        header = fh.readline().split()
        if not header or header[0].strip('"') != "symbols":
            raise ValueError("First column must be named symbols and must contain the gene symbols!")
        symbols, rows = [], []
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            symbols.append(parts[0].strip('"'))
            rows.append(np.array(parts[1:], dtype=np.int32))
    return symbols, [h.strip('"') for h in header[1:]], np.vstack(rows) if rows else np.zeros((0, len(header) - 1), np.int32)


def preprocess_table(symbols: Sequence[str], data: np.ndarray, threshold: float, n_cases: int, n_ctrls: int
                     ) -> Tuple[List[str], np.ndarray]:
    """PreprocessTable (R/Utils.R:162-199): drop NA / duplicated symbols, 2 -> 1, check the shape and the 0/1
    alphabet, keep genes with at most ``threshold * ncol(df)`` variant carriers (ncol counts the symbols column)."""
    data = np.array(data, dtype=np.int32)
    keep, seen = [], set()
    for i, s in enumerate(symbols):
        if s is None or s == "NA" or s in seen:
            continue
        seen.add(s)
        keep.append(i)
    data = data[keep]
    genes = [symbols[i] for i in keep]
    data[data == 2] = 1
    if data.shape[1] != n_cases + n_ctrls:
        raise ValueError("The number of patients in the dataset must be equal to nCases + nControls!")
    if ((data != 0) & (data != 1)).any():
        raise ValueError("The patient columns must consist of only 0,1 or 2 entries!")
    freqs = data.sum(axis=1)
    sel = np.flatnonzero(freqs <= threshold * (data.shape[1] + 1))
    return [genes[i] for i in sel], data[sel]


@dataclass
class Prepared:
    """The frames GWASPA holds after filtering the network against the dataset (ProcessPaths.R:131-176, 206-212)."""

    ents_uid: np.ndarray          # genes with data that occur in a kept relation, ascending uid
    ents_symbol: List[str]
    ents2_uid: np.ndarray         # genes with data that are the source of any relation (targets may lack data)
    ents2_symbol: List[str]
    src: np.ndarray               # kept relations sorted by (src, trg) -- ranks into ents_uid
    trg: np.ndarray
    sign: np.ndarray
    data1: np.ndarray             # rows follow ents_uid
    data2: np.ndarray             # rows follow ents2_uid


def prepare_inputs(genes: Sequence[str], data: np.ndarray, ents_uid: Sequence[int], ents_symbol: Sequence[str],
                   rel_src: Sequence[int], rel_trg: Sequence[int], rel_sign: Sequence[int]) -> Prepared:
    """ProcessPaths.R:131-176: intersect the knowledge base with the dataset.

    Ents: symbol != "-1", first occurrence of each symbol, present in the dataset, ordered by uid.  Relations:
    unique (src, trg, sign) rows whose source is in Ents (these define Ents2); those whose target is too and that
    are not self loops define the joined network and shrink Ents to the genes they touch.
    """
    ents_uid = np.asarray(ents_uid, dtype=np.int64)
    row_of = {}
    for i, g in enumerate(genes):
        row_of.setdefault(g, i)
    keep, seen = [], set()
    for i, s in enumerate(ents_symbol):
        if s == "-1" or s in seen:
            continue
        seen.add(s)
        if s in row_of:
            keep.append(i)
    keep = sorted(keep, key=lambda i: int(ents_uid[i]))
    e_uid = ents_uid[keep]
    e_sym = [ents_symbol[i] for i in keep]
    present = set(e_uid.tolist())

    rels = np.unique(np.stack([np.asarray(rel_src, np.int64), np.asarray(rel_trg, np.int64),
                               np.asarray(rel_sign, np.int64)], axis=1), axis=0) if len(rel_src) else np.zeros((0, 3), np.int64)
    src_ok = np.array([s in present for s in rels[:, 0].tolist()], dtype=bool)
    rels2 = rels[src_ok]
    trg_ok = np.array([t in present for t in rels2[:, 1].tolist()], dtype=bool)
    rels1 = rels2[trg_ok & (rels2[:, 0] != rels2[:, 1])]
    rels1 = rels1[np.lexsort((rels1[:, 1], rels1[:, 0]))]
    if len(rels1) > 1 and (np.diff(rels1[:, 0]) == 0)[np.diff(rels1[:, 1]) == 0].any():
        raise ValueError("the network lists a relation (src, trg) with two different signs")

    sym_of = dict(zip(e_uid.tolist(), e_sym))
    left2 = np.unique(rels2[:, 0])
    e2_uid = np.array([u for u in e_uid.tolist() if u in set(left2.tolist())], dtype=np.int64)
    e2_sym = [sym_of[u] for u in e2_uid.tolist()]
    left = set(np.unique(np.concatenate([rels1[:, 0], rels1[:, 1]])).tolist())
    e1_uid = np.array([u for u in e_uid.tolist() if u in left], dtype=np.int64)
    e1_sym = [sym_of[u] for u in e1_uid.tolist()]

    rank = {u: i for i, u in enumerate(e1_uid.tolist())}
    src = np.array([rank[u] for u in rels1[:, 0].tolist()], dtype=np.int32)
    trg = np.array([rank[u] for u in rels1[:, 1].tolist()], dtype=np.int32)
    data = np.asarray(data, dtype=np.int32)
    d1 = data[[row_of[s] for s in e1_sym]] if len(e1_sym) else np.zeros((0, data.shape[1]), np.int32)
    d2 = data[[row_of[s] for s in e2_sym]] if len(e2_sym) else np.zeros((0, data.shape[1]), np.int32)
    return Prepared(e1_uid, e1_sym, e2_uid, e2_sym, src, trg, rels1[:, 2].astype(np.int32), d1, d2)


def frames_of(prep: Prepared, levels) -> Dict[str, Dict[str, np.ndarray]]:
    """The uid-valued frames getPaths indexes (ProcessPaths.R:206-212, 240): ranks -> uids."""
    u = prep.ents_uid
    r3 = levels.rels3
    return {
        "rels_data": {"srcuid": u},
        "rels_data2": {"srcuid": prep.ents2_uid},
        "rels": {"srcuid": u[prep.src], "trguid": u[prep.trg], "sign": prep.sign},
        "rels3": {"srcuid": u[r3["srcuid"]], "trguid": u[r3["trguid"]], "sign": r3["sign"],
                  "trguid2": u[r3["trguid2"]], "sign2": r3["sign2"]},
    }


def gwaspa(genes: Sequence[str], data: np.ndarray, n_cases: int, n_ctrls: int, network, signed: bool = False,
           threshold: float = 0.05, top_k: int = 10, path_length: int = 5, n_permutations: int = 100,
           strata: Optional[Sequence[int]] = None, seed: int = 0, device: int = 0) -> Dict[str, object]:
    """GWASPA (R/ProcessPaths.R:87-344) without R: dataset -> GWASPA.Results, scored on the MI355X.

    ``network`` = (ents_uid, ents_symbol, rel_src, rel_trg, rel_sign): the knowledge base getStringKB() would load
    (the packaged STRING tables are data, not code -- callers bring their own).  ``strata`` gives one stratum id
    per patient column (what the strata file resolves to, ProcessPaths.R:180-191).  The scoring table, the level
    tables and the permutation masks are built by the native builders; permutations are
    drawn on the device from ``seed``, so two runs with the same seed return identical tables.  Decorated p-values
    (R/DecoratedPvalue.R) are not computed.
    """
    from . import api
    from .synth import Problem
    from .uids import UidRelSet

    method = "method2" if signed else "method1"
    check_input(n_cases, n_ctrls, method, threshold, top_k, path_length, n_permutations)
    genes, data = preprocess_table(genes, data, threshold, n_cases, n_ctrls)
    prep = prepare_inputs(genes, data, *network)
    g = len(prep.ents_uid)
    if g == 0:
        raise ValueError("no gene of the dataset takes part in a relation of the network")
    levels = api.build_levels(g, prep.src, prep.trg, prep.sign)
    # level 1 runs over Ents2, which may hold genes whose relations all point outside the dataset (ProcessPaths.R:150-160)
    n2 = len(prep.ents2_uid)
    ids2 = np.arange(n2, dtype=np.int32)
    levels.uids["1b"] = UidRelSet(1, ids2, ids2, np.ones(n2, np.int32), np.arange(n2, dtype=np.int64), np.ones(n2, np.int32))
    levels.data_inds["1b"] = ids2.copy()
    levels.n_paths["1b"] = n2

    table = api.values_table(n_cases, n_ctrls)
    problem = Problem(method, n_cases, n_ctrls, path_length, top_k, n_permutations, levels, prep.data1, prep.data2,
                      table, np.zeros((0, 0), np.int32), seed)
    ex = api.JoinExec(method, n_cases, n_ctrls, n_permutations, device=device)
    ex.top_k = top_k
    ex.set_value_table(table)
    if n_permutations > 0:
        ex.generate_permutations(seed, strata)
    lsts = api.process_paths(problem, device=device, exec_=ex)
    out = {"GWASPA.Results": results_table(lsts, path_length, frames_of(prep, levels),
                                           (prep.ents_uid, prep.ents_symbol), (prep.ents2_uid, prep.ents2_symbol)),
           "levels": lsts, "prepared": prep}
    ex.close()
    return out
