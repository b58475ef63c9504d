"""Synthetic inputs for the path-join scorer: STRINGdb-shaped signed networks, rare-variant matrices,
label permutations and the hypergeometric value table.

The reference reads its network from inst/extdata/Rels.dat, which is not shipped
(/root/reference/.MISSING_LARGE_BLOBS), so every test and benchmark here runs on generated networks
(SURVEY.md §8d).  The generators restate the *input-producing* R code next to the hot path:

* ``values_table``       -- getValuesTable, R/Utils.R:137-159
* ``case_or_control``    -- getRandIndicesMat + getCaseORControl, R/Utils.R:22-46, 246-262
* ``variant_matrix``     -- the shape PreprocessTable leaves behind, R/Utils.R:164-197

Value-table parity with R's ``stats::dhyper`` is unpinned (no R in this image; the builder restates R's published
algorithm); the hot path treats the table as opaque input, so bit-exactness of the scorer is defined given identical
tables.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .uids import LevelTables, build_level_tables


def values_table(n_cases: int, n_ctrls: int) -> np.ndarray:
    """-log of the two-sided hypergeometric p-value of seeing x cases among i carriers (Utils.R:137-159).

    Returns float64 [(n_cases+1), (n_ctrls+1)], entry [x][i-x]; infinities become max finite + 1 (:156).
    One builder for the product, the tests and the fixtures: the native ``gcre_values_table`` (host code of
    libgcre_hip.so, no GPU needed), which restates R's ``stats::dhyper`` as nmath publishes it and compares with R's
    exact ``<=``.  Its independent Python restatement lives with the test infrastructure (oracle/values_table.py).
    """
    from . import api
    return api.values_table(n_cases, n_ctrls)


def case_or_control(n_cases: int, n_ctrls: int, n_perm: int, rng: np.random.Generator,
                    strata: Optional[np.ndarray] = None) -> np.ndarray:
    """K x n int32 matrix, 1 = the patient keeps its label under permutation r (Utils.R:246-262).

    Each row comes from a uniform permutation of the patient columns (Utils.R:22-46); with ``strata`` the
    permutation only moves patients inside their stratum (Utils.R:8-13).
    """
    n = n_cases + n_ctrls
    out = np.empty((n_perm, n), dtype=np.int32)
    was_case = np.arange(n) < n_cases
    for r in range(n_perm):
        if strata is None:
            perm = rng.permutation(n)
        else:
            perm = np.arange(n)
            for s in np.unique(strata):
                idx = np.flatnonzero(strata == s)
                perm[idx] = rng.permutation(idx)
        out[r] = ((perm < n_cases) == was_case).astype(np.int32)
    return out


def variant_matrix(n_genes: int, n_patients: int, rng: np.random.Generator, threshold: float = 0.05,
                   fixed_rate: float = 0.0) -> np.ndarray:
    """genes x patients 0/1 int32, per-gene carrier rate heavy at the rare end, never above ``threshold``.
    ``fixed_rate`` > 0: every gene at that carrier rate (the sensitivity sweep of profiles/: how the kernels hold up when
    every gene sits at 1 %, 2.5 % or the 5 % admission limit of R/Utils.R:185-188)."""
    rate = threshold * rng.random(n_genes) ** 3
    if fixed_rate > 0:
        rate = np.full(n_genes, min(fixed_rate, threshold))
    data = (rng.random((n_genes, n_patients)) < rate[:, None]).astype(np.int32)
    cap = int(threshold * (n_patients + 1))            # freqs <= threshold * ncol(df), Utils.R:185-187
    for g in np.flatnonzero(data.sum(axis=1) > cap):
        on = np.flatnonzero(data[g])
        data[g, rng.choice(on, size=len(on) - cap, replace=False)] = 0
    for g in np.flatnonzero(data.sum(axis=1) == 0):     # every kept gene has at least one carrier
        data[g, rng.integers(n_patients)] = 1
    return data


def signed_network(n_genes: int, n_edges: int, rng: np.random.Generator, alpha: float = 1.5,
                   p_positive: float = 0.7):
    """Directed signed relations with heavy-tailed in/out degree, no self loops, unique, sorted by (src, trg)."""
    w_out = rng.pareto(alpha, n_genes) + 1.0
    w_in = rng.pareto(alpha, n_genes) + 1.0
    w_out /= w_out.sum()
    w_in /= w_in.sum()
    pairs = np.empty((0, 2), dtype=np.int64)
    need = n_edges
    for _ in range(64):
        s = rng.choice(n_genes, size=int(need * 1.3) + 16, p=w_out)
        t = rng.choice(n_genes, size=len(s), p=w_in)
        ok = s != t
        pairs = np.unique(np.vstack([pairs, np.stack([s[ok], t[ok]], axis=1)]), axis=0)
        if len(pairs) >= n_edges:
            break
        need = n_edges - len(pairs)
    if len(pairs) > n_edges:
        pairs = pairs[np.sort(rng.choice(len(pairs), size=n_edges, replace=False))]
    # every gene must occur in some relation (ProcessPaths.R:162-163 drops the others): renumber
    used = np.unique(pairs)
    remap = -np.ones(n_genes, dtype=np.int64)
    remap[used] = np.arange(len(used))
    pairs = remap[pairs]
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs = pairs[order]
    sign = np.where(rng.random(len(pairs)) < p_positive, 1, -1).astype(np.int32)
    return len(used), pairs[:, 0].astype(np.int32), pairs[:, 1].astype(np.int32), sign


@dataclass
class Problem:
    """All inputs of one ProcessPaths call (src/wrapper.cpp:177-185), as plain arrays."""

    method: str
    n_cases: int
    n_ctrls: int
    path_length: int
    top_k: int
    iterations: int
    levels: LevelTables
    data1: np.ndarray         # genes x patients, rows = Ents order
    data2: np.ndarray         # rows = Ents2 order (genes with outgoing relations)
    value_table: np.ndarray
    perm_cases: np.ndarray    # K x patients (may have 0 rows when iterations == 0)
    seed: int = 0

    def total_scores(self) -> int:
        """path x permutation scores of one ProcessPaths pass (SURVEY.md §8d unit of work)."""
        names = ["1a", "1b"] + [str(l) for l in range(2, self.path_length + 1)]
        return self.iterations * sum(self.levels.n_paths[k] for k in names)


def make_problem(n_genes: int, n_edges: int, n_cases: int, n_ctrls: int, n_perm: int, path_length: int,
                 method: str = "method1", top_k: int = 12, seed: int = 0, threshold: float = 0.05,
                 table: Optional[np.ndarray] = None) -> Problem:
    rng = np.random.default_rng(seed)
    g, src, trg, sign = signed_network(n_genes, n_edges, rng)
    levels = build_level_tables(g, src, trg, sign)
    data1 = variant_matrix(g, n_cases + n_ctrls, rng, threshold)
    data2 = data1[levels.uids["1b"].src]
    vt = values_table(n_cases, n_ctrls) if table is None else table
    perms = case_or_control(n_cases, n_ctrls, n_perm, rng) if n_perm > 0 else np.zeros((0, 0), np.int32)
    return Problem(method, n_cases, n_ctrls, path_length, top_k, n_perm, levels, data1, data2, vt, perms, seed)


def packed_case_masks(n_cases: int, n_ctrls: int, n_perm: int, rng: np.random.Generator) -> np.ndarray:
    """uint64 [n_perm][ceil(n/64)]: bit c of row r = patient c is a case under permutation r.

    Exactly what setPermutedCases derives from ``case_or_control`` (join_base.cpp:85-125: case_mask XOR flipped),
    without materialising the K x n int matrix -- mask bit c = (perm[c] < n_cases).
    """
    n = n_cases + n_ctrls
    W = (n + 63) // 64
    out = np.zeros((n_perm, W), dtype=np.uint64)
    weights = (np.uint64(1) << np.arange(64, dtype=np.uint64))
    for r in range(n_perm):
        bits = np.zeros(W * 64, dtype=np.uint64)
        bits[:n] = rng.permutation(n) < n_cases
        out[r] = (bits.reshape(W, 64) * weights).sum(axis=1, dtype=np.uint64)
    return out


def masks_from_case_or_control(perm_cases: np.ndarray, n_cases: int) -> np.ndarray:
    """Pack a K x n "label kept" matrix (Utils.R:246-262) the way setPermutedCases does (join_base.cpp:97-111)."""
    K, n = perm_cases.shape
    W = (n + 63) // 64
    is_case = np.arange(n) < n_cases
    bits = np.zeros((K, W * 64), dtype=np.uint64)
    bits[:, :n] = is_case[None, :] ^ (perm_cases != 1)
    weights = (np.uint64(1) << np.arange(64, dtype=np.uint64))
    return (bits.reshape(K, W, 64) * weights).sum(axis=2, dtype=np.uint64)
