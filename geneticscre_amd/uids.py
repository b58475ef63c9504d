"""Join-index tables for the path-join scorer (host side).

``UidRelSet`` mirrors the reference's per-level join index (src/gcre.h:49-90, uid_ref in
src/gcre_types.h:50-56).  ``assemble_uids`` restates the uid resolution of src/wrapper.cpp:99-140;
``build_level_tables`` restates the per-level tables GWASPA builds in R (R/ProcessPaths.R:214-256,
getUidsCountsLocations R/PathMethods.R:133-152, getRels3 src/wrapper.cpp:18-48) for a network that is
already filtered to the genes of the dataset.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Mapping, Sequence, Tuple

import numpy as np


@dataclass
class UidRelSet:
    """One join level: row k of ``paths0`` is joined with rows location[k] .. location[k]+count[k]-1 of ``paths1``."""

    path_length: int
    src: np.ndarray          # int32 [U]  uid_ref.src
    trg: np.ndarray          # int32 [U]  uid_ref.trg
    count: np.ndarray        # int32 [U]  uid_ref.count
    location: np.ndarray     # int64 [U]  uid_ref.location (-1 allowed where count == 0, PathMethods.R:147)
    signs: np.ndarray        # int32      UidRelSet::signs
    path_idx: np.ndarray = field(default=None)   # int64 [U+1] prefix sum of count (wrapper.cpp:128-130)

    def __post_init__(self):
        self.src = np.ascontiguousarray(self.src, dtype=np.int32)
        self.trg = np.ascontiguousarray(self.trg, dtype=np.int32)
        self.count = np.ascontiguousarray(self.count, dtype=np.int32)
        self.location = np.ascontiguousarray(self.location, dtype=np.int64)
        self.signs = np.ascontiguousarray(self.signs, dtype=np.int32)
        self.path_idx = np.zeros(len(self.count) + 1, dtype=np.int64)
        np.cumsum(np.maximum(self.count, 0), out=self.path_idx[1:])

    def __len__(self) -> int:
        return len(self.count)

    def count_total_paths(self) -> int:           # gcre.h:83-88
        return int(self.path_idx[-1])

    def need_flip(self, idx: int, loc: int) -> bool:
        """gcre.h:71-81.  True means: do NOT swap path1's pos/neg halves (methods.h:140-142)."""
        if self.path_length > 3:
            sign = self.signs[idx]
        elif self.path_length < 3:
            sign = self.signs[loc]
        else:
            sign = -1 if self.signs[idx] + self.signs[loc] == 0 else 1
        return sign == 1


def assemble_uids(path_length: int, src_uids: Sequence[int], trg_uids: Sequence[int],
                  count_locs: Mapping[int, Tuple[int, int]], signs: Sequence[int]) -> UidRelSet:
    """src/wrapper.cpp:99-140: row k takes (count, location) of the entry keyed by its *target* uid;
    a missing key yields (0, 0) (unordered_map::operator[] default)."""
    trg = np.asarray(trg_uids, dtype=np.int32)
    count = np.zeros(len(trg), dtype=np.int32)
    location = np.zeros(len(trg), dtype=np.int64)
    for k, t in enumerate(trg.tolist()):
        c, l = count_locs.get(int(t), (0, 0))
        count[k], location[k] = c, l
    return UidRelSet(path_length, np.asarray(src_uids, dtype=np.int32), trg, count, location,
                     np.asarray(signs, dtype=np.int32))


def count_locations(rels1_trgs: np.ndarray, rels2_srcs: np.ndarray) -> Dict[int, Tuple[int, int]]:
    """getUidsCountsLocations, R/PathMethods.R:133-152.  ``rels2_srcs`` must be sorted (run-length encoded
    in the reference).  Targets with no outgoing relation get (0, -1)."""
    out: Dict[int, Tuple[int, int]] = {}
    srcs = np.asarray(rels2_srcs)
    if len(srcs):
        change = np.flatnonzero(np.r_[True, srcs[1:] != srcs[:-1]])
        lengths = np.diff(np.r_[change, len(srcs)])
        for u, c, l in zip(srcs[change].tolist(), lengths.tolist(), change.tolist()):
            out[int(u)] = (int(c), int(l))
    for t in np.unique(np.asarray(rels1_trgs)).tolist():
        if int(t) not in out:
            out[int(t)] = (0, -1)
    return out


def _resolve(trg: np.ndarray, srcs_sorted: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Vectorised count_locations + assemble_uids for sorted ``srcs_sorted``."""
    lo = np.searchsorted(srcs_sorted, trg, side="left")
    hi = np.searchsorted(srcs_sorted, trg, side="right")
    count = (hi - lo).astype(np.int32)
    location = np.where(count > 0, lo, -1).astype(np.int64)
    return count, location


@dataclass
class LevelTables:
    """Everything ProcessPaths needs besides the data, value table and permutations."""

    uids: Dict[str, UidRelSet]        # keys "1a", "1b", "2", "3", "4", "5"
    data_inds: Dict[str, np.ndarray]  # keys "1a", "1b", "2", "3" (0-based rows of data1 / data2)
    rels3: Dict[str, np.ndarray]      # srcuid, trguid, sign, trguid2, sign2 (getRels3)
    n_paths: Dict[str, int]


def build_level_tables(n_genes: int, src: np.ndarray, trg: np.ndarray, sign: np.ndarray,
                       max_level: int = 5) -> LevelTables:
    """R/ProcessPaths.R:206-256 for a network whose genes 0..n_genes-1 all carry data and all occur in an edge.

    ``src, trg, sign`` are the relations sorted by (src, trg), no self loops, no duplicates
    (ProcessPaths.R:145-149,210).  Gene uid == row of the data matrix, so data_inds are the uids themselves.
    """
    src = np.asarray(src, dtype=np.int32)
    trg = np.asarray(trg, dtype=np.int32)
    sign = np.asarray(sign, dtype=np.int32)
    order = np.lexsort((trg, src))
    if not np.array_equal(order, np.arange(len(src))):
        raise ValueError("relations must be sorted by (src, trg)")
    genes = np.arange(n_genes, dtype=np.int32)
    ones = np.ones(n_genes, dtype=np.int32)
    uids: Dict[str, UidRelSet] = {}
    data_inds: Dict[str, np.ndarray] = {}

    # level 1 (ProcessPaths.R:214-224): every gene joined with its own data row
    self_loc = np.arange(n_genes, dtype=np.int64)
    uids["1a"] = UidRelSet(1, genes, genes, ones, self_loc, ones)
    data_inds["1a"] = genes.copy()
    # Ents2 = genes that are the source of some relation (ProcessPaths.R:153-155); data2 holds their rows
    genes2 = np.unique(src).astype(np.int32)
    uids["1b"] = UidRelSet(1, genes2, genes2, np.ones(len(genes2), np.int32),
                           np.arange(len(genes2), dtype=np.int64), np.ones(len(genes2), np.int32))
    data_inds["1b"] = np.arange(len(genes2), dtype=np.int32)

    # level 2 (ProcessPaths.R:226-230): gene g joined with the data of each of its targets
    c2, l2 = _resolve(genes, src)
    uids["2"] = UidRelSet(2, genes, genes, c2, l2, sign)
    data_inds["2"] = trg.copy()

    # level 3 (ProcessPaths.R:232-236): edge a->b joined with the data of each target of b
    c3, l3 = _resolve(trg, src)
    uids["3"] = UidRelSet(3, src, trg, c3, l3, sign)
    data_inds["3"] = trg.copy()

    # Rels3 (getRels3, wrapper.cpp:18-48): one row per 2-edge walk a->b->c, grouped by the first edge
    rep = np.repeat(np.arange(len(src)), np.maximum(c3, 0))
    start = np.repeat(l3, np.maximum(c3, 0))
    within = np.arange(len(rep)) - np.repeat(np.cumsum(np.maximum(c3, 0)) - np.maximum(c3, 0), np.maximum(c3, 0))
    second = (start + within).astype(np.int64)
    rels3 = {
        "srcuid": src[rep], "trguid": trg[rep], "sign": sign[rep],
        "trguid2": trg[second], "sign2": sign[second],
    }
    # sign of the third gene relative to a (+) first gene (ProcessPaths.R:243-245)
    third_sign = np.where(rels3["sign"] * rels3["sign2"] == -1, -1, 1).astype(np.int32)

    # level 4 (ProcessPaths.R:247-250): 3-path a->b->c joined with the stored 2-paths c->d
    c4, l4 = _resolve(rels3["trguid2"], src)
    uids["4"] = UidRelSet(4, rels3["srcuid"], rels3["trguid2"], c4, l4, third_sign)
    # level 5 (ProcessPaths.R:253-256): 3-path a->b->c joined with the stored 3-paths c->d->e
    c5, l5 = _resolve(rels3["trguid2"], rels3["srcuid"])
    uids["5"] = UidRelSet(5, rels3["srcuid"], rels3["trguid2"], c5, l5, third_sign)

    # tables above max_level are cheap to keep; ProcessPaths simply never joins them
    n_paths = {k: u.count_total_paths() for k, u in uids.items()}
    return LevelTables(uids, data_inds, rels3, n_paths)
