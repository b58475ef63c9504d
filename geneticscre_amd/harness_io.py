"""Reader / writer for the text dump consumed by the reference's stand-alone harness.

Format (one record per line, everything up to the first space is a free label -- test/test.cpp:23-30):
path_length, num_cases, num_ctrls (test/harness.cpp:38-40); six {uids ``src:trg:count:location``, signs}
pairs for levels 1a, 1b, 2, 3, 4, 5 (harness.cpp:75-97, test.cpp:32-74); four data-index lines
(harness.cpp:99-102); data1, data2, perms as ``r,r,r r,r,r`` rows (test.cpp:76-96); the value table
(test.cpp:98-118).  The reference ships neither a writer nor a sample; SURVEY.md Appendix B documents it.
"""
from __future__ import annotations

from typing import List

import numpy as np

from .synth import Problem
from .uids import LevelTables, UidRelSet

_LEVELS = [("1a", 1), ("1b", 1), ("2", 2), ("3", 3), ("4", 4), ("5", 5)]


def _payload(line: str) -> str:
    parts = line.rstrip("\n").split(" ", 1)
    return parts[1] if len(parts) > 1 else ""


def _matrix(text: str, dtype) -> np.ndarray:
    rows = [[dtype(v) for v in row.split(",")] for row in text.split(" ") if row]
    if not rows:
        return np.zeros((0, 0), dtype=dtype)
    return np.array(rows, dtype=dtype)


def read_problem(path: str, method: str = "method2", iterations: int = 10, top_k: int = 12,
                 path_length: int | None = None) -> Problem:
    """Parse a harness dump.  Defaults follow the harness flags (-m method2, -p 10, top_k 12; harness.cpp:42-56)."""
    with open(path) as f:
        lines: List[str] = [_payload(l) for l in f if l.strip()]
    it = iter(lines)
    file_len, n_cases, n_ctrls = int(next(it)), int(next(it)), int(next(it))
    uids, n_paths = {}, {}
    for name, plen in _LEVELS:
        recs = [tuple(int(v) for v in tok.split(":")) for tok in next(it).split(" ") if tok]
        signs = [int(v) for v in next(it).split(" ") if v]
        arr = np.array(recs, dtype=np.int64).reshape(-1, 4)
        uids[name] = UidRelSet(plen, arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3], signs)
        n_paths[name] = uids[name].count_total_paths()
    data_inds = {name: np.array([int(v) for v in next(it).split(" ") if v], dtype=np.int32)
                 for name in ("1a", "1b", "2", "3")}
    data1 = _matrix(next(it), int).astype(np.int32)
    data2 = _matrix(next(it), int).astype(np.int32)
    perms = _matrix(next(it), int).astype(np.int32)
    table = _matrix(next(it), float).astype(np.float64)
    levels = LevelTables(uids, data_inds, {}, n_paths)
    return Problem(method, n_cases, n_ctrls, file_len if path_length is None else path_length, top_k, iterations,
                   levels, data1, data2, table, perms)


def write_problem(path: str, p: Problem) -> None:
    def rows(m, fmt):
        return " ".join(",".join(fmt(v) for v in row) for row in m)

    with open(path, "w") as f:
        f.write(f"path_length {p.path_length}\nnum_cases {p.n_cases}\nnum_ctrls {p.n_ctrls}\n")
        for i, (name, _) in enumerate(_LEVELS):
            u = p.levels.uids[name]
            f.write(f"uids{i} " + " ".join(f"{s}:{t}:{c}:{l}" for s, t, c, l in
                                           zip(u.src.tolist(), u.trg.tolist(), u.count.tolist(), u.location.tolist())) + "\n")
            f.write(f"sign{i} " + " ".join(str(v) for v in u.signs.tolist()) + "\n")
        for i, name in enumerate(("1a", "1b", "2", "3")):
            f.write(f"idx{i} " + " ".join(str(v) for v in p.levels.data_inds[name].tolist()) + "\n")
        f.write("data1 " + rows(p.data1.tolist(), str) + "\n")
        f.write("data2 " + rows(p.data2.tolist(), str) + "\n")
        f.write("perms " + rows(p.perm_cases.tolist(), str) + "\n")
        f.write("table " + rows(p.value_table.tolist(), repr) + "\n")
