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


def write_problem_bin(path: str, p: Problem, nthreads: int = 0) -> None:
    """Every ProcessPaths input (src/wrapper.cpp:177-185) as one binary file ("GCREBIN2"): the text format above needs
    ~20 bytes per table cell (half a gigabyte for a 5,000-patient table).  Read by the native harness driver and by the
    partial reference build (oracle/ref_partial/ref_driver.cpp --bin); layout = the order written here."""
    import struct
    M = 1 if p.method == "method1" else 2
    perms = np.asarray(p.perm_cases)
    with open(path, "wb") as f:
        f.write(b"GCREBIN2")
        f.write(struct.pack("<8i", M, p.n_cases, p.n_ctrls, p.iterations, p.top_k, p.path_length, nthreads, perms.shape[0]))
        for name, _ in _LEVELS:
            u = p.levels.uids[name]
            f.write(struct.pack("<2q", len(u.src), len(u.signs)))
            for arr, dt in ((u.src, np.int32), (u.trg, np.int32), (u.count, np.int32), (u.location, np.int64), (u.signs, np.int32)):
                f.write(np.ascontiguousarray(arr, dtype=dt).tobytes())
        for name in ("1a", "1b", "2", "3"):
            idx = np.ascontiguousarray(p.levels.data_inds[name], dtype=np.int32)
            f.write(struct.pack("<q", idx.size))
            f.write(idx.tobytes())
        for m in (p.data1, p.data2, perms):
            m = np.asarray(m)
            rows, cols = (m.shape if m.ndim == 2 else (0, 0))
            f.write(struct.pack("<2q", rows, cols))
            f.write(np.ascontiguousarray(m, dtype=np.uint8).tobytes())
        t = np.ascontiguousarray(p.value_table, dtype=np.float64)
        f.write(struct.pack("<2q", t.shape[0], t.shape[1]))
        f.write(t.tobytes())


def problem_digest(p: Problem) -> str:
    """SHA-256 over every input array of a problem: fixtures that are regenerated from a seed record it, so that a drift of
    the generator (or of the value-table builder) fails loudly instead of comparing against outputs of other inputs."""
    import hashlib
    h = hashlib.sha256()
    h.update(f"{p.method}|{p.n_cases}|{p.n_ctrls}|{p.path_length}|{p.top_k}|{p.iterations}".encode())
    for name, _ in _LEVELS:
        u = p.levels.uids[name]
        for arr, dt in ((u.src, np.int32), (u.trg, np.int32), (u.count, np.int32), (u.location, np.int64), (u.signs, np.int32)):
            h.update(np.ascontiguousarray(arr, dtype=dt).tobytes())
    for name in ("1a", "1b", "2", "3"):
        h.update(np.ascontiguousarray(p.levels.data_inds[name], dtype=np.int32).tobytes())
    for m in (p.data1, p.data2, p.perm_cases):
        h.update(np.ascontiguousarray(m, dtype=np.uint8).tobytes())
    h.update(np.ascontiguousarray(p.value_table, dtype=np.float64).tobytes())
    return h.hexdigest()
