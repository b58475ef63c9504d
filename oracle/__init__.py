"""CPU oracle for the path-join scorer -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``geneticscre_amd``) never does.  It wraps ``oracle/gcre_oracle.cpp`` -- a CPU
restatement of the reference's ``JoinExec`` (src/join_base.cpp, src/methods.h) -- through ctypes.

Parity status: pinned by the SURVEY.md Appendix B known-answer vectors and by goldens generated from a partial
build of the reference's own scoring headers (oracle/ref_partial -> tests/golden/ref_cases); see gcre_oracle.h.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _cpu_tag() -> str:
    """-march=native code must not travel between hosts: key the .so on the host's CPU flags."""
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


def lib_path() -> str:
    return os.path.join(_HERE, f"libgcre_oracle.{_cpu_tag()}.so")


def build(force: bool = False) -> str:
    """Compile the oracle for THIS host (g++ -O3 -march=native, the reference's INSTALL:4-9 flags)."""
    out = lib_path()
    src = os.path.join(_HERE, "gcre_oracle.cpp")
    hdr = os.path.join(_HERE, "gcre_oracle.h")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return out
    cmd = ["g++", "-std=c++17", "-O3", "-march=native", "-mpopcnt", "-fPIC", "-pthread", "-shared",
           "-o", out + ".tmp", src]
    subprocess.run(cmd, check=True, cwd=_HERE)
    os.replace(out + ".tmp", out)
    return out


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    lib = ctypes.CDLL(build())
    c_i, c_i64 = ctypes.c_int, ctypes.c_int64
    P = ctypes.c_void_p
    lib.gcre_o_create.restype = P
    lib.gcre_o_create.argtypes = [c_i, c_i, c_i, c_i]
    lib.gcre_o_destroy.argtypes = [P]
    lib.gcre_o_width.argtypes = [P]
    lib.gcre_o_vlen.argtypes = [P]
    lib.gcre_o_set_value_table.argtypes = [P, P, c_i, c_i]
    lib.gcre_o_set_perm_cases.argtypes = [P, P, c_i, c_i]
    lib.gcre_o_set_perm_masks.argtypes = [P, P, c_i]
    lib.gcre_o_get_perm_mask.argtypes = [P, c_i, P]
    lib.gcre_o_pack_dense.argtypes = [P, P, c_i, c_i, P]
    lib.gcre_o_join.argtypes = [P, c_i, P, P, c_i64, P, c_i64, P, c_i64, P, c_i64, P,
                                c_i, c_i, c_i, P, P, P, P, P, P, P, P, P, P]
    _LIB = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@dataclass
class OracleResult:
    scores: np.ndarray      # float64, ascending
    src: np.ndarray         # idx (row of paths0 / uids)
    trg: np.ndarray         # loc (row of paths1)
    cases: np.ndarray
    ctrls: np.ndarray
    null: np.ndarray        # float32 [iters], the per-permutation maxima
    paths_res: Optional[np.ndarray]
    all_scores: np.ndarray  # float64 per joined path, path order
    all_cases: np.ndarray
    all_ctrls: np.ndarray


class OracleJoinExec:
    """Mirror of the reference's JoinExec (src/gcre.h:103-180) on top of the C++ restatement."""

    def __init__(self, method, num_cases: int, num_ctrls: int, iters: int):
        if isinstance(method, str):
            method = 1 if method == "method1" else 2      # JoinExec::to_method, gcre.h:125-133
        self.method = int(method)
        self.num_cases, self.num_ctrls, self.iters = int(num_cases), int(num_ctrls), int(iters)
        self.top_k = 12                                   # gcre.h:120
        self.nthreads = 0
        self._h = _lib().gcre_o_create(self.method, self.num_cases, self.num_ctrls, self.iters)
        if not self._h:
            raise ValueError("assertion")                 # check_true, join_base.cpp:47
        self.width = _lib().gcre_o_width(self._h)
        self.vlen = _lib().gcre_o_vlen(self._h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib().gcre_o_destroy(h)

    def set_value_table(self, table) -> None:
        t = np.ascontiguousarray(table, dtype=np.float64)
        assert t.ndim == 2
        _check(_lib().gcre_o_set_value_table(self._h, _ptr(t), t.shape[0], t.shape[1]))

    def set_permuted_cases(self, perms) -> None:
        p = np.ascontiguousarray(perms, dtype=np.int32)
        if p.ndim != 2:
            p = p.reshape(0, 0)
        _check(_lib().gcre_o_set_perm_cases(self._h, _ptr(p), p.shape[0], p.shape[1]))

    def set_packed_masks(self, masks) -> None:
        """Masks already packed (uint64 [rows][width]); same reuse / truncation rules as set_permuted_cases."""
        m = np.ascontiguousarray(masks, dtype=np.uint64).reshape(-1, self.width)
        _check(_lib().gcre_o_set_perm_masks(self._h, _ptr(m), m.shape[0]))

    def perm_mask(self, r: int) -> np.ndarray:
        out = np.zeros(self.width, dtype=np.uint64)
        _check(_lib().gcre_o_get_perm_mask(self._h, r, _ptr(out)))
        return out

    def create_path_set(self, size: int) -> np.ndarray:
        return np.zeros((int(size), self.vlen), dtype=np.uint64)

    def load(self, data) -> np.ndarray:
        d = np.ascontiguousarray(data, dtype=np.int32)
        out = np.zeros((d.shape[0], self.vlen), dtype=np.uint64)
        _check(_lib().gcre_o_pack_dense(self._h, _ptr(d), d.shape[0], d.shape[1] if d.ndim == 2 else 0, _ptr(out)))
        return out

    def join(self, uids, paths0: np.ndarray, paths1: np.ndarray, keep: bool = False,
             order: str = "reference") -> OracleResult:
        count = np.ascontiguousarray(uids.count, dtype=np.int32)
        location = np.ascontiguousarray(uids.location, dtype=np.int64)
        signs = np.ascontiguousarray(uids.signs, dtype=np.int32)
        p0 = np.ascontiguousarray(paths0, dtype=np.uint64).reshape(-1, self.vlen)
        p1 = np.ascontiguousarray(paths1, dtype=np.uint64).reshape(-1, self.vlen)
        total = int(np.maximum(count, 0).sum())
        res = np.zeros((total, self.vlen), dtype=np.uint64) if keep else None
        k = int(self.top_k)
        scores = np.zeros(k, dtype=np.float64)
        src, trg = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32)
        cases, ctrls = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32)
        n_out = ctypes.c_int(0)
        null = np.zeros(self.iters, dtype=np.float32)
        all_scores = np.zeros(total, dtype=np.float64)
        all_cases, all_ctrls = np.zeros(total, dtype=np.int32), np.zeros(total, dtype=np.int32)
        rc = _lib().gcre_o_join(self._h, int(uids.path_length), _ptr(count), _ptr(location), len(count),
                                _ptr(signs), len(signs), _ptr(p0), p0.shape[0], _ptr(p1), p1.shape[0], _ptr(res),
                                k, int(self.nthreads), 1 if order == "canonical" else 0,
                                _ptr(scores), _ptr(src), _ptr(trg), _ptr(cases), _ptr(ctrls), ctypes.byref(n_out),
                                _ptr(null), _ptr(all_scores), _ptr(all_cases), _ptr(all_ctrls))
        _check(rc)
        n = n_out.value
        return OracleResult(scores[:n], src[:n], trg[:n], cases[:n], ctrls[:n], null, res,
                            all_scores, all_cases, all_ctrls)


def _check(rc: int) -> None:
    if rc == -2:
        raise IndexError("assertion")     # std::out_of_range, gcre_types.h:68-76
    if rc != 0:
        raise ValueError("assertion")     # std::logic_error, gcre_types.h:58-66


def process_paths(problem, order: str = "reference", nthreads: int = 0, packed_masks=None) -> dict:
    """The six-join sequence of ProcessPaths (src/wrapper.cpp:216-276 == test/harness.cpp:112-181) on the oracle.

    ``problem`` is a ``geneticscre_amd.synth.Problem`` (duck-typed).  Returns {"lst1": OracleResult, ...} for
    the levels up to ``problem.path_length``, plus the kept path sets under "paths1".."paths3".
    """
    ex = OracleJoinExec(problem.method, problem.n_cases, problem.n_ctrls, problem.iterations)
    ex.top_k, ex.nthreads = problem.top_k, nthreads
    ex.set_value_table(problem.value_table)
    if packed_masks is not None:      # uint64 [K][ceil(n/64)] case masks instead of the K x n matrix (bench-sized problems)
        ex.set_packed_masks(packed_masks)
    else:
        ex.set_permuted_cases(problem.perm_cases)
    lv, out = problem.levels, {}
    parsed1 = ex.load(problem.data1)
    L = problem.path_length
    if L >= 1:
        idx1a, idx1b = lv.data_inds["1a"], lv.data_inds["1b"]
        r = ex.join(lv.uids["1a"], ex.create_path_set(len(idx1a)), parsed1[idx1a], keep=True, order=order)
        out["paths1"] = r.paths_res
        parsed2 = ex.load(problem.data2)
        out["lst1"] = ex.join(lv.uids["1b"], ex.create_path_set(len(idx1b)), parsed2[idx1b], keep=False, order=order)
    if L >= 2:
        r = ex.join(lv.uids["2"], out["paths1"], parsed1[lv.data_inds["2"]], keep=True, order=order)
        out["paths2"], out["lst2"] = r.paths_res, r
    if L >= 3:
        r = ex.join(lv.uids["3"], out["paths2"], parsed1[lv.data_inds["3"]], keep=True, order=order)
        out["paths3"], out["lst3"] = r.paths_res, r
    if L >= 4:
        out["lst4"] = ex.join(lv.uids["4"], out["paths3"], out["paths2"], keep=False, order=order)
    if L >= 5:
        out["lst5"] = ex.join(lv.uids["5"], out["paths3"], out["paths3"], keep=False, order=order)
    return out
