"""Host-side mirror of the reference's join interface on top of libgcre_hip.so (ctypes).

``JoinExec`` / ``PathSet`` follow the reference classes of the same name (src/gcre.h:103-180,
src/gcre_paths.h:10-98); ``process_paths`` follows ``ProcessPaths`` (src/wrapper.cpp:177-281).  All
compute happens in the HIP library; there is no CPU path here -- if the library or a gfx950 device is
missing, construction raises.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from . import build as _build

_LIB = None

GCRE_OK, GCRE_ERR_ASSERT, GCRE_ERR_RANGE, GCRE_ERR_DEVICE, GCRE_ERR_ARG = 0, -1, -2, -3, -4
EXPECTED_ABI = 4   # GCRE_ABI_VERSION of include/gcre_hip.h this mirror was written against


class GcreError(RuntimeError):
    pass


class gcre_result(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("scores", ctypes.POINTER(ctypes.c_double)),
                ("src", ctypes.POINTER(ctypes.c_int32)), ("trg", ctypes.POINTER(ctypes.c_int32)),
                ("cases", ctypes.POINTER(ctypes.c_int32)), ("ctrls", ctypes.POINTER(ctypes.c_int32)),
                ("n_perm", ctypes.c_int32), ("null_max", ctypes.POINTER(ctypes.c_float))]


class gcre_join_opts(ctypes.Structure):
    _fields_ = [("sharded", ctypes.c_int32), ("keep_ranged", ctypes.c_int32), ("shard_begin", ctypes.c_int64),
                ("shard_end", ctypes.c_int64), ("d_null_out", ctypes.c_void_p), ("keep_begin", ctypes.c_int64),
                ("keep_end", ctypes.c_int64), ("exchanges", ctypes.c_int32), ("exchange", ctypes.c_void_p),
                ("exchange_user", ctypes.c_void_p)]


EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32)


class gcre_profile(ctypes.Structure):
    _fields_ = [("null_kernel_ms", ctypes.c_double), ("null_kernel_launches", ctypes.c_int64),
                ("stats_kernel_ms", ctypes.c_double), ("select_ms", ctypes.c_double), ("total_ms", ctypes.c_double),
                ("paths", ctypes.c_int64), ("scores", ctypes.c_int64), ("null_alg_bytes", ctypes.c_double),
                ("null_row_loads", ctypes.c_double), ("ie_launches", ctypes.c_int64),
                ("ie_overlap_lists", ctypes.c_int64), ("ie_hinted_joins", ctypes.c_int64),
                ("ie_plane_joins", ctypes.c_int64), ("ie_lookup_tiles", ctypes.c_int64), ("prepare_ms", ctypes.c_double), ("inspect_ms", ctypes.c_double),
                ("ie_quad_launches", ctypes.c_int64), ("inspect_replays", ctypes.c_int64)]


class gcre_level(ctypes.Structure):
    _fields_ = [("uid_count", ctypes.c_void_p), ("uid_location", ctypes.c_void_p), ("n_uids", ctypes.c_int64),
                ("signs", ctypes.c_void_p), ("n_signs", ctypes.c_int64)]


class gcre_level_table(ctypes.Structure):
    _fields_ = [("path_length", ctypes.c_int32), ("n_uids", ctypes.c_int64),
                ("src", ctypes.POINTER(ctypes.c_int32)), ("trg", ctypes.POINTER(ctypes.c_int32)),
                ("count", ctypes.POINTER(ctypes.c_int32)), ("location", ctypes.POINTER(ctypes.c_int64)),
                ("n_signs", ctypes.c_int64), ("signs", ctypes.POINTER(ctypes.c_int32)), ("total_paths", ctypes.c_int64)]


class gcre_levels(ctypes.Structure):
    _fields_ = [("level", gcre_level_table * 6), ("n_data_inds", ctypes.c_int64 * 4),
                ("data_inds", ctypes.POINTER(ctypes.c_int32) * 4), ("n_rels3", ctypes.c_int64),
                ("r3_src", ctypes.POINTER(ctypes.c_int32)), ("r3_trg", ctypes.POINTER(ctypes.c_int32)),
                ("r3_sign", ctypes.POINTER(ctypes.c_int32)), ("r3_trg2", ctypes.POINTER(ctypes.c_int32)),
                ("r3_sign2", ctypes.POINTER(ctypes.c_int32))]


class gcre_pp_input(ctypes.Structure):
    _fields_ = [("level", gcre_level * 6), ("data_inds", ctypes.c_void_p * 4), ("n_data_inds", ctypes.c_int64 * 4),
                ("data1", ctypes.c_void_p), ("data1_rows", ctypes.c_int64),
                ("data2", ctypes.c_void_p), ("data2_rows", ctypes.c_int64), ("data_col_major", ctypes.c_int),
                ("value_table", ctypes.c_void_p), ("vt_rows", ctypes.c_int), ("vt_cols", ctypes.c_int),
                ("vt_col_major", ctypes.c_int),
                ("perm_cases", ctypes.c_void_p), ("perm_rows", ctypes.c_int), ("perm_col_major", ctypes.c_int),
                ("path_length", ctypes.c_int), ("shard_rank", ctypes.c_int), ("shard_world", ctypes.c_int),
                ("window_perms", ctypes.c_int)]


# every symbol include/gcre_hip.h declares; tests check that the library exports all of them
EXPORTS = [
    "gcre_create", "gcre_destroy", "gcre_last_error", "gcre_abi_version", "gcre_set_top_k", "gcre_width_ul",
    "gcre_vlen", "gcre_set_value_table", "gcre_set_perm_cases", "gcre_set_perm_masks", "gcre_pathset_zeros",
    "gcre_pathset_from_dense", "gcre_pathset_from_words", "gcre_pathset_select", "gcre_pathset_size",
    "gcre_pathset_read", "gcre_pathset_free", "gcre_join", "gcre_result_free", "gcre_uids_create",
    "gcre_uids_total_paths", "gcre_uids_free", "gcre_join_uids", "gcre_get_profile",
    "gcre_process_paths", "gcre_resolve_count_locs", "gcre_build_levels", "gcre_levels_free", "gcre_values_table", "gcre_values_table_exact_order",
    "gcre_generate_perm_masks", "gcre_mix64", "gcre_get_perm_mask", "gcre_uids_set_reduced",
    "gcre_set_perm_window", "gcre_plan_perm_window", "gcre_process_paths_devices",
    "gcre_set_inspect_cache", "gcre_drop_inspections", "gcre_build_flags", "gcre_device_count",
    "gcre_rccl_selftest", "gcre_rccl_collectives", "gcre_join_ahead",
]


def lib_path() -> str:
    return _build.LIB


def load_library():
    """dlopen libgcre_hip.so (building it first if the tree is newer).  Fails loudly when it cannot."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # GCRE_LIB: a variant library built by tools/build_variant.py for an A/B measurement (never set by the product)
    path = os.environ.get("GCRE_LIB") or _build.build()
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:   # no silent fallback: the product IS this library
        raise GcreError(f"cannot load {path}: {e}") from e
    V, I, I64, P = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    lib.gcre_create.restype = V
    lib.gcre_create.argtypes = [I, I, I, I, I]
    lib.gcre_destroy.argtypes = [V]
    lib.gcre_destroy.restype = None
    lib.gcre_last_error.restype = ctypes.c_char_p
    lib.gcre_last_error.argtypes = [V]
    lib.gcre_abi_version.restype = I
    lib.gcre_build_flags.restype = ctypes.c_char_p
    if os.environ.get("GCRE_LIB"):
        # a variant library skips build()'s staleness and diagnostics checks: make up for them here
        if not hasattr(lib, "gcre_abi_version") or lib.gcre_abi_version() != EXPECTED_ABI:
            raise GcreError(f"GCRE_LIB={path}: ABI {lib.gcre_abi_version() if hasattr(lib, 'gcre_abi_version') else '?'}, "
                            f"this tree expects {EXPECTED_ABI} (rebuild the variant: tools/build_variant.py)")
        flags = lib.gcre_build_flags().decode()
        diag = [f for f in flags.split() if f.startswith("-D") and (f[2:].split("=")[0] in _build.DIAG_DEFINES or "ZHACK" in f or
                                                                      (f[2:].startswith("GCRE_") and "_NO" in f[2:]))]
        if diag and os.environ.get("GCRE_ALLOW_DIAG_BUILD") != "1":
            raise GcreError(f"GCRE_LIB={path} is a diagnostics build ({' '.join(diag)}): its results are wrong by design; "
                            "set GCRE_ALLOW_DIAG_BUILD=1 for a timing experiment")
        import sys
        print(f"[gcre] GCRE_LIB: using variant library {path} (build flags: {flags or 'none'})", file=sys.stderr)
    lib.gcre_set_top_k.argtypes = [V, I]
    lib.gcre_width_ul.argtypes = [V]
    lib.gcre_vlen.argtypes = [V]
    lib.gcre_set_value_table.argtypes = [V, P, I, I, I]
    lib.gcre_set_perm_cases.argtypes = [V, P, I, I, I]
    lib.gcre_set_perm_masks.argtypes = [V, P, I]
    lib.gcre_pathset_zeros.restype = V
    lib.gcre_pathset_zeros.argtypes = [V, I64]
    lib.gcre_pathset_from_dense.restype = V
    lib.gcre_pathset_from_dense.argtypes = [V, P, I64, I, I]
    lib.gcre_pathset_from_words.restype = V
    lib.gcre_pathset_from_words.argtypes = [V, P, I64]
    lib.gcre_pathset_select.restype = V
    lib.gcre_pathset_select.argtypes = [V, V, P, I64]
    lib.gcre_pathset_size.restype = I64
    lib.gcre_pathset_size.argtypes = [V]
    lib.gcre_pathset_read.argtypes = [V, V, P]
    lib.gcre_pathset_free.argtypes = [V]
    lib.gcre_pathset_free.restype = None
    lib.gcre_join.argtypes = [V, I, P, P, I64, P, I64, V, V, V, ctypes.POINTER(gcre_join_opts),
                              ctypes.POINTER(gcre_result)]
    lib.gcre_result_free.argtypes = [ctypes.POINTER(gcre_result)]
    lib.gcre_result_free.restype = None
    lib.gcre_uids_create.restype = V
    lib.gcre_uids_create.argtypes = [V, I, P, P, I64, P, I64]
    lib.gcre_uids_total_paths.restype = I64
    lib.gcre_uids_total_paths.argtypes = [V]
    lib.gcre_uids_free.argtypes = [V]
    lib.gcre_uids_free.restype = None
    lib.gcre_join_uids.argtypes = [V, V, V, V, V, ctypes.POINTER(gcre_join_opts), ctypes.POINTER(gcre_result)]
    lib.gcre_join_ahead.argtypes = [V, V, V, V, V, ctypes.POINTER(gcre_join_opts)]
    lib.gcre_get_profile.argtypes = [V, ctypes.POINTER(gcre_profile)]
    lib.gcre_process_paths.argtypes = [V, ctypes.POINTER(gcre_pp_input), ctypes.POINTER(gcre_result)]
    lib.gcre_resolve_count_locs.argtypes = [P, I64, P, P, P, I64, P, P]
    lib.gcre_build_levels.argtypes = [ctypes.c_int32, P, P, P, I64, ctypes.POINTER(gcre_levels)]
    lib.gcre_levels_free.argtypes = [ctypes.POINTER(gcre_levels)]
    lib.gcre_levels_free.restype = None
    lib.gcre_values_table.argtypes = [I, I, P]
    lib.gcre_values_table_exact_order.argtypes = [I, I]
    lib.gcre_rccl_selftest.argtypes = [I, ctypes.c_char_p, ctypes.c_size_t]
    lib.gcre_rccl_collectives.restype = ctypes.c_int64
    lib.gcre_generate_perm_masks.argtypes = [V, ctypes.c_uint64, P, I]
    lib.gcre_mix64.restype = ctypes.c_uint64
    lib.gcre_mix64.argtypes = [ctypes.c_uint64]
    lib.gcre_get_perm_mask.argtypes = [V, I, P]
    lib.gcre_uids_set_reduced.argtypes = [V, V, P, I64]
    lib.gcre_set_perm_window.argtypes = [V, I, I]
    lib.gcre_plan_perm_window.argtypes = [V, P, I]
    lib.gcre_set_inspect_cache.argtypes = [V, I]
    lib.gcre_drop_inspections.argtypes = [V, I]
    _LIB = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@dataclass
class JoinResult:
    """joined_res (src/gcre_types.h:45-48) as arrays; what make_score_list turns into an R list (wrapper.cpp:142-174)."""

    scores: np.ndarray     # float64 ascending
    src: np.ndarray        # idx; R sees ids[,1] = src + 1
    trg: np.ndarray        # loc; R sees ids[,2] = trg + 1
    cases: np.ndarray
    ctrls: np.ndarray
    null: np.ndarray       # float32 [iterations] -- "TestScores"

    def pvalues(self) -> np.ndarray:
        """#(TestScores >= score) / length(TestScores), R/ProcessPaths.R:316 (f64 score vs f32 maxima)."""
        t = self.null.astype(np.float64)
        if len(t) == 0:
            return np.full(len(self.scores), np.nan)
        return np.array([(t >= s).sum() / len(t) for s in self.scores])

    def as_r_list(self) -> Dict[str, object]:
        """The named list make_score_list builds (wrapper.cpp:167-173)."""
        return {
            "scores": self.scores,
            "ids": np.stack([self.src + 1, self.trg + 1], axis=1) if len(self.src) else np.zeros((0, 2), np.int32),
            "TestScores": self.null.astype(np.float64),
            "cases": self.cases.astype(np.float64),
            "controls": self.ctrls.astype(np.float64),
            "debug": [f"[debug] {s}:{t} {c}/{d}" for s, t, c, d in
                      zip(self.src.tolist(), self.trg.tolist(), self.cases.tolist(), self.ctrls.tolist())],
        }


def _take_result(lib, r: gcre_result) -> JoinResult:
    n, k = r.n, r.n_perm
    out = JoinResult(
        np.ctypeslib.as_array(r.scores, (n,)).copy() if n > 0 else np.zeros(0),
        np.ctypeslib.as_array(r.src, (n,)).copy() if n > 0 else np.zeros(0, np.int32),
        np.ctypeslib.as_array(r.trg, (n,)).copy() if n > 0 else np.zeros(0, np.int32),
        np.ctypeslib.as_array(r.cases, (n,)).copy() if n > 0 else np.zeros(0, np.int32),
        np.ctypeslib.as_array(r.ctrls, (n,)).copy() if n > 0 else np.zeros(0, np.int32),
        np.ctypeslib.as_array(r.null_max, (k,)).copy() if k > 0 else np.zeros(0, np.float32),
    )
    lib.gcre_result_free(ctypes.byref(r))
    return out


class PathSet:
    """Device-resident PathSet (src/gcre_paths.h:10-98)."""

    def __init__(self, owner: "JoinExec", handle: int):
        if not handle:
            owner._raise()
        self._owner, self._h = owner, handle
        self.size = int(owner._lib.gcre_pathset_size(handle))
        self.vlen = owner.vlen

    def select(self, indices) -> "PathSet":
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        return PathSet(self._owner, self._owner._lib.gcre_pathset_select(self._owner._h, self._h, _ptr(idx), len(idx)))

    def to_numpy(self) -> np.ndarray:
        out = np.zeros((self.size, self.vlen), dtype=np.uint64)
        self._owner._check(self._owner._lib.gcre_pathset_read(self._owner._h, self._h, _ptr(out)))
        return out

    def free(self) -> None:
        h, self._h = self._h, None
        if h and self._owner._h:
            self._owner._lib.gcre_pathset_free(h)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceUids:
    """UidRelSet (src/gcre.h:49-90) resident on the device."""

    def __init__(self, owner: "JoinExec", uids):
        count = np.ascontiguousarray(uids.count, dtype=np.int32)
        location = np.ascontiguousarray(uids.location, dtype=np.int64)
        signs = np.ascontiguousarray(uids.signs, dtype=np.int32)
        self._owner = owner
        self.path_length = int(uids.path_length)
        self._h = owner._lib.gcre_uids_create(owner._h, self.path_length, _ptr(count), _ptr(location), len(count),
                                              _ptr(signs), len(signs))
        if not self._h:
            owner._raise()
        self.total_paths = int(owner._lib.gcre_uids_total_paths(self._h))
        self._reduced = None

    def set_reduced(self, reduced: Optional["PathSet"], index=None) -> None:
        """gcre_uids_set_reduced: paths0[idx] | paths1[loc] == paths0[idx] | reduced[index[loc]] (checked per join)."""
        if reduced is None:
            self._owner._check(self._owner._lib.gcre_uids_set_reduced(self._h, None, None, 0))
            self._reduced = None
            return
        idx = np.ascontiguousarray(np.asarray(index).astype(np.int64) & 0xFFFFFFFF, dtype=np.uint32)   # bit 31 = swap halves
        self._owner._check(self._owner._lib.gcre_uids_set_reduced(self._h, reduced._h, _ptr(idx), len(idx)))
        self._reduced = reduced     # keeps the operand alive

    def free(self) -> None:
        h, self._h = self._h, None
        if h and self._owner._h:
            self._owner._lib.gcre_uids_free(h)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class JoinExec:
    """JoinExec (src/gcre.h:103-180) on one MI355X.  ``method`` is "method1" | "method2" or 1 | 2."""

    def __init__(self, method, num_cases: int, num_ctrls: int, iters: int, device: int = 0):
        self._lib = load_library()
        if isinstance(method, str):
            method = 1 if method == "method1" else 2   # JoinExec::to_method, gcre.h:125-133
        self.method, self.num_cases, self.num_ctrls, self.iters = int(method), int(num_cases), int(num_ctrls), int(iters)
        self._h = self._lib.gcre_create(self.method, self.num_cases, self.num_ctrls, self.iters, int(device))
        if not self._h:
            msg = self._lib.gcre_last_error(None).decode()
            if msg.startswith("assertion"):
                raise ValueError(msg)
            raise GcreError(msg)
        self.width_ul = self._lib.gcre_width_ul(self._h)
        self.vlen = self._lib.gcre_vlen(self._h)
        self._top_k = 12
        self.nthreads = 0   # accepted for interface parity; the device schedules the work

    # -- plumbing --
    def _raise(self, rc: int = GCRE_ERR_DEVICE):
        msg = self._lib.gcre_last_error(self._h).decode()
        if rc == GCRE_ERR_RANGE or "out of range" in msg:
            raise IndexError(msg)          # std::out_of_range
        if rc == GCRE_ERR_ASSERT or msg.startswith("assertion"):
            raise ValueError(msg)          # std::logic_error
        raise GcreError(msg)

    def _check(self, rc: int):
        if rc != GCRE_OK:
            self._raise(rc)

    def close(self):
        h, self._h = self._h, None
        if h:
            self._lib.gcre_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def top_k(self) -> int:
        return self._top_k

    @top_k.setter
    def top_k(self, k: int):
        self._check(self._lib.gcre_set_top_k(self._h, int(k)))
        self._top_k = int(k)

    # -- JoinExec interface --
    def set_value_table(self, table) -> None:
        t = np.ascontiguousarray(table, dtype=np.float64)
        self._check(self._lib.gcre_set_value_table(self._h, _ptr(t), t.shape[0], t.shape[1], 0))

    def set_permuted_cases(self, perms) -> None:
        p = np.ascontiguousarray(perms, dtype=np.int32)
        if p.ndim != 2:
            p = p.reshape(0, 0)
        self._check(self._lib.gcre_set_perm_cases(self._h, _ptr(p), p.shape[0], p.shape[1], 0))

    def set_permuted_masks(self, masks) -> None:
        m = np.ascontiguousarray(masks, dtype=np.uint64).reshape(-1, self.width_ul)
        self._check(self._lib.gcre_set_perm_masks(self._h, _ptr(m), m.shape[0]))

    def generate_permutations(self, seed: int, strata=None) -> None:
        """Device-side getRandIndicesMat + getCaseORControl + setPermutedCases (R/Utils.R:22-46, 246-262)."""
        if strata is None:
            self._check(self._lib.gcre_generate_perm_masks(self._h, int(seed) & (2**64 - 1), None, 0))
        else:
            st = np.ascontiguousarray(strata, dtype=np.int32)
            self._check(self._lib.gcre_generate_perm_masks(self._h, int(seed) & (2**64 - 1), _ptr(st), int(st.max()) + 1))

    def set_perm_window(self, k0: int, k1: int) -> None:
        """Joins that follow score permutations [k0, k1) only (tile aligned); see gcre_set_perm_window."""
        self._check(self._lib.gcre_set_perm_window(self._h, int(k0), int(k1)))

    def set_inspect_cache(self, on: bool) -> None:
        """Keep every join's mask-independent (inspector) output with its join index; see gcre_set_inspect_cache."""
        self._check(self._lib.gcre_set_inspect_cache(self._h, 1 if on else 0))

    def drop_inspections(self, release_memory: bool = False) -> None:
        """Forget the cached inspector outputs (the next join on every index recomputes them)."""
        self._check(self._lib.gcre_drop_inspections(self._h, 1 if release_memory else 0))

    def plan_perm_window(self, set_rows) -> int:
        """Permutations per window so that the count planes of path sets with ``set_rows`` rows fit in device memory."""
        rows = np.ascontiguousarray(np.atleast_1d(set_rows), dtype=np.int64)
        w = self._lib.gcre_plan_perm_window(self._h, _ptr(rows), len(rows))
        if w < 0:
            self._check(w)
        return max(int(w), 1)

    def perm_mask(self, r: int) -> np.ndarray:
        out = np.zeros(self.width_ul, dtype=np.uint64)
        self._check(self._lib.gcre_get_perm_mask(self._h, int(r), _ptr(out)))
        return out

    def create_path_set(self, size: int) -> PathSet:
        return PathSet(self, self._lib.gcre_pathset_zeros(self._h, int(size)))

    def load(self, data) -> PathSet:
        d = np.ascontiguousarray(data, dtype=np.int32)
        ncol = d.shape[1] if d.ndim == 2 else 0
        return PathSet(self, self._lib.gcre_pathset_from_dense(self._h, _ptr(d), d.shape[0], ncol, 0))

    def from_words(self, rows) -> PathSet:
        r = np.ascontiguousarray(rows, dtype=np.uint64).reshape(-1, self.vlen)
        return PathSet(self, self._lib.gcre_pathset_from_words(self._h, _ptr(r), r.shape[0]))

    def join(self, uids, paths0: PathSet, paths1: PathSet, paths_res: Optional[PathSet] = None,
             shard: Optional[Tuple[int, int]] = None, d_null_out: int = 0,
             keep: Optional[Tuple[int, int]] = None, keep_mode: int = 1, exchange=None, exchanges: int = 0) -> JoinResult:
        """JoinExec::join (src/join_base.cpp:189-264).  ``paths_res`` receives the joined rows when given.
        ``uids`` is a UidRelSet (uploaded for this call) or a DeviceUids (already resident).  ``shard`` restricts
        scoring to a range of joined paths; ``keep`` restricts the rows written to ``paths_res`` to a range (plus
        the scored shard) -- the rows this device's shards of the later joins will read -- or, with ``keep_mode`` 2,
        only the rows that get count planes (all rows are still written).  ``exchange(k0, k1)`` is called ``exchanges``
        times during the join (gcre_join_opts.exchange): it MAX-all-reduces d_null_out[k0:k1] across the ranks in place."""
        opts = gcre_join_opts(0, 0, 0, 0, None, 0, 0, 0, None, None)
        cb = None
        if exchange is not None and exchanges > 0:
            if not d_null_out:
                raise GcreError("exchange needs d_null_out")

            def _cb(_user, _d_null, k0, k1):
                try:
                    exchange(int(k0), int(k1))
                    return 0
                except Exception:      # an exception must not unwind through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = EXCHANGE_FN(_cb)      # kept alive until the join returns
            opts.exchanges = int(exchanges)
            opts.exchange = ctypes.cast(cb, ctypes.c_void_p)
        if shard is not None:
            opts.sharded, opts.shard_begin, opts.shard_end = 1, int(shard[0]), int(shard[1])
        if keep is not None:
            opts.keep_ranged, opts.keep_begin, opts.keep_end = int(keep_mode), int(keep[0]), int(keep[1])
        if d_null_out:
            opts.d_null_out = ctypes.c_void_p(int(d_null_out))
        res = gcre_result()
        res_h = paths_res._h if paths_res is not None else None
        if isinstance(uids, DeviceUids):
            rc = self._lib.gcre_join_uids(self._h, uids._h, paths0._h, paths1._h, res_h, ctypes.byref(opts),
                                          ctypes.byref(res))
        else:
            count = np.ascontiguousarray(uids.count, dtype=np.int32)
            location = np.ascontiguousarray(uids.location, dtype=np.int64)
            signs = np.ascontiguousarray(uids.signs, dtype=np.int32)
            rc = self._lib.gcre_join(self._h, int(uids.path_length), _ptr(count), _ptr(location), len(count),
                                     _ptr(signs), len(signs), paths0._h, paths1._h, res_h, ctypes.byref(opts),
                                     ctypes.byref(res))
        self._check(rc)
        return _take_result(self._lib, res)

    def join_ahead(self, uids: Optional["DeviceUids"], paths0: Optional[PathSet] = None, paths1: Optional[PathSet] = None,
                   paths_res: Optional[PathSet] = None, shard: Optional[Tuple[int, int]] = None,
                   keep: Optional[Tuple[int, int]] = None, keep_mode: int = 1) -> None:
        """gcre_join_ahead: register a LATER join of a sequence (call it once per later join, in order; ``None`` cancels).
        The join call that follows inspects and launches every registered join in turn once its own work is queued -- the
        inspectors on a stream of their own, beside the permutation kernels of the joins before them -- and the registered
        joins, called afterwards with the same arguments, only collect their results.  Needs the inspection cache
        (``set_inspect_cache(True)``) and GCRE_AHEAD=1."""
        if uids is None:
            self._check(self._lib.gcre_join_ahead(self._h, None, None, None, None, None))
            return
        opts = gcre_join_opts(0, 0, 0, 0, None, 0, 0, 0, None, None)
        if shard is not None:
            opts.sharded, opts.shard_begin, opts.shard_end = 1, int(shard[0]), int(shard[1])
        if keep is not None:
            opts.keep_ranged, opts.keep_begin, opts.keep_end = int(keep_mode), int(keep[0]), int(keep[1])
        res_h = paths_res._h if paths_res is not None else None
        self._check(self._lib.gcre_join_ahead(self._h, uids._h, paths0._h, paths1._h, res_h, ctypes.byref(opts)))

    def profile(self) -> Dict[str, float]:
        p = gcre_profile()
        self._check(self._lib.gcre_get_profile(self._h, ctypes.byref(p)))
        return {f: getattr(p, f) for f, _ in gcre_profile._fields_}


def resolve_count_locs(trg_uids, keys, counts, locations):
    """assemble_uids' lookup (src/wrapper.cpp:106-132) through the C ABI."""
    lib = load_library()
    trg = np.ascontiguousarray(trg_uids, dtype=np.int32)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    locations = np.ascontiguousarray(locations, dtype=np.int32)
    oc = np.zeros(len(trg), dtype=np.int32)
    ol = np.zeros(len(trg), dtype=np.int64)
    rc = lib.gcre_resolve_count_locs(_ptr(trg), len(trg), _ptr(keys), _ptr(counts), _ptr(locations), len(keys),
                                     _ptr(oc), _ptr(ol))
    if rc != GCRE_OK:
        raise GcreError(f"gcre_resolve_count_locs failed: {rc}")
    return oc, ol


def build_levels(n_genes: int, src, trg, sign):
    """Native build of the per-level join tables (R/ProcessPaths.R:206-256) -> geneticscre_amd.uids.LevelTables."""
    from .uids import LevelTables, UidRelSet
    lib = load_library()
    s = np.ascontiguousarray(src, dtype=np.int32)
    t = np.ascontiguousarray(trg, dtype=np.int32)
    g = np.ascontiguousarray(sign, dtype=np.int32)
    out = gcre_levels()
    rc = lib.gcre_build_levels(int(n_genes), _ptr(s), _ptr(t), _ptr(g), len(s), ctypes.byref(out))
    if rc == GCRE_ERR_RANGE:
        raise IndexError("relation endpoint outside 0..n_genes-1")
    if rc != GCRE_OK:
        raise ValueError("relations must be sorted by (src, trg), unique, without self loops")

    def arr(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, (n,)).astype(dt, copy=True) if n > 0 else np.zeros(0, dt)

    names = ["1a", "1b", "2", "3", "4", "5"]
    uids, n_paths = {}, {}
    for i, name in enumerate(names):
        lt = out.level[i]
        uids[name] = UidRelSet(lt.path_length, arr(lt.src, lt.n_uids, np.int32), arr(lt.trg, lt.n_uids, np.int32),
                               arr(lt.count, lt.n_uids, np.int32), arr(lt.location, lt.n_uids, np.int64),
                               arr(lt.signs, lt.n_signs, np.int32))
        n_paths[name] = int(lt.total_paths)
    data_inds = {name: arr(out.data_inds[i], out.n_data_inds[i], np.int32) for i, name in enumerate(names[:4])}
    n3 = out.n_rels3
    rels3 = {"srcuid": arr(out.r3_src, n3, np.int32), "trguid": arr(out.r3_trg, n3, np.int32),
             "sign": arr(out.r3_sign, n3, np.int32), "trguid2": arr(out.r3_trg2, n3, np.int32),
             "sign2": arr(out.r3_sign2, n3, np.int32)}
    lib.gcre_levels_free(ctypes.byref(out))
    return LevelTables(uids, data_inds, rels3, n_paths)


def values_table(n_cases: int, n_ctrls: int) -> np.ndarray:
    """Native getValuesTable (R/Utils.R:137-159)."""
    out = np.zeros((n_cases + 1, n_ctrls + 1), dtype=np.float64)
    rc = load_library().gcre_values_table(int(n_cases), int(n_ctrls), _ptr(out))
    if rc != GCRE_OK:
        raise ValueError("bad table dimensions")
    return out


def rccl_selftest(device: int = 0) -> None:
    """One-rank RCCL communicator + MAX all-reduce through the library's own (dlopen'ed) RCCL binding; raises on failure."""
    buf = ctypes.create_string_buffer(512)
    if load_library().gcre_rccl_selftest(int(device), buf, 512) != GCRE_OK:
        raise GcreError("gcre_rccl_selftest: " + buf.value.decode())


def rccl_collectives() -> int:
    return int(load_library().gcre_rccl_collectives())


def values_table_exact_order(n_cases: int, n_ctrls: int) -> bool:
    """True: ``values_table`` sums every cell in R's index order at this size; False: the sorted prefix sum of very large
    cohorts (the last bit of a cell in a hundred can differ).  Recorded with fixtures and bench lines."""
    return bool(load_library().gcre_values_table_exact_order(int(n_cases), int(n_ctrls)))


def _pp_input(problem, keep):
    """gcre_pp_input for a synth.Problem; numpy buffers are appended to ``keep`` (they must outlive the call)."""
    def arr(a, dt):
        b = np.ascontiguousarray(a, dtype=dt)
        keep.append(b)
        return b

    inp = gcre_pp_input()
    for i, name in enumerate(["1a", "1b", "2", "3", "4", "5"]):
        u = problem.levels.uids[name]
        c, l, s = arr(u.count, np.int32), arr(u.location, np.int64), arr(u.signs, np.int32)
        inp.level[i] = gcre_level(_ptr(c), _ptr(l), len(c), _ptr(s), len(s))
    for i, name in enumerate(["1a", "1b", "2", "3"]):
        d = arr(problem.levels.data_inds[name], np.int32)
        inp.data_inds[i], inp.n_data_inds[i] = _ptr(d), len(d)
    d1, d2 = arr(problem.data1, np.int32), arr(problem.data2, np.int32)
    inp.data1, inp.data1_rows, inp.data2, inp.data2_rows, inp.data_col_major = _ptr(d1), d1.shape[0], _ptr(d2), d2.shape[0], 0
    vt = arr(problem.value_table, np.float64)
    inp.value_table, inp.vt_rows, inp.vt_cols, inp.vt_col_major = _ptr(vt), vt.shape[0], vt.shape[1], 0
    pc = arr(problem.perm_cases, np.int32)
    inp.perm_cases = _ptr(pc) if pc.size else None
    inp.perm_rows, inp.perm_col_major = (pc.shape[0] if pc.ndim == 2 else 0), 0
    inp.path_length = int(problem.path_length)
    return inp


def process_paths_devices(problem, devices=None) -> Dict[str, object]:
    """ProcessPaths on several GPUs of the node from this one process (gcre_process_paths_devices): one context and host
    thread per entry of ``devices`` (None: every visible device; an id may repeat), joined paths sharded, maxima and top-k
    tables merged.  Bit-identical to ``process_paths`` for any device list."""
    lib = load_library()
    lib.gcre_process_paths_devices.restype = ctypes.c_int
    lib.gcre_process_paths_devices.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_char_p, ctypes.c_size_t]
    keep = []
    inp = _pp_input(problem, keep)
    outs = (gcre_result * 5)()
    err = ctypes.create_string_buffer(512)
    dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
    rc = lib.gcre_process_paths_devices(1 if problem.method == "method1" else 2, int(problem.n_cases), int(problem.n_ctrls),
                                        int(problem.iterations), int(problem.top_k), _ptr(dev) if dev is not None else None,
                                        0 if dev is None else len(dev), ctypes.byref(inp), outs, err, len(err))
    if rc != GCRE_OK:
        msg = err.value.decode() or f"gcre error {rc}"
        if rc == GCRE_ERR_RANGE or "out of range" in msg:
            raise IndexError(msg)
        if rc == GCRE_ERR_ASSERT or "assertion" in msg:
            raise ValueError(msg)
        raise GcreError(msg)
    return {f"lst{i + 1}": (None if outs[i].n < 0 else _take_result(lib, outs[i])) for i in range(5)}


def process_paths(problem, device: int = 0, exec_: Optional[JoinExec] = None) -> Dict[str, object]:
    """ProcessPaths (src/wrapper.cpp:177-281) in one native call.  Returns {"lst1": JoinResult | None, ...}.

    ``problem`` carries the 39 arguments as arrays (geneticscre_amd.synth.Problem).
    """
    ex = exec_ or JoinExec(problem.method, problem.n_cases, problem.n_ctrls, problem.iterations, device)
    ex.top_k = problem.top_k
    lib = ex._lib
    keep = []   # keep numpy buffers alive across the call

    def arr(a, dt):
        b = np.ascontiguousarray(a, dtype=dt)
        keep.append(b)
        return b

    inp = gcre_pp_input()
    for i, name in enumerate(["1a", "1b", "2", "3", "4", "5"]):
        u = problem.levels.uids[name]
        c, l, s = arr(u.count, np.int32), arr(u.location, np.int64), arr(u.signs, np.int32)
        inp.level[i] = gcre_level(_ptr(c), _ptr(l), len(c), _ptr(s), len(s))
    for i, name in enumerate(["1a", "1b", "2", "3"]):
        d = arr(problem.levels.data_inds[name], np.int32)
        inp.data_inds[i], inp.n_data_inds[i] = _ptr(d), len(d)
    d1, d2 = arr(problem.data1, np.int32), arr(problem.data2, np.int32)
    inp.data1, inp.data1_rows, inp.data2, inp.data2_rows, inp.data_col_major = _ptr(d1), d1.shape[0], _ptr(d2), d2.shape[0], 0
    vt = arr(problem.value_table, np.float64)
    inp.value_table, inp.vt_rows, inp.vt_cols, inp.vt_col_major = _ptr(vt), vt.shape[0], vt.shape[1], 0
    pc = arr(problem.perm_cases, np.int32)
    inp.perm_cases = _ptr(pc) if pc.size else None
    inp.perm_rows, inp.perm_col_major = (pc.shape[0] if pc.ndim == 2 else 0), 0
    inp.path_length = int(problem.path_length)
    outs = (gcre_result * 5)()
    rc = lib.gcre_process_paths(ex._h, ctypes.byref(inp), outs)
    ex._check(rc)
    result: Dict[str, object] = {}
    for i in range(5):
        result[f"lst{i + 1}"] = None if outs[i].n < 0 else _take_result(lib, outs[i])
    result["profile"] = ex.profile()
    if exec_ is None:
        ex.close()
    return result


class ResidentPlan:
    """The ProcessPaths join sequence (src/wrapper.cpp:216-276) with every input already in HBM.

    ``prepare`` uploads data, table, masks and the per-level join indices once; ``run`` then performs the
    joins of one pass: levels 1a, 1b, 2 .. path_length.  With ``world > 1`` each rank scores a contiguous
    slice of every level's joined paths (kept rows are materialised in full on every rank) and the caller
    merges null maxima (MAX) and top-k tables across ranks -- see bench.py.
    """

    LEVELS = ["1a", "1b", "2", "3", "4", "5"]

    def __init__(self, problem, device: int = 0, packed_masks: Optional[np.ndarray] = None,
                 mask_seed: Optional[int] = None):
        self.problem = problem
        self._needed: Dict[tuple, Optional[Tuple[int, int]]] = {}
        self._first: Dict[str, np.ndarray] = {}
        self._window: Optional[int] = None
        self._cache_on = False
        ex = self.ex = JoinExec(problem.method, problem.n_cases, problem.n_ctrls, problem.iterations, device)
        ex.top_k = problem.top_k
        ex.set_value_table(problem.value_table)
        if mask_seed is not None:
            ex.generate_permutations(mask_seed)      # drawn on the device: every rank with the same seed gets the same masks
        elif packed_masks is not None:
            ex.set_permuted_masks(packed_masks)
        else:
            ex.set_permuted_cases(problem.perm_cases)
        lv, L = problem.levels, problem.path_length
        self.names = ["1a", "1b"] + [str(l) for l in range(2, L + 1)]
        self.uids = {k: DeviceUids(ex, lv.uids[k]) for k in self.names}
        parsed1 = ex.load(problem.data1)
        parsed2 = ex.load(problem.data2)
        self.inputs = {"1a": parsed1.select(lv.data_inds["1a"]), "1b": parsed2.select(lv.data_inds["1b"])}
        self.zeros = {"1a": ex.create_path_set(len(lv.data_inds["1a"])), "1b": ex.create_path_set(len(lv.data_inds["1b"]))}
        if L >= 2:
            self.inputs["2"] = parsed1.select(lv.data_inds["3"])   # wrapper.cpp:207 reads data_idx2 from r_data_inds3
        if L >= 3:
            self.inputs["3"] = parsed1.select(lv.data_inds["3"])
        self.parsed = (parsed1, parsed2)
        self.kept = {"1": ex.create_path_set(self.uids["1a"].total_paths)}
        if L >= 2:
            self.kept["2"] = ex.create_path_set(self.uids["2"].total_paths)
        if L >= 3:
            self.kept["3"] = ex.create_path_set(self.uids["3"].total_paths)
        # what every join really adds to paths0 (gcre_uids_set_reduced; the same hints gcre_process_paths attaches)
        self.uids["1a"].set_reduced(parsed1, lv.data_inds["1a"])
        self.uids["1b"].set_reduced(parsed2, lv.data_inds["1b"])
        signed = problem.method == "method2"
        rel_neg = (np.asarray(lv.uids["2"].signs) != 1) if signed else np.zeros(len(lv.uids["2"].signs), bool)
        for name in ("2", "3"):
            if name in self.uids:
                self.uids[name].set_reduced(parsed1, lv.data_inds["3"])
        self._reduced_args: Dict[str, tuple] = {}
        self._pivot: Dict[tuple, DeviceUids] = {}
        # off by default: measured -- the level-4 shard gains 3 %, keeping all of level 2's planes costs more
        self.pivot_shards = os.environ.get("GCRE_PIVOT_SHARDS", "0") == "1"
        # Inspect- and launch-ahead (gcre_join_ahead): the later joins of a pass are inspected and launched while the first one's
        # results are collected.  Worth 5-8 % on BASELINE configs[1] (per-join host gaps), 0.5 % on configs[2], nothing or less on
        # configs[3] (long joins: inspector and permutation kernel each fill the GPU); it keeps every chunk's inspection for the
        # length of the pass (~130 B per joined path).  Default: plans of up to 64 M joined paths and 10^12 scores per pass.
        # GCRE_AHEAD=1 / 0: always / never (the library honours the same variable).
        env_ahead = os.environ.get("GCRE_AHEAD", "")
        total_paths = sum(int(self.uids[k].total_paths) for k in self.names)
        self.ahead = env_ahead == "1" or (env_ahead == "" and total_paths <= 64_000_000 and
                                          total_paths * max(int(problem.iterations), 1) <= 10 ** 12)
        if "4" in self.uids:    # level 2 put the added gene into the (-) half of paths2[loc] when the relation is negative
            self._reduced_args["4"] = (parsed1, np.asarray(lv.data_inds["3"], np.int64) | (rel_neg.astype(np.int64) << 31))
            self.uids["4"].set_reduced(*self._reduced_args["4"])
        if "5" in self.uids:    # paths3[loc] = (c, d, e): the join adds paths2[(d, e)], swapped when (c, d) is negative
            u3 = lv.uids["3"]
            cnt = np.maximum(np.asarray(u3.count, dtype=np.int64), 0)
            start = np.repeat(np.asarray(u3.location, dtype=np.int64), cnt)
            within = np.arange(int(cnt.sum()), dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            first_neg = np.repeat((np.asarray(u3.signs)[:len(cnt)] != 1) if signed else np.zeros(len(cnt), bool), cnt)
            self._reduced_args["5"] = (self.kept["2"], (start + within) | (first_neg.astype(np.int64) << 31))
            self.uids["5"].set_reduced(*self._reduced_args["5"])

    def operands(self, name: str):
        """(paths0, paths1, paths_res) of one level, as in wrapper.cpp:227-276."""
        k = self.kept
        return {
            "1a": (self.zeros["1a"], self.inputs["1a"], k["1"]),
            "1b": (self.zeros["1b"], self.inputs["1b"], None),
            "2": (k["1"], self.inputs.get("2"), k.get("2")),
            "3": (k.get("2"), self.inputs.get("3"), k.get("3")),
            "4": (k.get("3"), k.get("2"), None),
            "5": (k.get("3"), k.get("3"), None),
        }[name]

    def _uids_of(self, name: str, b: int, e: int) -> Tuple[int, int]:
        """Rows of paths0 (uids) that own the joined paths [b, e) of level ``name``."""
        if e <= b:
            return (0, 0)
        if name not in self._first:
            u = self.problem.levels.uids[name]
            self._first[name] = np.concatenate([[0], np.cumsum(np.maximum(np.asarray(u.count, dtype=np.int64), 0))])
        first = self._first[name]
        return (int(np.searchsorted(first, b, side="right")) - 1, int(np.searchsorted(first, e - 1, side="right")))

    def pivot_sharded(self, name: str, world: int) -> bool:
        """``GCRE_PIVOT_SHARDS=1``: levels that keep nothing (4 and 5) are sharded by PIVOT GROUP when there are several ranks:
        all uids that join the same paths1 rows (same ``location``: the 3-paths ending in one gene) go to one rank, so the
        planes of those rows are fetched on one GPU only.  The other levels keep an ordinal range (their kept rows must be
        ranges).  Exact like any sharding; not the default because it does not pay."""
        return self.pivot_shards and world > 1 and name in ("4", "5") and name in self.uids

    def pivot_uids(self, name: str, rank: int, world: int) -> DeviceUids:
        """The join index of level ``name`` with ``count = 0`` outside this rank's pivot groups (the reference skips such
        uids, src/join_base.cpp:236): a shard by uid subset that needs nothing below the C ABI.  Groups are cut at equal
        cumulative path counts; a rank's joined paths stay in the level's order, so ties are cut as on one GPU."""
        key = (name, rank, world)
        if key not in self._pivot:
            u = self.problem.levels.uids[name]
            cnt = np.maximum(np.asarray(u.count, dtype=np.int64), 0)
            loc = np.asarray(u.location, dtype=np.int64)
            groups, inv = np.unique(np.where(cnt > 0, loc, -1), return_inverse=True)
            weight = np.bincount(inv, weights=cnt.astype(np.float64), minlength=len(groups))
            cum = np.concatenate([[0.0], np.cumsum(weight)])
            total = cum[-1]
            g_lo = int(np.searchsorted(cum, total * rank / world, side="left"))
            g_hi = int(np.searchsorted(cum, total * (rank + 1) / world, side="left")) if rank + 1 < world else len(groups)
            mine = (inv >= g_lo) & (inv < g_hi)
            from .uids import UidRelSet
            sub = UidRelSet(u.path_length, u.src, u.trg, np.where(mine, cnt, 0).astype(np.int32), u.location, u.signs)
            du = DeviceUids(self.ex, sub)
            du.set_reduced(*self._reduced_args[name])
            self._pivot[key] = du
        return self._pivot[key]

    def needed_rows(self, name: str, rank: int, world: int) -> Optional[Tuple[int, int]]:
        """What THIS rank needs of the set level ``name`` keeps, as a range of its rows (None: everything).
        Level 3's rows are only read as paths0 of the level-4 shard: the others are not produced at all
        (gcre_join_opts.keep_ranged = 1).  Levels 1 and 2 are also read as paths1 / reduced operands, so all their
        rows are written, but count planes are only needed for the rows this rank's work on the next level reads
        as paths0 (keep_ranged = 2, see ``keep_mode``).  With level 5 in the run every rank needs all of levels 2
        and 3 (level 5 joins level 3 with itself through level 2's planes)."""
        if world == 1 or "5" in self.uids or name not in ("1a", "2", "3"):
            return None
        if self.pivot_sharded("4", world) and name in ("2", "3"):
            # a rank's pivot groups read level-3 rows from everywhere, and their recipes start at arbitrary rows of level 2:
            # every rank keeps all of both (level 3 as recipes, level 2 with its planes)
            return None
        key = (name, rank, world)
        if key in self._needed:
            return self._needed[key]
        nxt = {"1a": "2", "2": "3", "3": "4"}[name]
        if nxt not in self.uids:
            self._needed[key] = None if name != "3" else (0, 0)
            return self._needed[key]
        # joined paths of the next level this rank works on: its shard, and whatever the level after needs of it
        b, e = self.shard(nxt, rank, world)
        more = self.needed_rows(nxt, rank, world) if nxt in ("2", "3") else (0, 0)
        if more is None:
            more = (0, self.uids[nxt].total_paths)
        if more[1] > more[0]:
            b, e = (min(b, more[0]), max(e, more[1])) if e > b else more
        self._needed[key] = self._uids_of(nxt, b, e)
        return self._needed[key]

    def keep_mode(self, name: str) -> int:
        return 1 if name == "3" else 2

    def shard(self, name: str, rank: int, world: int) -> Tuple[int, int]:
        total = self.uids[name].total_paths
        return (total * rank) // world, (total * (rank + 1)) // world

    def exchange_count(self, name: str, world: int) -> int:
        """How often a rank shares its running maxima with the others during the join of level ``name`` (the same on
        every rank: it only depends on the level's size): one exchange per doubling of the work beyond
        GCRE_EXCHANGE_UNIT joined-path x 2048-permutation tiles per rank (default 2 M, about 0.3 ms of the pruned kernel),
        at most 8.  Small joins do not exchange at all."""
        if world <= 1 or self.problem.iterations <= 0:
            return 0
        unit = float(os.environ.get("GCRE_EXCHANGE_UNIT", 2e6))
        window = self._window if self._window else self.problem.iterations
        tiles = -(-min(window, self.problem.iterations) // 2048)
        work = self.uids[name].total_paths / world * tiles
        if unit <= 0 or work < 2 * unit:
            return 0
        most = min(max(float(os.environ.get("GCRE_EXCHANGE_MAX", 8)), 1.0), 32.0)
        return int(min(most, np.floor(np.log2(work / unit))))

    def run(self, rank: int = 0, world: int = 1, d_null_out: int = 0, on_level=None,
            keep_inspections: bool = False, exchange=None) -> Dict[str, JoinResult]:
        """One pass over all levels.  Large permutation counts run in windows of whole 2048-permutation tiles (the count
        planes of the kept sets are per tile and have to fit in device memory): all levels for window 0, then all levels
        for window 1, ...  ``on_level(name, result, shard, window)`` sees every (level, window) result -- its null
        maxima are those of the window's permutations; ``d_null_out`` (device pointer to K floats) receives them at
        the window's offset.  The returned results carry the first window's top-k (they do not depend on the window)
        and the concatenated null maxima.

        A pass of several windows runs every join's inspector (expansion, observed scores, top-k, kept rows, lists) for the
        first window only: the library's inspection cache is on for the pass, and forgotten when the next pass starts --
        unless ``keep_inspections``: then a later pass over the same resident inputs starts every join at its null
        kernel (steady state of a service that re-scores the same network against new permutations).

        ``exchange(name, k0, k1)`` (multi-GPU): MAX-all-reduce ``d_null_out[k0:k1]`` across the ranks in place; called
        ``exchange_count(name, world)`` times during a level's join so that every rank prunes against the whole level's
        running maxima, not only its shard's (gcre_join_opts.exchange).  The null maxima a rank then returns include what
        it learned from the others; their MAX over the ranks is unchanged."""
        K = self.problem.iterations
        if self._window is None:
            self._window = self.planned_window()
        n_windows = len(range(0, max(K, 1), self._window))
        # inspect-ahead (gcre_join_ahead): every join's inspector runs beside the permutation kernel of the join before it,
        # into the inspection cache -- which therefore is on for the pass and, unless the caller keeps inspections, forgotten
        # when the pass ends (nothing of one pass serves the next)
        ahead = self.ahead and K > 0
        if keep_inspections or n_windows > 1 or ahead:
            self.ex.set_inspect_cache(True)
            if not keep_inspections:
                self.ex.drop_inspections()
        elif self._cache_on:
            self.ex.set_inspect_cache(False)
        self._cache_on = keep_inspections or n_windows > 1 or ahead
        out: Dict[str, JoinResult] = {}
        nulls: Dict[str, list] = {}
        prof: Dict[str, float] = {}
        for k0 in range(0, max(K, 1), self._window):
            k1 = min(K, k0 + self._window)
            if K > 0:
                self.ex.set_perm_window(k0, k1)
            def spec(name):
                p0, p1, res = self.operands(name)
                b, e = self.shard(name, rank, world)
                by_pivot = self.pivot_sharded(name, world)
                return (self.pivot_uids(name, rank, world) if by_pivot else self.uids[name], p0, p1, res,
                        (b, e) if (world > 1 and not by_pivot) else None, self.needed_rows(name, rank, world), self.keep_mode(name))

            for pos, name in enumerate(self.names):
                p0, p1, res = self.operands(name)
                b, e = self.shard(name, rank, world)
                n_ex = self.exchange_count(name, world) if (exchange is not None and d_null_out) else 0
                by_pivot = self.pivot_sharded(name, world)
                if ahead and pos == 0:
                    # the chain: every later level of this window, in order, up to the first one that exchanges thresholds
                    # with other ranks inside its join (that one, and what reads its rows, runs in its own call)
                    self.ex.join_ahead(None)
                    for later in self.names[1:]:
                        if exchange is not None and d_null_out and self.exchange_count(later, world) > 0:
                            break
                        u_n, p0_n, p1_n, res_n, shard_n, keep_n, mode_n = spec(later)
                        self.ex.join_ahead(u_n, p0_n, p1_n, res_n, shard=shard_n, keep=keep_n, keep_mode=mode_n)
                r = self.ex.join(self.pivot_uids(name, rank, world) if by_pivot else self.uids[name], p0, p1, res,
                                 shard=(b, e) if (world > 1 and not by_pivot) else None,
                                 d_null_out=(d_null_out + 4 * k0) if d_null_out else 0,
                                 keep=self.needed_rows(name, rank, world), keep_mode=self.keep_mode(name),
                                 exchange=(lambda a, b_, name=name: exchange(name, a, b_)) if n_ex else None, exchanges=n_ex)
                for k, v in self.ex.profile().items():
                    prof[k] = prof.get(k, 0) + v
                if on_level is not None:
                    r = on_level(name, r, (b, e), (k0, k1))
                nulls.setdefault(name, []).append(r.null)
                if name not in out:
                    out[name] = r
        if K > 0:
            self.ex.set_perm_window(0, K)
        if self._cache_on and not keep_inspections:
            self.ex.drop_inspections()     # this pass's inspections end with it (the buffers stay for the next pass)
        for name, r in out.items():
            r.null = np.concatenate(nulls[name]) if len(nulls[name]) > 1 else nulls[name][0]
        self.last_profile = prof
        return out

    def planned_window(self) -> int:
        """Permutations per window THIS device would choose (from its free memory, gcre_plan_perm_window).  Ranks of a
        multi-GPU job must agree on one value before the first pass -- every (level, window) is one round of collectives --
        see ``geneticscre_amd.dist.agree_window`` and ``set_window``."""
        if self.problem.iterations <= 0:
            return 1
        sets = [ps.size for ps in self.kept.values()] + [ps.size for ps in self.parsed]
        return self.ex.plan_perm_window(sets)

    def set_window(self, perms: int) -> None:
        """Fix the permutation window (a multiple of the 2048-permutation tile unless it covers everything)."""
        K = max(self.problem.iterations, 1)
        perms = int(perms)
        self._window = K if perms >= K else max(2048, perms // 2048 * 2048)

    def windows(self):
        """[(k0, k1), ...] the passes of ``run`` will walk."""
        K = self.problem.iterations
        w = self._window if self._window is not None else self.planned_window()
        return [(k0, min(K, k0 + w)) for k0 in range(0, max(K, 1), w)]

    def total_scores(self) -> int:
        return self.problem.iterations * sum(self.uids[k].total_paths for k in self.names)

    def close(self):
        self.ex.close()
