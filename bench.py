#!/usr/bin/env python3
"""bench.py -- path x permutation scores/sec of the MI355X path-join scorer.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config roofline]

One "step" = one pass of the hot path (the ProcessPaths join sequence, reference src/wrapper.cpp:225-276:
levels 1a, 1b, 2 .. path_length) over one synthetic problem whose inputs are already resident in HBM.
For N > 1 it is launched by torch.distributed.run, one rank per GPU: every level's joined paths are sharded
into N contiguous slices, each rank scores its slice, and the per-permutation null maxima (MAX all-reduce) and
the top-k tables (all-gather + merge) are exchanged over RCCL; inside a large join the ranks also share their running
maxima (the pruning thresholds) a few times.  Workload: BASELINE configs[2] on one GPU; on several GPUs configs[3], the
geometry BASELINE names for the 8-GPU run (same network, 10,000 patients, 100,000 permutations) -- `--config` overrides
either.  "strong" scaling: the workload does not depend on N (results identical for every N; the line also carries the
same workload timed on ONE of the GPUs, `one_gpu_same_workload`); `--scaling weak` runs the config's permutation count
PER GPU instead (K x N in total).

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GCRE_QUIET", "1")

# BASELINE.json configs; "roofline" (configs[2]) is the one the headline target is quoted on
CONFIGS = {
    "plumbing": dict(idx=0, genes=100, edges=300, cases=100, ctrls=100, perms=100, length=3, method="method1"),
    "subgraph": dict(idx=1, genes=5000, edges=60000, cases=500, ctrls=500, perms=1000, length=3, method="method1"),
    "roofline": dict(idx=2, genes=17000, edges=200000, cases=2500, ctrls=2500, perms=10000, length=4, method="method1"),
    "sharded": dict(idx=3, genes=17000, edges=200000, cases=5000, ctrls=5000, perms=100000, length=4, method="method1"),
    "signed": dict(idx=4, genes=17000, edges=60000, cases=25000, ctrls=25000, perms=100000, length=5, method="method2"),
}
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
VALU_PEAK_OPS = 256 * 4 * 32 * 2.4e9        # 256 CUs x 4 SIMD-32 x 2.4 GHz lane-ops/s (32-bit integer VALU)
# wave64 VALU instructions per second the chip issues: a SIMD-32 takes 2 clocks per wave64 instruction (one wave alone: 4).
# Measured with tools/valu_rate.hip (profiles/r02_valu_rate.txt): v_and+v_xor 117.8, v_bitop3 full adders 105-106 of 128
# lane-ops/clk/CU with >= 2 waves per SIMD -- 0.92 / 0.83 of the nominal 256 x 4 x 2.4e9 / 2
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2.0
ROW_LOAD_PEAK_G = 42.0                      # G wave-loads/s of random 256-B L2-resident rows (tools/row_gather_rate.hip)


def fast_table(n_cases, n_ctrls):
    """A cheap stand-in for the -log hypergeometric table at sizes where building the real one takes minutes
    (tests that compare two kernels with each other, where the values do not matter but the valley shape does):
    half the chi-square statistic of the 2 x 2 split."""
    n = float(n_cases + n_ctrls)
    out = np.empty((n_cases + 1, n_ctrls + 1), dtype=np.float64)
    j = np.arange(n_ctrls + 1, dtype=np.float64)[None, :]
    step = max(1, (1 << 21) // (n_ctrls + 1))          # ~16 MB of rows at a time: stays in cache, no 5 GB temporaries
    for r0 in range(0, n_cases + 1, step):
        i = np.arange(r0, min(r0 + step, n_cases + 1), dtype=np.float64)[:, None]
        tot = i + j
        den = tot * (n - tot)
        np.multiply(den, 2.0 * n_cases * n_ctrls / n, out=den)
        num = i * n_ctrls - j * n_cases
        np.multiply(num, num, out=num)
        with np.errstate(divide="ignore", invalid="ignore"):
            np.divide(num, den, out=num)
        num[~np.isfinite(num)] = 0.0
        out[r0:r0 + i.shape[0]] = num
    return out


def build_inputs(cfg, seed, top_k, table_fn=None):
    from geneticscre_amd import synth
    rng = np.random.default_rng(seed)
    g, src, trg, sign = synth.signed_network(cfg["genes"], cfg["edges"], rng)
    levels = synth.build_level_tables(g, src, trg, sign)
    n = cfg["cases"] + cfg["ctrls"]
    data1 = synth.variant_matrix(g, n, rng, fixed_rate=float(cfg.get("carrier_rate", 0.0)))
    data2 = data1[levels.uids["1b"].src]
    big = n * cfg["perms"] > 2_000_000_000 or n > 20000 or cfg["perms"] > 20000
    if big:
        # configs[3]/[4] scale: the numpy table builder is O(n m^2) and the host mask generator minutes -- use the native
        # table builder (gcre_values_table, no GPU needed) and let the device draw the masks (gcre_generate_perm_masks)
        from geneticscre_amd import api
        table = (table_fn or api.values_table)(cfg["cases"], cfg["ctrls"])
        masks = None
    else:
        table = (table_fn or synth.values_table)(cfg["cases"], cfg["ctrls"])
        masks = synth.packed_case_masks(cfg["cases"], cfg["ctrls"], cfg["perms"], rng)
    prob = synth.Problem(cfg["method"], cfg["cases"], cfg["ctrls"], cfg["length"], top_k, cfg["perms"], levels,
                         data1, data2, table, np.zeros((0, 0), np.int32), seed)
    return prob, masks


def host_threads() -> int:
    """CPU threads this process may actually use: the cgroup quota on the GPU box (16 for a 1-GPU share),
    the affinity mask, or the core count -- whichever is smallest."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def host_cpu():
    """CPU model of this host and whether it has AVX-512 VPOPCNTDQ (what the reference's popcount loop vectorises to under
    -O3 -march=native, reference INSTALL:8 / src/methods.h:81-82)."""
    model, flags = "unknown", set()
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("flags") and not flags:
                flags = set(ln.split(":", 1)[1].split())
    except OSError:
        pass
    return {"model": model, "avx512_vpopcntdq": "avx512_vpopcntdq" in flags, "avx512f": "avx512f" in flags, "avx2": "avx2" in flags}


def binary_has(path, mnemonic) -> bool:
    """Does the compiled reference code contain this instruction (objdump of the binary that is timed)?"""
    import shutil
    import subprocess
    for tool in ("objdump", "/opt/rocm/lib/llvm/bin/llvm-objdump"):
        exe = shutil.which(tool) or (tool if os.path.exists(tool) else None)
        if not exe:
            continue
        try:
            out = subprocess.run([exe, "-d", path], capture_output=True, text=True, timeout=60).stdout
            return mnemonic in out
        except (OSError, subprocess.SubprocessError):
            continue
    return False


def cpu_baseline(prob, masks, budget_s=15.0):
    """The CPU oracle (oracle/, a port of the reference's JoinExec) timed on a bounded sample of the same
    workload: a prefix of the deepest level's join index, all K permutations, all host cores."""
    import oracle
    from geneticscre_amd.uids import UidRelSet
    threads = host_threads()
    ex = oracle.OracleJoinExec(prob.method, prob.n_cases, prob.n_ctrls, prob.iterations)
    ex.top_k, ex.nthreads = prob.top_k, threads
    ex.set_value_table(prob.value_table)
    ex.set_packed_masks(masks)
    lv = prob.levels
    # operands of the deepest level need the kept path sets: build them with zero permutations (cheap)
    ex0 = oracle.OracleJoinExec(prob.method, prob.n_cases, prob.n_ctrls, 0)
    ex0.top_k, ex0.nthreads = prob.top_k, threads
    ex0.set_value_table(prob.value_table)
    ex0.set_permuted_cases(np.zeros((0, 0), np.int32))
    parsed1 = ex0.load(prob.data1)
    paths1 = ex0.join(lv.uids["1a"], ex0.create_path_set(len(lv.data_inds["1a"])), parsed1[lv.data_inds["1a"]], keep=True).paths_res
    L = prob.path_length
    name = str(L) if L >= 2 else "1b"
    if L >= 2:
        paths2 = ex0.join(lv.uids["2"], paths1, parsed1[lv.data_inds["2"]], keep=True).paths_res
    if L >= 3:
        paths3 = ex0.join(lv.uids["3"], paths2, parsed1[lv.data_inds["3"]], keep=True).paths_res
    ops = {"1b": None, "2": (paths1, parsed1[lv.data_inds["2"]]) if L >= 2 else None,
           "3": (paths2, parsed1[lv.data_inds["3"]]) if L >= 3 else None,
           "4": (paths3, paths2) if L >= 4 else None, "5": (paths3, paths3) if L >= 5 else None}
    if ops[name] is None:
        return None
    p0, p1 = ops[name]
    u = lv.uids[name]

    def run(n_uids):
        sub = UidRelSet(u.path_length, u.src[:n_uids], u.trg[:n_uids], u.count[:n_uids], u.location[:n_uids], u.signs)
        t0 = time.perf_counter()
        res = ex.join(sub, p0[:n_uids], p1, keep=False)
        return time.perf_counter() - t0, sub.count_total_paths(), res

    n_try = max(1, min(len(u), 64 * threads))
    t, paths = run(n_try)[:2]
    rate = paths * prob.iterations / max(t, 1e-9)
    want_paths = rate * budget_s / max(prob.iterations, 1)
    n_uids = int(min(len(u), max(n_try, np.searchsorted(u.path_idx[1:], want_paths) + 1)))
    t, paths, port_res = run(n_uids)
    port = {"value": paths * prob.iterations / t, "unit": "scores/s", "cores": threads, "kind": "port", "cpu": host_cpu(),
            "sample": f"level-{name} join, first {n_uids} uids = {paths} joined paths x {prob.iterations} permutations, "
                      f"{t:.1f} s wall, oracle/gcre_oracle.cpp -O3 -march=native"}
    ref = reference_baseline(prob, masks, u, n_uids, p0, p1, threads, port_res)
    if ref is None:
        return port
    ref["port"] = port
    return ref


def reference_baseline(prob, masks, u, n_uids, p0, p1, threads, port_res):
    """The same sample through the reference's own scoring code: oracle/_ref/ref_driver links src/methods.h
    (JoinMethod1/2::score_permute, >95 % of the reference's cycles) compiled from the reference tree;
    the thread pool around it is the driver's (the reference's join_base.cpp needs Rcpp and cannot be built)."""
    import struct
    import subprocess
    import tempfile
    binary = None
    for cand in ("ref_driver_v4", "ref_driver"):
        path = os.path.join(ROOT, "oracle", "_ref", cand)
        if os.path.exists(path):
            try:
                if subprocess.run([path, "--selftest"], capture_output=True, timeout=20).returncode == 0:
                    binary = path
                    break
            except (OSError, subprocess.SubprocessError):
                pass
    if binary is None:
        return None
    M = 1 if prob.method == "method1" else 2
    W = (prob.n_cases + prob.n_ctrls + 63) // 64
    K = prob.iterations
    table = np.ascontiguousarray(prob.value_table, dtype=np.float64)
    rows0 = np.ascontiguousarray(p0[:n_uids], dtype=np.uint64)
    rows1 = np.ascontiguousarray(p1, dtype=np.uint64)
    with tempfile.NamedTemporaryFile(suffix=".gcrebin", delete=False) as f:
        f.write(b"GCREBIN1")
        f.write(struct.pack("<8i", M, prob.n_cases, prob.n_ctrls, K, W, prob.top_k, u.path_length, threads))
        f.write(struct.pack("<5q", n_uids, rows1.shape[0], len(u.signs), table.shape[0], table.shape[1]))
        f.write(np.ascontiguousarray(u.count[:n_uids], dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(u.location[:n_uids], dtype=np.int64).tobytes())
        f.write(np.ascontiguousarray(u.signs, dtype=np.int32).tobytes())
        f.write(rows0.tobytes())
        f.write(rows1.tobytes())
        f.write(np.ascontiguousarray(masks[:K], dtype=np.uint64).tobytes())
        f.write(table.tobytes())
        tmp = f.name
    try:
        out = subprocess.run([binary, "--bench", tmp], capture_output=True, text=True, timeout=900)
        if out.returncode != 0:
            return None
        res = json.loads(out.stdout)
    except (OSError, subprocess.SubprocessError, ValueError):
        return None
    finally:
        os.unlink(tmp)
    same = [f"{int(b):08x}" for b in port_res.null.view(np.uint32)] == res["null"]
    out = {"value": res["paths"] * K / res["seconds"], "unit": "scores/s", "cores": threads, "kind": "reference",
           "sample": f"level-{u.path_length} join, first {n_uids} uids = {res['paths']} joined paths x {K} permutations, "
                     f"{res['seconds']:.1f} s wall; reference src/methods.h score_permute via {os.path.basename(binary)} "
                     f"(partial reference build, -O3 AVX-512/AVX2; driver-side thread pool)",
           "null_maxima_equal_port": same}
    out["cpu"] = host_cpu()
    out["binary"] = {"name": os.path.basename(binary),
                     "flags": "-O3 -march=x86-64-v4 -mavx512vpopcntdq" if binary.endswith("_v4") else "-O3 -mavx2 -mpopcnt",
                     "vpopcntq_emitted": binary_has(binary, "vpopcntq")}
    # the same sample through the HIP library: reference code <-> GPU at this mask width and permutation count, every run
    try:
        from geneticscre_amd import api
        from geneticscre_amd.uids import UidRelSet
        sub = UidRelSet(u.path_length, u.src[:n_uids], u.trg[:n_uids], u.count[:n_uids], u.location[:n_uids], u.signs)
        ex = api.JoinExec(prob.method, prob.n_cases, prob.n_ctrls, K)
        try:
            ex.top_k = prob.top_k
            ex.set_value_table(table)
            ex.set_permuted_masks(np.ascontiguousarray(masks[:K], dtype=np.uint64))
            g = ex.join(sub, ex.from_words(rows0), ex.from_words(rows1))
            out["null_maxima_equal_gpu"] = [f"{int(b):08x}" for b in g.null.view(np.uint32)] == res["null"]
            out["best_score_equal_gpu"] = f"{int(g.scores.view(np.uint64)[-1]):016x}" == res["best"]
        finally:
            ex.close()
    except Exception as e:   # reported, never required
        out["null_maxima_equal_gpu"] = repr(e)
    return out


def end_to_end(prob, masks, device, total_scores):
    """What a GWASPA() user pays (reference src/wrapper.cpp:205-217): one gcre_process_paths call on a COLD context, from
    host buffers (genotype ints, the K x n permutation matrix, the value table) to host results."""
    from geneticscre_amd import api
    n = prob.n_cases + prob.n_ctrls
    W = (n + 63) // 64
    bits = np.unpackbits(np.ascontiguousarray(masks[:, :W]).view(np.uint8), axis=1, bitorder="little")[:, :n]
    is_case = (np.arange(n) < prob.n_cases).astype(np.uint8)
    perm_cases = (bits == is_case[None, :]).astype(np.int32)       # 1 = label unchanged (R/Utils.R:246-262)
    import dataclasses
    p2 = dataclasses.replace(prob, perm_cases=perm_cases) if dataclasses.is_dataclass(prob) else prob
    if p2 is prob:
        prob.perm_cases = perm_cases
    t0 = time.perf_counter()
    out = api.process_paths(p2, device=device)
    wall = time.perf_counter() - t0
    lib = {k: round(v, 2) for k, v in out["profile"].items() if k.endswith("_ms")}
    host_mb = (prob.data1.size + prob.data2.size + perm_cases.size) * 4 / 1e6 + prob.value_table.size * 8 / 1e6
    return {"ms": wall * 1e3, "scores_per_s": total_scores / wall, "host_input_MB": host_mb,
            "what": "cold context: gcre_create + one gcre_process_paths call from host arrays (genotype ints, K x n permutation ints, "
                    "value table) to host results, PCIe included; never `value`",
            "last_join_profile_ms": lib}


def ref_driver_binary():
    """oracle/_ref/ref_driver(_v4): the partial reference build (travels with the tree; built where /root/reference exists)."""
    import subprocess
    for cand in ("ref_driver_v4", "ref_driver"):
        path = os.path.join(ROOT, "oracle", "_ref", cand)
        if os.path.exists(path):
            try:
                if subprocess.run([path, "--selftest"], capture_output=True, timeout=20).returncode == 0:
                    return path
            except (OSError, subprocess.SubprocessError):
                pass
    return None


def six_join_baseline(cfg, seed, top_k, level_sample, budget_s=10.0):
    """BASELINE.md §3's CPU baseline: the full six-join sequence of the reference harness (test/harness.cpp:121-181:
    levels 1a, 1b, 2 .. L) on a network cut down until the run takes ~10 s, at threads = cores and at threads = 0 (the
    reference's inline mode, src/join_base.cpp:170-171; fewer permutations) -- through the reference's OWN scoring code
    (oracle/_ref/ref_driver --bin --time: src/methods.h score_permute / merge_scores, src/gcre_paths.h PathSet for all six
    joins, `kind: "reference"`), with the oracle port beside it on the same input (`port`) and their digests compared."""
    import subprocess
    import tempfile
    import oracle
    from geneticscre_amd import synth
    from geneticscre_amd.harness_io import write_problem_bin
    threads = host_threads()
    rate = float(level_sample.get("value", 2e9))    # scores/s of the reference's score_permute at `threads`
    K = cfg["perms"]
    best = None
    for genes, edges in ((6000, 40000), (5000, 30000), (4000, 20000), (3000, 12000), (2000, 7000), (1000, 3000)):
        p = synth.make_problem(genes, edges, cfg["cases"], cfg["ctrls"], 0, cfg["length"], method=cfg["method"], top_k=top_k, seed=seed + 1,
                               table=np.zeros((cfg["cases"] + 1, cfg["ctrls"] + 1)))
        names = ["1a", "1b"] + [str(l) for l in range(2, cfg["length"] + 1)]
        paths = sum(int(np.maximum(np.asarray(p.levels.uids[k].count), 0).sum()) for k in names)
        best = (p, paths, genes, edges)
        if paths * K <= rate * budget_s * 1.5:
            break
    p, paths, genes, edges = best
    rng = np.random.default_rng(seed + 2)
    p.value_table = synth.values_table(cfg["cases"], cfg["ctrls"])
    binary = ref_driver_binary()
    out = {"network": f"{genes} genes / {edges} relations, {cfg['cases']}+{cfg['ctrls']} patients, path length {cfg['length']}: {paths} joined paths over six joins",
           "kind": "reference" if binary else "port",
           "code": (f"reference src/methods.h + src/gcre_paths.h through oracle/_ref/{os.path.basename(binary)} --bin --time (join_base.cpp's members are the "
                    "driver's: it needs Rcpp)") if binary else "oracle/gcre_oracle.cpp (port of JoinExec, -O3 -march=native): no partial reference build in this tree"}

    def fnv_null(null):
        h = 1469598103934665603
        for w in np.asarray(null, dtype=np.float32).view(np.uint32).tolist():
            h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return f"{h:016x}"

    for label, nthreads, k_run in (("threads_eq_cores", threads, K), ("threads_0", 0, max(16, K // max(threads, 1)))):
        p.iterations = k_run
        p.perm_cases = synth.case_or_control(cfg["cases"], cfg["ctrls"], k_run, rng)
        entry = {"unit": "scores/s", "threads": nthreads, "permutations": k_run}
        ref = None
        if binary:
            with tempfile.NamedTemporaryFile(suffix=".gcrebin", delete=False) as f:
                tmp = f.name
            try:
                write_problem_bin(tmp, p, nthreads=nthreads)
                r = subprocess.run([binary, "--bin", tmp, "--time"], capture_output=True, text=True, timeout=900)
                if r.returncode == 0:
                    ref = json.loads(r.stdout)
            except (OSError, subprocess.SubprocessError, ValueError):
                ref = None
            finally:
                os.unlink(tmp)
        if ref is not None:
            t = sum(ref["seconds"].values())
            entry.update({"value": paths * k_run / t, "seconds": t, "seconds_per_join": ref["seconds"]})
        if ref is None or label == "threads_eq_cores":
            t0 = time.perf_counter()
            port = oracle.process_paths(p, order="reference", nthreads=nthreads)
            tp = time.perf_counter() - t0
            if ref is None:
                entry.update({"value": paths * k_run / tp, "seconds": tp})
            else:
                entry["port"] = {"value": paths * k_run / tp, "seconds": tp,
                                 "what": "oracle/gcre_oracle.cpp on the same input (its time includes packing the inputs; the reference's is its six join calls)"}
                entry["equal_port"] = all(ref[f"lst{l}"]["null_fnv"] == fnv_null(port[f"lst{l}"].null) and
                                          ref[f"lst{l}"]["best"] == f"{int(port[f'lst{l}'].scores.view(np.uint64)[-1]):016x}"
                                          for l in range(1, cfg["length"] + 1))
        out[label] = entry
    return out


def sensitivity(prob, masks, cfg, seed, device, rates=(0.01, 0.025, 0.05), steps=2):
    """How the headline holds up when the data gets denser: the same network, patients, masks and table with every gene at
    1 %, 2.5 % and 5 % carriers (5 % is the reference's own admission limit, R/Utils.R:185-188), one warm-up + `steps` timed
    passes each.  Reported beside the headline, never part of it."""
    import dataclasses
    import torch
    from geneticscre_amd import api, synth
    out = {}
    n = prob.n_cases + prob.n_ctrls
    for rate in rates:
        rng = np.random.default_rng(seed + int(rate * 1e4))
        data1 = synth.variant_matrix(prob.data1.shape[0], n, rng, fixed_rate=rate)
        p2 = dataclasses.replace(prob, data1=data1, data2=data1[prob.levels.uids["1b"].src])
        plan = api.ResidentPlan(p2, device=device, packed_masks=masks)
        try:
            plan.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                plan.run()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / steps
            tiles = sum(plan.uids[k].total_paths for k in plan.names) * ((prob.iterations + 2047) // 2048)
            out[f"{rate:g}"] = {"value": plan.total_scores() / t, "unit": "scores/s", "ms_per_step": t * 1e3,
                                "lookups_per_path_tile": plan.last_profile.get("ie_lookup_tiles", 0) / max(tiles, 1)}
        finally:
            plan.close()
        del data1, p2
    out["what"] = ("configs[2] with every gene at the given carrier rate instead of 0.05 * U^3 (mean 1.25 %); same network, patients, "
                   "permutation masks and value table as the headline; not part of `value`")
    return out


def result_digest(names, results) -> str:
    """SHA-256 over every level's scores, ids, counts and null maxima: equal digests = bit-identical results."""
    import hashlib
    h = hashlib.sha256()
    for name in names:
        r = results[name]
        for arr in (r.scores, r.src, r.trg, r.cases, r.ctrls, r.null):
            h.update(np.ascontiguousarray(arr).tobytes())
    return h.hexdigest()


def other_configs(args, device, steps=2):
    """The other single-GPU BASELINE configs, measured by the same process right after the headline (one warm-up + `steps`
    timed passes each, inputs resident): configs[1], configs[3] on ONE GPU, the signed method on configs[2]'s geometry and
    configs[4].  Parity-test cases with a clock on them, never part of `value`; each carries the digest of its results."""
    import torch
    from geneticscre_amd import api
    runs = [("subgraph", {}, "BASELINE configs[1]"),
            ("roofline", {"method": "method2"}, "configs[2] geometry, method2 (the reference's default method, test/harness.cpp:46)"),
            ("sharded", {}, "BASELINE configs[3] on ONE GPU (the geometry BASELINE names for eight)"),
            ("signed", {}, "BASELINE configs[4] on ONE GPU, synthetic network of 60,000 relations (see `network_note`)")]
    out = {}
    for name, override, what in runs:
        key = name if not override else f"{name}_{override['method']}"
        t_all = time.perf_counter()
        try:
            cfg = dict(CONFIGS[name])
            cfg.update(override)
            prob, masks = build_inputs(cfg, args.seed, args.top_k)
            plan = api.ResidentPlan(prob, device=device, packed_masks=masks, mask_seed=args.seed if masks is None else None)
            try:
                plan.set_window(plan.planned_window())
                plan.run()
                torch.cuda.synchronize()
                acc = {}
                t0 = time.perf_counter()
                for _ in range(steps):
                    last = plan.run()
                    for k, v in plan.last_profile.items():
                        acc[k] = acc.get(k, 0) + v
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / steps
                null_s = acc.get("null_kernel_ms", 0.0) / 1e3
                ach = acc.get("null_alg_bytes", 0.0) / 1e9 / null_s if null_s > 0 else 0.0
                K = prob.iterations
                out[key] = {
                    "what": what, "value": plan.total_scores() / t, "unit": "scores/s", "ms_per_step": t * 1e3, "steps": steps,
                    "workload": f"{cfg['genes']} genes / {cfg['edges']} relations, {prob.n_cases}+{prob.n_ctrls} patients, {K} permutations, "
                                f"path length {prob.path_length}, {prob.method}",
                    "paths_per_level": {k: plan.uids[k].total_paths for k in plan.names},
                    "permutation_windows": len(plan.windows()),
                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "avg_null_ms_per_join": null_s * 1e3 / max(int(acc.get("null_kernel_launches", 0)), 1)},
                    "phases_ms_per_step": {k: acc.get(k, 0.0) / steps for k in ("null_kernel_ms", "stats_kernel_ms", "total_ms")},
                    "lookup_tiles_per_step": int(acc.get("ie_lookup_tiles", 0)) // steps,
                    "masks": "uploaded" if masks is not None else "drawn on the device (gcre_generate_perm_masks)",
                    "table": "hypergeometric (numpy restatement of getValuesTable)" if masks is not None
                             else ("hypergeometric (native gcre_values_table: R's dhyper restated, unpinned against R; summation order: "
                                   + ("R's index order" if api.values_table_exact_order(prob.n_cases, prob.n_ctrls) else "sorted prefix sum") + ")"),
                    "result_sha256": result_digest(plan.names, last),
                }
                if name == "signed":
                    out[key]["network_note"] = ("60,000 relations give 2.0 M level-5 paths; a network of 200,000 relations (the other "
                                                "configs') has ~3e8 level-5 paths and has not been run on one GPU")
            finally:
                plan.close()
            del prob, masks, plan
        except Exception as e:   # reported, never required
            out[key] = {"what": what, "error": repr(e)}
        out[key]["wall_s"] = time.perf_counter() - t_all
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: roofline (BASELINE configs[2]) on one GPU, sharded (configs[3], the geometry BASELINE names "
                         "for 8 GPUs: 10x the permutations) on several")
    ap.add_argument("--edges", type=int, default=0, help="override the synthetic network's edge count")
    ap.add_argument("--perms", type=int, default=0)
    ap.add_argument("--carrier-rate", type=float, default=0.0,
                    help="sensitivity sweep: every gene at this carrier rate instead of 0.05 * U^3 (mean 1.25 %%)")
    ap.add_argument("--method", default="", choices=["", "method1", "method2"], help="override the config's scoring method")
    ap.add_argument("--top-k", type=int, default=100)
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong = the same workload for every N (K permutations in total), weak = K per GPU (K x N)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the cold one-shot gcre_process_paths measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the 1 % / 2.5 % / 5 % carrier-rate passes")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the extra passes with kept inspections")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the passes over the other single-GPU configs (configs[1], [3], [4], method2) after the headline")
    ap.add_argument("--no-one-gpu-reference", action="store_true",
                    help="N > 1: skip the two passes of the whole workload on rank 0's GPU alone")
    ap.add_argument("--no-exchange", action="store_true", help="N > 1: ranks do not share their running maxima during a join")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + several ranks on one GPU is a rehearsal of the N > 1 path, not a measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library has no CPU fallback")
    local_rank %= max(torch.cuda.device_count(), 1)      # rehearsal: several ranks may share the one GPU of a test box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    if args.config is None:
        args.config = "roofline" if world == 1 else "sharded"
    cfg = dict(CONFIGS[args.config])
    if args.method:
        cfg["method"] = args.method
    if args.edges:
        cfg["edges"] = args.edges
    if args.perms:
        cfg["perms"] = args.perms
    if args.carrier_rate:
        cfg["carrier_rate"] = args.carrier_rate
    perms_per_gpu = cfg["perms"]
    if args.scaling == "weak":
        cfg["perms"] *= world
    else:
        perms_per_gpu = None
    prob, masks = build_inputs(cfg, args.seed, args.top_k)

    from geneticscre_amd import api
    from geneticscre_amd.dist import agree_window, exchange_level
    plan = api.ResidentPlan(prob, device=local_rank, packed_masks=masks, mask_seed=args.seed if masks is None else None)
    # every rank sizes its permutation window from its own free memory: they must walk the same windows
    plan.set_window(agree_window(plan.planned_window(), world, device=dev if (world > 1 and args.backend == "nccl") else None))
    K, top_k = prob.iterations, prob.top_k
    d_null = torch.zeros(max(K, 1), dtype=torch.float32, device=dev)
    prof_acc = {}

    def on_level(name, r, shard, window):
        if world == 1:
            return r
        # RCCL: MAX all-reduce of this window's null maxima + all-gather/merge of the top-k tables (geneticscre_amd/dist.py)
        k0, k1 = window
        if args.backend == "gloo":   # CPU collectives: stage the maxima through host memory
            h_null = d_null[k0:k1].cpu()
            best = exchange_level(r.scores, r.src, r.trg, r.cases, r.ctrls, h_null, top_k, world)
            null = h_null.numpy()
        else:
            best = exchange_level(r.scores, r.src, r.trg, r.cases, r.ctrls, d_null[k0:k1], top_k, world, device=dev)
            null = d_null[k0:k1].cpu().numpy()
        return api.JoinResult(best[:, 0].copy(), best[:, 1].astype(np.int32), best[:, 2].astype(np.int32),
                              best[:, 3].astype(np.int32), best[:, 4].astype(np.int32), null)

    def exchange(name, k0, k1):
        # mid-join: every rank learns the others' running maxima and prunes against them (gcre_join_opts.exchange)
        if args.backend == "gloo":
            h = d_null[k0:k1].cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            d_null[k0:k1].copy_(h)
        else:
            dist.all_reduce(d_null[k0:k1], op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()

    share = exchange if (world > 1 and not args.no_exchange) else None

    def step():
        out = plan.run(rank, world, d_null_out=d_null.data_ptr(), on_level=on_level, exchange=share)
        for k, v in plan.last_profile.items():
            prof_acc[k] = prof_acc.get(k, 0) + v
        return out

    for _ in range(args.warmup):
        step()
    prof_acc.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_scores = plan.total_scores()            # whole job, all ranks
    value = total_scores * args.steps / elapsed
    timed_prof = dict(prof_acc)

    # Not part of `value`: the same pass with every join's mask-independent half (expansion, observed scores, top-k, kept
    # rows, lists) kept from the pass before -- what a service re-scoring one network against fresh permutations runs.
    steady = None
    if not args.no_steady_state:
        def kept_step():
            return plan.run(rank, world, d_null_out=d_null.data_ptr(), on_level=on_level, keep_inspections=True, exchange=share)
        kept_step()                                # fills the cache
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            kept_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        steady_s = time.perf_counter() - ts
        steady = {"ms_per_step": steady_s * 1e3 / args.steps, "value": total_scores * args.steps / steady_s, "unit": "scores/s",
                  "inspect_replays_per_step": plan.last_profile.get("inspect_replays", 0),
                  "note": "inspection cache kept across passes (gcre_set_inspect_cache): null kernels only; NOT the headline"}
    prof_acc.clear()
    prof_acc.update(timed_prof)
    W = (prob.n_cases + prob.n_ctrls + 63) // 64
    M = 1 if prob.method == "method1" else 2

    # roofline of the dominant kernel on THIS rank: time from HIP events on the library's stream (live), bytes from
    # SURVEY.md §8(d)'s formula (live), HBM traffic and instruction counts from the newest committed PMC passes of the
    # same workload (profiles/rNN_x_pmc.json, tools/profile_round.sh) -- labelled with the file they come from
    null_s = prof_acc.get("null_kernel_ms", 0.0) / 1e3
    launches = max(int(prof_acc.get("null_kernel_launches", 0)), 1)
    alg_bytes = prof_acc.get("null_alg_bytes", 0.0)
    my_scores = prof_acc.get("scores", 0)
    achieved = alg_bytes / 1e9 / null_s if null_s > 0 else 0.0
    row_loads = prof_acc.get("null_row_loads", 0.0)
    roofline = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
        "launches": launches, "avg_launch_ms": null_s * 1e3 / launches, "alg_bytes_per_launch": alg_bytes / launches,
    }
    ie = prof_acc.get("ie_launches", 0) > 0
    pmc, pmc_name = None, None
    default_run = not args.edges and not args.perms and not args.method and not args.carrier_rate and world == 1
    if ie and default_run:
        import glob
        for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
            try:
                d = json.load(open(cand))
            except (OSError, ValueError):
                continue
            if d.get("workload") == args.config and d.get("path_tiles"):
                pmc, pmc_name = d, os.path.basename(cand)
                break
    if pmc is not None:
        roofline["traffic"] = pmc["traffic_bytes_per_null_launch"]
        roofline["traffic_source"] = (f"profiles/{pmc_name}, collected at {pmc.get('head', '?')}: rocprofv3 FETCH_SIZE x 2 (the guide's gfx950 "
                                      "correction for 16-B/lane loads; an upper bound for the 4-B/lane row loads) + WRITE_SIZE of the null "
                                      "kernels, raw KB x 1024, per join; the count planes, lists and recipes -- not in SURVEY's formula -- are most of it")
    if ie:
        nkt = (K + 2047) // 2048
        tiles = sum(plan.uids[k].total_paths for k in plan.names) * nkt / max(world, 1)
        roofline["kernel"] = ("k_null_ie_m2" if prob.method == "method2" else "k_null_ie_q / k_null_ie_m1") + " (+ warm-up slice on k_null_ie)"
        roofline["note"] = ("north-star accounting (algorithmic HBM bytes / kernel time).  The kernels are not HBM-bound: they are bound by "
                            "instruction issue of their VALU + SALU mix (tools/valu_rate.hip mode fulladd+s: ~0.6 of the VALU peak at 4 "
                            "waves per SIMD) and by memory latency at 3-4 waves per SIMD; `valu` gives the fraction of the wave64 VALU "
                            "issue peak; `launches` counts joins (one warm-up + one pruned launch each)")
        if pmc is not None:
            valu = sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in pmc["kernels"].items() if k.startswith("k_null_ie"))
            per_tile = valu / pmc["path_tiles"]
            ach = per_tile * tiles * args.steps / null_s if null_s > 0 else 0.0
            roofline["valu"] = {"achieved": ach, "peak": VALU_PEAK_WAVE_INSTR, "unit": "wave-instr/s", "frac": ach / VALU_PEAK_WAVE_INSTR,
                                "instr_per_path_tile": per_tile,
                                "source": f"SQ_INSTS_VALU of the null kernels in profiles/{pmc_name} / path-tiles of that run; peak = 256 CUs x 4 SIMDs "
                                          "x 2.4 GHz / 2 clocks per wave64 instruction (tools/valu_rate.hip reaches 0.92 of it)"}
    elif row_loads > 0:
        # sparse bit-sliced kernel: bound by the rate at which a CU pulls random 256-byte mask rows out of L2
        # (tools/row_gather_rate.hip measures ~40 G wave-loads/s on this chip), not by HBM and not by the VALU
        roofline["kernel"] = "k_null_sparse"
        roofline["note"] = ("north-star accounting (algorithmic HBM bytes / kernel time); the binding resource is the "
                            "L2->L1 mask-row load rate, see rows")
        roofline["rows"] = {"achieved": row_loads / null_s / 1e9 if null_s > 0 else 0.0, "peak": ROW_LOAD_PEAK_G,
                            "unit": "G wave-loads/s", "frac": (row_loads / null_s / 1e9 / ROW_LOAD_PEAK_G) if null_s > 0 else 0.0}
    else:
        valu_ops = my_scores * 4.0 * W * M        # 2 x (v_and_b32 + v_bcnt_u32_b32) per 64-bit word per score
        roofline["kernel"] = "k_null"
        roofline["note"] = "north-star accounting (algorithmic HBM bytes / kernel time); the kernel is integer-VALU bound, see valu"
        roofline["valu"] = {"achieved": valu_ops / null_s / 1e12 if null_s > 0 else 0.0, "peak": VALU_PEAK_OPS / 1e12,
                            "unit": "Tlane-op/s", "frac": (valu_ops / null_s / VALU_PEAK_OPS) if null_s > 0 else 0.0}

    line = {
        "metric": "path_x_permutation_scores_per_sec", "value": value, "unit": "scores/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[{cfg['idx']}] '{args.config}': synthetic STRINGdb-shaped signed network, "
                        f"{cfg['genes']} genes / {cfg['edges']} relations, {prob.n_cases}+{prob.n_ctrls} patients, "
                        f"{K} permutations, path length {prob.path_length}, {prob.method}"
                        + (f", every gene at {args.carrier_rate:.3%} carriers" if args.carrier_rate else "")
                        + (f" (weak scaling: {perms_per_gpu} permutations per GPU x {world} GPUs)" if world > 1 and args.scaling == "weak" else ""),
            "paths_per_level": {k: plan.uids[k].total_paths for k in plan.names},
            "scores_per_step": total_scores, "permutations_total": K, "permutations_per_gpu": perms_per_gpu if perms_per_gpu else K,
            "top_k": top_k, "seed": args.seed,
            "parallelism": f"paths sharded over {world} GPU(s), RCCL max-all-reduce + top-k all-gather per level",
        },
        "roofline": roofline,
        "kernel_time_frac": null_s / elapsed if elapsed > 0 else None,
        "phases_ms_per_step": {k: prof_acc.get(k, 0.0) / args.steps for k in
                               ("null_kernel_ms", "stats_kernel_ms", "select_ms", "prepare_ms", "inspect_ms", "total_ms")},
        "ie": {k: int(prof_acc.get(k, 0)) // args.steps for k in
               ("ie_launches", "ie_quad_launches", "ie_overlap_lists", "ie_hinted_joins", "ie_plane_joins", "ie_lookup_tiles",
                "inspect_replays")},
    }
    if steady is not None:
        line["steady_state"] = steady
    if world > 1 and not args.no_one_gpu_reference:
        # the same workload on ONE of these GPUs (rank 0 alone, the others wait): what the N-GPU value is a speed-up of
        ref = None
        if rank == 0:
            plan.run(0, 1, d_null_out=d_null.data_ptr())
            torch.cuda.synchronize()
            tr = time.perf_counter()
            for _ in range(2):
                plan.run(0, 1, d_null_out=d_null.data_ptr())
            torch.cuda.synchronize()
            one_s = (time.perf_counter() - tr) / 2
            ref = {"ms_per_step": one_s * 1e3, "value": total_scores / one_s, "unit": "scores/s",
                   "speedup_of_this_run": (elapsed / args.steps and one_s / (elapsed / args.steps)),
                   "note": "same workload, same inputs, rank 0's GPU alone (2 passes after the timed region)"}
        dist.barrier()
        if ref is not None:
            line["one_gpu_same_workload"] = ref
    if rank == 0 and world == 1 and not args.no_end_to_end and masks is not None:
        try:
            line["end_to_end"] = end_to_end(prob, masks, local_rank, total_scores)
        except Exception as e:   # reported, never required
            line["end_to_end"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_sensitivity and masks is not None and args.config == "roofline" and not args.carrier_rate:
        try:
            line["sensitivity"] = sensitivity(prob, masks, cfg, args.seed, local_rank)
        except Exception as e:   # reported, never required
            line["sensitivity"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and masks is None:
        line["cpu_baseline"] = {"skipped": "device-drawn masks at this scale; the baseline is timed on configs[2]"}
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(prob, masks)
            line["cpu_baseline"]["six_joins"] = six_join_baseline(cfg, args.seed, args.top_k, line["cpu_baseline"])
        except Exception as e:   # the baseline is reported, never required
            line["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        # digest of the last step's results: equal for every rank count, kernel form and tuning knob
        line["result_sha256"] = result_digest(plan.names, last)
        lib = api.load_library()
        line["library"] = {"path": os.path.relpath(os.environ.get("GCRE_LIB") or api.lib_path(), ROOT),
                           "abi": int(lib.gcre_abi_version()), "build_flags": lib.gcre_build_flags().decode()}
    if rank == 0 and world == 1 and default_run and args.config == "roofline" and not args.no_other_configs:
        plan.close()          # the headline's sets and planes leave the device first
        line["other_configs"] = other_configs(args, local_rank)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
