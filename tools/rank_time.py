"""Per-rank compute time of a sharded pass, one rank at a time on one GPU (no collectives): what each GPU of an N-GPU
run has to do.  python tools/rank_time.py [world ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GCRE_QUIET", "1")
import numpy as np
import bench
from geneticscre_amd import api

if os.environ.get("EXCH") == "1":      # torch first: it does not find the GPU once the library has initialised HIP
    import torch
    torch.zeros(1, device="cuda")
WEAK = os.environ.get("WEAK") == "1"      # weak scaling: 10,000 permutations per rank instead of 10,000 in total
plan = None
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    times = []
    if plan is None or WEAK:
        cfg = dict(bench.CONFIGS[os.environ.get("CONFIG", "roofline")])
        if WEAK:
            cfg["perms"] *= world
        prob, masks = bench.build_inputs(cfg, 20261003, 100)
        plan = api.ResidentPlan(prob, packed_masks=masks, mask_seed=None if masks is not None else 1)
    if os.environ.get("EXCH") == "1" and world > 1:
        # Mid-join threshold exchange, rehearsed without the other GPUs: pass A runs every rank and records the maxima it
        # would send at each exchange (maxima do not depend on the thresholds), pass B runs the timed ranks again with the
        # MAX over all ranks' records handed back -- the compute a rank of a real N-GPU run does (collective latency aside).
        import torch
        K = plan.problem.iterations
        d_null = torch.zeros(max(K, 1), dtype=torch.float32, device="cuda")
        rec = {}
        for rank in range(world):
            def record(name, k0, k1, rank=rank):
                rec.setdefault((name, k0), {}).setdefault(rank, []).append(d_null[k0:k1].clone())
            plan.run(rank, world, d_null_out=d_null.data_ptr(), exchange=record)
        for rank in sorted({0, world // 2, world - 1}):
            seen = {}
            def merged(name, k0, k1):
                e = seen.get((name, k0), 0)
                seen[(name, k0)] = e + 1
                m = d_null[k0:k1]
                for r, lst in rec[(name, k0)].items():
                    torch.maximum(m, lst[e], out=m)
                torch.cuda.synchronize()
            plan.run(rank, world, d_null_out=d_null.data_ptr(), exchange=merged)
            seen.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.run(rank, world, d_null_out=d_null.data_ptr(), exchange=merged)
            times.append((rank, (time.perf_counter() - t0) * 1e3, {k: round(v, 1) for k, v in plan.last_profile.items() if k.endswith("_ms") or k == "ie_lookup_tiles"},
                          {n: plan.exchange_count(n, world) for n in plan.names}))
        for r, t, pr, ex in times:
            print(f"world {world} rank {r} (exchanges {ex}): {t:.1f} ms  {pr}", flush=True)
        continue
    for rank in sorted({0, world // 2, world - 1}):
        plan.run(rank, world)
        t0 = time.perf_counter()
        plan.run(rank, world)
        times.append((rank, (time.perf_counter() - t0) * 1e3, {k: round(v, 1) for k, v in plan.last_profile.items() if k.endswith("_ms") or k == "ie_lookup_tiles"}))
    for r, t, pr in times:
        print(f"world {world} rank {r}: {t:.1f} ms  {pr}", flush=True)
