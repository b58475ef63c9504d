"""Per-rank compute time of a sharded pass, one rank at a time on one GPU (no collectives): what each GPU of an N-GPU
run has to do.  python tools/rank_time.py [world ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GCRE_QUIET", "1")
import numpy as np
import bench
from geneticscre_amd import api

WEAK = os.environ.get("WEAK") == "1"      # weak scaling: 10,000 permutations per rank instead of 10,000 in total
plan = None
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    times = []
    if plan is None or WEAK:
        cfg = dict(bench.CONFIGS["roofline"])
        if WEAK:
            cfg["perms"] *= world
        prob, masks = bench.build_inputs(cfg, 20261003, 100)
        plan = api.ResidentPlan(prob, packed_masks=masks, mask_seed=None if masks is not None else 1)
    for rank in sorted({0, world // 2, world - 1}):
        plan.run(rank, world)
        t0 = time.perf_counter()
        plan.run(rank, world)
        times.append((rank, (time.perf_counter() - t0) * 1e3, {k: round(v, 1) for k, v in plan.last_profile.items() if k.endswith("_ms")}))
    for r, t, pr in times:
        print(f"world {world} rank {r}: {t:.1f} ms  {pr}", flush=True)
