"""Kernel timeline of ONE rank's pass of an N-GPU run (run under rocprofv3 --kernel-trace): python tools/rank_trace.py RANK WORLD"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GCRE_QUIET", "1")
import bench
from geneticscre_amd import api

rank, world = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(bench.CONFIGS[os.environ.get("CONFIG", "roofline")])
if os.environ.get("WEAK") == "1":      # weak scaling: 10,000 permutations per rank
    cfg["perms"] *= world
prob, masks = bench.build_inputs(cfg, 20261003, 100)
plan = api.ResidentPlan(prob, packed_masks=masks, mask_seed=None if masks is not None else 1)
for _ in range(3):
    t0 = time.perf_counter()
    plan.run(rank, world)
    print(f"rank {rank}/{world}: {(time.perf_counter() - t0) * 1e3:.1f} ms", {k: round(v, 2) for k, v in plan.last_profile.items() if k.endswith("_ms")}, flush=True)
