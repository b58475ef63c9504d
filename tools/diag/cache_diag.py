"""Diagnostic: which (rank, level, window) differs from the oracle with the inspection cache on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import oracle
from geneticscre_amd import api
from geneticscre_amd.synth import make_problem

os.environ["GCRE_NULL_KERNEL"] = "ie"
os.environ["GCRE_PLANES_OUT_MAX_MB"] = sys.argv[1] if len(sys.argv) > 1 else "0"
os.environ["GCRE_WINDOW_TILES"] = "1"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p = make_problem(70, 260, 310, 335, 2300, L, method="method1", top_k=15, seed=14, threshold=0.05)
want = oracle.process_paths(p, order="canonical")
for world in (1, 2):
    for rank in range(world):
        plan = api.ResidentPlan(p)
        seen = {}
        def on_level(name, r, shard, window, seen=seen):
            seen[(name, window)] = r.null.copy()
            return r
        out = plan.run(rank=rank, world=world, on_level=on_level)
        print("world", world, "rank", rank, "replays", plan.last_profile["inspect_replays"])
        for (name, window), null in sorted(seen.items()):
            lvl = {"1a": None, "1b": 1, "2": 2, "3": 3, "4": 4, "5": 5}[name]
            if lvl is None:
                continue
            w = want[f"lst{lvl}"].null[window[0]:window[1]]
            bad = int((null > w).sum())      # a shard's maxima never exceed the whole level's
            if world == 1:
                bad = int((null.view(np.uint32) != w.view(np.uint32)).sum())
            print("   ", name, window, "bad", bad, "of", len(w))
        plan.close()
