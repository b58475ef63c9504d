"""Diagnostic: the scenario of test_recipes_replace_stored_planes[2-0] repeated, per knob setting; counts mismatching levels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["GCRE_QUIET"] = "1"
os.environ["GCRE_NULL_KERNEL"] = "ie"
os.environ["GCRE_PLANES_OUT_MAX_MB"] = "0"
os.environ["GCRE_WINDOW_TILES"] = "1"
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    os.environ[k] = v
import oracle
from geneticscre_amd import api
from geneticscre_amd.synth import make_problem
reps = int(sys.argv[1])
p = make_problem(70, 260, 310, 335, 2300, 5, method="method1", top_k=15, seed=14, threshold=0.05)
want = oracle.process_paths(p, order="canonical")
bad = {}
for rep in range(reps):
    parts, replays = [], []
    for rank in range(2):
        plan = api.ResidentPlan(p)
        parts.append(plan.run(rank=rank, world=2))
        replays.append(plan.last_profile["inspect_replays"])
        plan.close()
    for name, lvl in (("1b", 1), ("2", 2), ("3", 3), ("4", 4), ("5", 5)):
        null = np.maximum(parts[0][name].null, parts[1][name].null)
        n = int((null.view(np.uint32) != want[f"lst{lvl}"].null.view(np.uint32)).sum())
        if n:
            bad.setdefault(name, []).append((rep, n, int(np.nonzero(null != want[f"lst{lvl}"].null)[0].min())))
print(sys.argv[2:], "reps", reps, "replays", replays, "bad", bad, flush=True)
