"""Wall time of the gcre_process_paths call (what the R shim makes) on configs[2], against the resident plan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GCRE_QUIET", "1")
import numpy as np
import bench
from geneticscre_amd import api

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "roofline"]
prob, masks = bench.build_inputs(cfg, 20261003, 100)
ex = api.JoinExec(prob.method, prob.n_cases, prob.n_ctrls, prob.iterations)
ex.top_k = prob.top_k
ex.set_permuted_masks(masks)
for rep in range(3):
    t0 = time.perf_counter()
    out = api.process_paths(prob, exec_=ex)
    t1 = time.perf_counter()
    print(f"process_paths call {rep}: {(t1 - t0) * 1e3:.1f} ms wall; library profile:",
          {k: round(v, 1) for k, v in out["profile"].items() if k.endswith("_ms")}, flush=True)
