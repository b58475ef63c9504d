"""Where the cold one-shot call spends its time: python tools/diag/e2e_time.py (GCRE_HOST_TIMING=1 prints the host phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GCRE_QUIET", "1")
import numpy as np
import bench
from geneticscre_amd import api
cfg = dict(bench.CONFIGS["roofline"])
prob, masks = bench.build_inputs(cfg, 20261003, 100)
total = prob.total_scores()
for rep in range(2):
    print(f"---- cold call {rep}", flush=True)
    r = bench.end_to_end(prob, masks, 0, total)
    print({k: v for k, v in r.items() if k in ("ms", "scores_per_s", "host_input_MB", "last_join_profile_ms")}, flush=True)
