import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ["GCRE_FUZZ_LIGHT"] = "1"; os.environ["GCRE_QUIET"] = "1"
import numpy as np
import test_gpu_fuzz as f
import oracle
from geneticscre_amd.synth import make_problem
from test_gpu_exchange import run_ranks
case = int(sys.argv[1]); world = int(sys.argv[2])
cfg, env = f.draw(case)
env["GCRE_WINDOW_TILES"] = ""
# (the sharded-plan test adds GCRE_EXCHANGE_UNIT = [5, 50, 2000][case % 3] and GCRE_PIVOT_SHARDS = case % 2 to the draw: pass them here)
for kv in sys.argv[3:]:
    k, v = kv.split("="); env[k] = v
for k, v in env.items():
    if v: os.environ[k] = v
    else: os.environ.pop(k, None)
K = max(cfg["perms"], 1); L = min(cfg["length"], 4)
p = make_problem(cfg["genes"], cfg["edges"], cfg["n_cases"], cfg["n_ctrls"], K, L, method=cfg["method"], top_k=cfg["top_k"], seed=cfg["seed"],
                 threshold=cfg["threshold"], table=f.value_table(cfg["table"], cfg["n_cases"], cfg["n_ctrls"], cfg["seed"]))
want = oracle.process_paths(p, order="canonical")
parts, _, _ = run_ranks(p, world, K)
out = []
for name, lst in (("1b","lst1"),("2","lst2"),("3","lst3"),("4","lst4"))[:L]:
    null = np.maximum.reduce([r[name].null for r in parts])
    w = want[lst].null
    bad = np.nonzero(null.view(np.uint32) != w.view(np.uint32))[0]
    per_rank = [int((r[name].null > w).sum()) for r in parts]
    out.append((name, len(bad), "too high per rank", per_rank, "lower" if len(bad) and (null[bad] < w[bad]).all() else ("higher/mixed" if len(bad) else "")))
print(" ".join(sys.argv[3:]) or "as drawn", [(n, b, pr) for n, b, _, pr, _ in out if b or n == "4"])
