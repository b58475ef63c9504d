#!/bin/bash
# Sensitivity of the headline to the data: configs[2] with every gene at 1 %, 2.5 %, 5 % carriers, and
# the default data with the pruning off; one JSON line per run in gpurun_out/sens/, summary on stdout.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sens
for r in 0.01 0.025 0.05; do
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs --carrier-rate $r > gpurun_out/sens/rate_$r.json 2> gpurun_out/sens/rate_$r.err || exit 1
done
GCRE_IE_PRUNE=0 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > gpurun_out/sens/noprune.json 2> gpurun_out/sens/noprune.err || exit 1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > gpurun_out/sens/default.json 2> gpurun_out/sens/default.err || exit 1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/sens/*.json")):
    d = json.load(open(f))
    tiles = sum(d["config"]["paths_per_level"].values()) * 5
    print(f"{f.split('/')[-1]:18s} {d['value']:.3e} scores/s  {d['ms_per_step']:8.1f} ms/step  kernel {d['roofline']['kernel'][:28]:28s} "
          f"ie {d['ie']['ie_launches']} quad {d['ie']['ie_quad_launches']} lookups/path-tile {d['ie']['ie_lookup_tiles'] / tiles:.3f}")
PY
