#!/bin/bash
# the other BASELINE geometries on one GPU (parity-test cases, not bench lines): one JSON line each in gpurun_out/sweep/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
for c in subgraph sharded signed; do
  python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > gpurun_out/sweep/$c.json 2> gpurun_out/sweep/$c.err || exit 1
done
python3 bench.py --method method2 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > gpurun_out/sweep/roofline_method2.json 2> gpurun_out/sweep/m2.err || exit 1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/sweep/*.json")):
    d = json.load(open(f))
    print(f"{f.split('/')[-1]:24s} {d['value']:.3e} scores/s {d['ms_per_step']:9.1f} ms/step  scores/step {d['config']['scores_per_step']:.3e}  null {d['phases_ms_per_step']['null_kernel_ms']:8.1f} ms inspector {d['phases_ms_per_step']['stats_kernel_ms']:7.1f} ms  ie {d['ie']['ie_launches']} quad {d['ie']['ie_quad_launches']}")
PY
