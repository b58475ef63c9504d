#!/bin/bash
# A/B of variant libraries (tools/build_variant.py): one bench run per name in VARIANTS ("-" = the product library),
# prints scores/s, ms per step and the mean time of the null launches of each.   VARIANTS="- q3w4" tools/ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for v in ${VARIANTS:--}; do
  if [ "$v" = "-" ]; then unset GCRE_LIB; else export GCRE_LIB=geneticscre_amd/variants/libgcre_hip_$v.so; fi
  python3 bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs ${BENCH_ARGS} > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab/$v.err; continue; }
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab/{v}.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"{v:12s} {d['value']:.3e} scores/s  {d['ms_per_step']:7.2f} ms/step  null launch {r.get('avg_launch_ms', 0):6.3f} ms x {r.get('launches', 0)}  lookups {d.get('ie', {}).get('ie_lookup_tiles', 0)}  sha {d.get('result_sha256', '')[:12]}")
PY
done
