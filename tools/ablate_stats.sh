#!/bin/bash
for a in ${ABL:-0 1 2 3}; do
  GCRE_STATS_ABLATE=$a python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/abl_s$a.json 2>/dev/null
  python - "$a" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abl_s{sys.argv[1]}.json"))
print("stats ablate", sys.argv[1], "stats_ms/step %.1f" % d["phases_ms_per_step"]["stats_kernel_ms"], flush=True)
PY
done
