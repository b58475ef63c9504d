#!/bin/bash
# time split of the IE null kernel: GCRE_IE_ABLATE bits (wrong results, diagnostics only)
for a in ${ABL:-0 4 7}; do
  GCRE_IE_ABLATE=$a python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/abl_$a.json 2>/dev/null
  python - "$a" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abl_{sys.argv[1]}.json"))
print("ablate", sys.argv[1], "null_ms/step %.1f" % d["phases_ms_per_step"]["null_kernel_ms"], "step %.1f" % d["ms_per_step"], d["phases_ms_per_step"], flush=True)
PY
done
