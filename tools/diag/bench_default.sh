cd $GRAFT_REPO_ROOT
s=$(date +%s)
python bench.py > gpurun_out/r04n_bench.json 2> gpurun_out/r04n_bench.err
echo "rc=$? wall=$(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04n_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
for k,v in d["other_configs"].items():
    if isinstance(v, dict): print(k, v.get("value"), v.get("ms_per_step"))
print({k: v.get("value") for k,v in d["sensitivity"].items() if isinstance(v, dict)})
print(d["end_to_end"].get("ms"), d["steady_state"].get("ms_per_step"))
PY
