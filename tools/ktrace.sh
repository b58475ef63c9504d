#!/bin/bash
# per-kernel time of one bench step (rocprofv3 --kernel-trace --stats); BENCH_ARGS = extra bench.py arguments
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs ${BENCH_ARGS} > gpurun_out/kt/bench.json 2> gpurun_out/kt/err.log || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_ms {float(r['AverageNs'])/1e6:8.3f} {r['Percentage']}%")
PY
