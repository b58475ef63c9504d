#!/bin/bash
# timeline (kernel, start offset, duration, gap before) of the last pass of one rank: RANK=3 WORLD=8 tools/rank_trace.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/rt && mkdir -p gpurun_out/rt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rt -- python3 tools/rank_trace.py ${RANK:-3} ${WORLD:-8} > gpurun_out/rt/out.log 2> gpurun_out/rt/err.log || exit 1
cat gpurun_out/rt/out.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/rt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last pass = after the last gap > 20 ms ... passes are back to back, so split on the k_stats of level 1a instead: take the last third
n = len(rows) // 3
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev = t0
out = open("gpurun_out/rt/timeline.txt", "w")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write(f"{(s - t0) / 1e6:9.3f} ms  gap {(s - prev) / 1e3:8.1f} us  dur {(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:60]}\n")
    prev = e
out.write(f"total {(prev - t0) / 1e6:.3f} ms, kernels {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows) / 1e6:.3f} ms\n")
out.close()
print(open("gpurun_out/rt/timeline.txt").read()[-600:])
PY
