"""GPU idle time between the kernels of one bench step, from a rocprofv3 kernel trace (tools/ktrace.sh)."""
import csv, glob, sys
f = sorted(glob.glob("gpurun_out/kt/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step = everything after the last but one k_null_ie_q<12 launch ... simpler: take the last N kernels between two level-4 quad launches
longest = max(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))["Kernel_Name"]
big = [i for i, r in enumerate(rows) if r["Kernel_Name"] == longest]   # one per step: the last level's null kernel
if len(big) < 2:
    sys.exit("need two steps")
lo, hi = big[-2] + 1, big[-1] + 1
step = rows[lo:hi]
t0 = int(rows[big[-2]]["End_Timestamp"])
busy, gaps, prev_end = 0, [], t0
per = {}
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > prev_end:
        gaps.append((s - prev_end, r["Kernel_Name"][:60]))
    busy += e - max(s, prev_end) if e > prev_end else 0
    prev_end = max(prev_end, e)
    n = r["Kernel_Name"].split("(")[0][:50]
    per[n] = per.get(n, [0, 0]); per[n][0] += 1; per[n][1] += e - s
span = prev_end - t0
print(f"step span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {sum(g for g,_ in gaps)/1e6:.2f} ms in {len(gaps)} gaps, {len(step)} kernels")
for g, n in sorted(gaps, reverse=True)[:15]:
    print(f"  gap {g/1e3:8.1f} us before {n}")
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {n:50s} x{c:3d} {t/1e6:8.3f} ms")
