#!/bin/bash
# what profiles/ holds for a build: kernel stats (rocprofv3 --kernel-trace --stats), PMC passes (separate), bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_a}
HEAD_SHA=${HEAD_SHA:-unknown}
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > gpurun_out/prof/bench_kt.json 2> gpurun_out/prof/kt.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/prof/tcc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/prof/sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof/sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > /dev/null 2>&1 || exit 1
python3 - "$TAG" "$HEAD_SHA" <<'PY'
import csv, glob, collections, json, sys, shutil
tag, head = sys.argv[1], sys.argv[2]
out = {"_provenance": "rocprofv3 --pmc passes (separate: FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | SQ issue/wait | SQ instruction mix), --kernel-trace only, "
       "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs; raw counter sums over the dispatches of each kernel; FETCH/WRITE_SIZE in KB "
       "(MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of 16-B/lane streams: traffic_bytes_per_null_launch "
       "doubles it -- exact for the plane loads, an upper bound for the 4-B/lane mask-row loads)",
       "workload": "roofline", "head": head, "kernels": {}}
for d in ("fetch", "tcc", "sq", "sq2"):
    for f in glob.glob(f"gpurun_out/prof/{d}/**/*counter_collection.csv", recursive=True):
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gcre::", "")
            e = out["kernels"].setdefault(k, {})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
        for k, v in seen.items():
            out["kernels"][k]["dispatches"] = len(v)
null = [k for k in out["kernels"] if k.startswith("k_null_ie")]
fetch = sum(out["kernels"][k].get("FETCH_SIZE", 0) for k in null) * 1024
write = sum(out["kernels"][k].get("WRITE_SIZE", 0) for k in null) * 1024
launches = 5   # joins per pass: every join is one null "launch" in bench.py's accounting (warm-up slice + pruned kernel)
out["null_kernels"] = null
out["traffic_bytes_per_null_launch"] = (2 * fetch + write) / launches
out["fetch_bytes_raw"], out["write_bytes_raw"] = fetch, write
b = json.loads(open("gpurun_out/prof/bench_kt.json").read().strip().splitlines()[-1])
K = b["config"]["permutations_total"]
out["path_tiles"] = sum(b["config"]["paths_per_level"].values()) * ((K + 2047) // 2048)
json.dump(out, open(f"gpurun_out/prof/{tag}_pmc.json", "w"), indent=1)
shutil.copy(f"gpurun_out/prof/{tag}_pmc.json", f"profiles/{tag}_pmc.json")   # the bench line below reads the newest one
f = glob.glob("gpurun_out/prof/kt/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(f, f"gpurun_out/prof/{tag}_kernel_stats_roofline_steps2.csv")
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_ms {float(r['AverageNs'])/1e6:8.3f}")
PY
# the full default line (cpu_baseline, end_to_end, steady_state), with this run's counters behind roofline.traffic / .valu
python3 bench.py --steps 3 --warmup 1 > gpurun_out/prof/${TAG}_bench_roofline.json 2> gpurun_out/prof/bench.err || exit 1
head -c 1500 gpurun_out/prof/${TAG}_bench_roofline.json
