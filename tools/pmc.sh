#!/bin/bash
# PMC passes over one bench step (separate passes: SQ issue/wait split, instruction mix; optional HBM/L2 passes with FULL=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc/sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs ${BENCH_ARGS} > gpurun_out/pmc/sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc/sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs ${BENCH_ARGS} > gpurun_out/pmc/sq2.log 2>&1 || exit 1
if [ -n "$FULL" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > gpurun_out/pmc/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc/tcc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs > gpurun_out/pmc/tcc.log 2>&1 || exit 1
fi
python3 - <<'PY'
import csv, glob, collections
for d in ("sq", "sq2", "fetch", "tcc"):
    for f in glob.glob(f"gpurun_out/pmc/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "null_ie" in k or "k_stats" in k or "ie_fill" in k:
                print(d, k, {a: f"{b:.4g}" for a, b in v.items()})
    for f in glob.glob(f"gpurun_out/pmc/{d}/**/*kernel_trace.csv", recursive=True):
        t = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            t[r["Kernel_Name"].split("(")[0][:60]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d == "sq":
            print({k: f"{v:.1f} ms" for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:12]})
PY
