#!/bin/bash
# instruction mix, issue and wait cycles of the kernels whose name contains $KERNEL, one pass of bench.py $BENCH_ARGS
# (two PMC passes, --kernel-trace only):  KERNEL=k_null_ie_m2 BENCH_ARGS="--method method2" tools/pmc_kernel.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmck && mkdir -p gpurun_out/pmck
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs $BENCH_ARGS"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmck/a -- $B > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmck/b -- $B > /dev/null 2>&1 || exit 1
python3 - "${KERNEL:-k_null}" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in "ab":
    for f in glob.glob(f"gpurun_out/pmck/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if sys.argv[1] in n:
                acc[n.split("(")[0][-48:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, {c: f"{x:.4g}" for c, x in sorted(v.items())})
PY
