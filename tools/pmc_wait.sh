#!/bin/bash
# where the waves of the kernels whose name contains $KERNEL wait: LDS against vector memory, queue levels, FIFO stalls
# (two PMC passes, --kernel-trace only):  KERNEL=k_stats_ie3 tools/pmc_wait.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcw && mkdir -p gpurun_out/pmcw
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs $BENCH_ARGS"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d gpurun_out/pmcw/a -- $B > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmcw/b -- $B > /dev/null 2>&1 || exit 1
python3 - "${KERNEL:-k_stats}" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in "ab":
    for f in glob.glob(f"gpurun_out/pmcw/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if sys.argv[1] in n:
                acc[n.split("(")[0][-48:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, {c: f"{x:.4g}" for c, x in sorted(v.items())})
PY
