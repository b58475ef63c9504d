cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-steady-state --no-sensitivity --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('%.3e'%d['value'], round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms_per_step'].items() if k in ('null_kernel_ms','stats_kernel_ms')})"; }
for k in ${KNOBS:-GCRE_QUIET=1}; do run $(echo $k | tr ',' ' '); done
