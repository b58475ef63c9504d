#!/bin/bash
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], 'value %.3e' % d['value'], 'ms/step %.1f' % d['ms_per_step'], {k: round(v,1) for k,v in d['phases_ms_per_step'].items()}, d.get('ie'))" "$*"; }
for w in 32 16 12; do echo "waves/CU $w"; GCRE_SPARSE_WAVES_PER_CU=$w run --steps 2 --warmup 1; done
run --config subgraph --steps 5 --warmup 2
run --config plumbing --steps 5 --warmup 2
run --config sharded --steps 1 --warmup 1
run --config signed --perms 4096 --steps 1 --warmup 1
