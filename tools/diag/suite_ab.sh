#!/bin/bash
# diagnostic: the GPU test files that precede and include test_gpu_ie.py, run with a knob set, failures listed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
i=0
for env in "$@"; do
  i=$((i+1))
  echo "== run $i: $env"
  env $(echo $env | tr ',' ' ') timeout -k 10 400 python -m pytest tests/test_gpu_api.py tests/test_gpu_configs.py tests/test_gpu_exchange.py tests/test_gpu_ie.py -q -m gpu -p no:cacheprovider > gpurun_out/ab/run_$i.log 2>&1
  echo "rc $?"; grep -E "^FAILED|passed|failed|Aborted" gpurun_out/ab/run_$i.log | head -8
done
