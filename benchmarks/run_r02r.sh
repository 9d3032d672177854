#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 900 gpurun_out/r02r_pytest.log python -m pytest tests -x -q -m gpu; tail -6 gpurun_out/r02r_pytest.log
step 900 gpurun_out/r02r_bench.log python bench.py --gpus 1 --steps 20 --warmup 5; tail -c 6000 gpurun_out/r02r_bench.log
step 900 gpurun_out/r02r_bench2.log env HPCLA_ALLOW_SHARED_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline; tail -c 5000 gpurun_out/r02r_bench2.log
