#!/bin/bash
# r02d: full GPU suite + 2-rank bench rehearsal on the shared GPU
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 1000 gpurun_out/r02d_pytest.log python -m pytest tests -x -q -m gpu; tail -15 gpurun_out/r02d_pytest.log
step 600 gpurun_out/r02d_bench2.log env HPCLA_ALLOW_SHARED_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --strong-size 4096; tail -c 3000 gpurun_out/r02d_bench2.log
