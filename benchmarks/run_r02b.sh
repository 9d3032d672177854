#!/bin/bash
# r02b: first run of the peer-window transport
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { # step <timeout> <log> <cmd...>
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; tail -20 "$log"; exit $rc; fi
  return $rc
}
step 600 gpurun_out/r02b_selfworker_push.log env HPCLA_FORCE_RCCL=1 HPCLA_HALO_MODE=push python tests/_halo_self_worker.py || { tail -30 gpurun_out/r02b_selfworker_push.log; exit 1; }
step 900 gpurun_out/r02b_multirank.log python -m pytest tests/test_gpu_multirank.py -x -q -m gpu; tail -40 gpurun_out/r02b_multirank.log
step 600 gpurun_out/r02b_halo.log python benchmarks/bench_halo_overhead.py; tail -30 gpurun_out/r02b_halo.log
step 600 gpurun_out/r02b_bench.log python bench.py --steps 50 --warmup 5; tail -2 gpurun_out/r02b_bench.log
