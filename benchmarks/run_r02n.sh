#!/bin/bash
# r02n: device-resident epochs (graph-capturable distributed steps)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 600 gpurun_out/r02n_selfworker_push.log env HPCLA_FORCE_RCCL=1 HPCLA_HALO_MODE=push python tests/_halo_self_worker.py || { tail -30 gpurun_out/r02n_selfworker_push.log; exit 1; }
step 900 gpurun_out/r02n_multirank.log python -m pytest tests/test_gpu_multirank.py -x -q -m gpu || { tail -40 gpurun_out/r02n_multirank.log; exit 1; }
tail -3 gpurun_out/r02n_multirank.log
step 600 gpurun_out/r02n_halo2d.log python benchmarks/bench_halo_overhead.py; grep -E "plain|halo \+|overhead|timed_out" gpurun_out/r02n_halo2d.log
step 900 gpurun_out/r02n_pytest.log python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_multirank.py; tail -4 gpurun_out/r02n_pytest.log
