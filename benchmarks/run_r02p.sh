#!/bin/bash
# r02p: sharded epoch release
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 600 gpurun_out/r02p_selfworker_push.log env HPCLA_FORCE_RCCL=1 HPCLA_HALO_MODE=push python tests/_halo_self_worker.py || { tail -30 gpurun_out/r02p_selfworker_push.log; exit 1; }
step 600 gpurun_out/r02p_halo2d.log python benchmarks/bench_halo_overhead.py; grep -E "plain|halo \+|overhead|timed_out" gpurun_out/r02p_halo2d.log
step 600 gpurun_out/r02p_halo3d.log python benchmarks/bench_halo_overhead.py --dim3; grep -E "3-D|plain|halo \+|overhead|timed_out" gpurun_out/r02p_halo3d.log
step 1000 gpurun_out/r02p_pytest.log python -m pytest tests -x -q -m gpu; tail -4 gpurun_out/r02p_pytest.log
