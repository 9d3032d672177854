#!/bin/bash
# r02q: uncached vector windows; end-to-end rehearsal
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 600 gpurun_out/r02q_halo2d.log python benchmarks/bench_halo_overhead.py; grep -E "plain|\[push\] halo \+|overhead|timed_out" gpurun_out/r02q_halo2d.log
step 600 gpurun_out/r02q_halo3d.log python benchmarks/bench_halo_overhead.py --dim3; grep -E "plain|\[push\] halo \+|overhead|timed_out" gpurun_out/r02q_halo3d.log
step 900 gpurun_out/r02q_multirank.log python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "exchange and (2 or 3)" || { tail -40 gpurun_out/r02q_multirank.log; exit 1; }
tail -2 gpurun_out/r02q_multirank.log
step 900 gpurun_out/r02q_bench2.log env HPCLA_ALLOW_SHARED_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3; tail -c 1500 gpurun_out/r02q_bench2.log
step 600 gpurun_out/r02q_bench.log python bench.py --steps 50 --warmup 5; tail -c 800 gpurun_out/r02q_bench.log
