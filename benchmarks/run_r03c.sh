#!/bin/bash
# r03c: SpMM candidates 13-18 (last entry first, touches, block starts, unrolled staging), both workloads
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03c_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 600 gpurun_out/r03c_tune_spmm.log python benchmarks/tune_spmm.py
tail -14 gpurun_out/r03c_tune_spmm.log | cut -c1-300
step 600 gpurun_out/r03c_tune_spmm_sprand.log python benchmarks/tune_spmm.py --workload sprand --variants 100,0,13,15,16,17,18 --rounds 5 --reps 5
tail -10 gpurun_out/r03c_tune_spmm_sprand.log | cut -c1-300
step 600 gpurun_out/r03c_tune_spmm_sprand1.log python benchmarks/tune_spmm.py --workload sprand --bmult 1 --variants 100,0,13,15,16,17,18 --rounds 5 --reps 5
tail -10 gpurun_out/r03c_tune_spmm_sprand1.log | cut -c1-300
