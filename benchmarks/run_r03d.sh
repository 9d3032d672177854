#!/bin/bash
# r03d: SpMM candidates: unrolled staging, block starts, loader wave (T = 2..16), two halves
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03d_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 600 gpurun_out/r03d_tune_spmm.log python benchmarks/tune_spmm.py
tail -14 gpurun_out/r03d_tune_spmm.log | cut -c1-300
