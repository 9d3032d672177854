#!/bin/bash
# r03b: GPU suite on the round-3 code (timeout poison, k = 1 plan, panel order, native CG loop) + SpMM candidates
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03b_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 600 gpurun_out/r03b_tune_spmm.log python benchmarks/tune_spmm.py
tail -16 gpurun_out/r03b_tune_spmm.log | cut -c1-400
step 600 gpurun_out/r03b_tune_spmm_sprand.log python benchmarks/tune_spmm.py --workload sprand --variants 100,0,9,11,5 --rounds 5 --reps 5
tail -9 gpurun_out/r03b_tune_spmm_sprand.log | cut -c1-400
step 1000 gpurun_out/r03b_pytest.log python -m pytest tests -m gpu -q
tail -25 gpurun_out/r03b_pytest.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 120 rocprofv3 -L > gpurun_out/r03b_counters.txt 2>&1
grep -c . gpurun_out/r03b_counters.txt
