#!/bin/bash
# r03i: the distributed step's overhead on one GPU after round 3's changes to the wait / push code (poison branch, ack flag),
# 2-D and 3-D slabs, all three orderings
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03i_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 400 gpurun_out/r03i_halo2d.log python benchmarks/bench_halo_overhead.py
grep -E "plain split|halo \+ interior|overhead|timed_out" gpurun_out/r03i_halo2d.log
step 400 gpurun_out/r03i_halo3d.log python benchmarks/bench_halo_overhead.py --dim3
grep -E "plain split|halo \+ interior|overhead|timed_out" gpurun_out/r03i_halo3d.log
