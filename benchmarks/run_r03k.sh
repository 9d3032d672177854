#!/bin/bash
# r03k: SpMM with WIDE gather rounds (3 / 5 / 6 / 8 entries per round): fewer, fatter waves, more bytes in flight per wave
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python benchmarks/tune_spmm.py --variants 100,0,21,23:3,23:5,23:6,23:8 --rounds 9 > gpurun_out/r03k_tune_spmm.log 2>&1; echo rc=$?
tail -10 gpurun_out/r03k_tune_spmm.log | cut -c1-200
timeout -k 10 400 python benchmarks/tune_spmm.py --workload sprand --variants 100,0,23:5,23:8 --rounds 5 --reps 5 > gpurun_out/r03k_tune_spmm_sprand.log 2>&1; echo rc=$?
tail -7 gpurun_out/r03k_tune_spmm_sprand.log | cut -c1-200
true
