#!/bin/bash
# r03x: SpMM library variants (spmm.hip under -DHPCLA_EXP=N) against the shipped library, stencil and config 5's pattern
set -o pipefail
mkdir -p gpurun_out
V=${1:-100,200,201}
timeout -k 10 300 python benchmarks/tune_spmm.py --workload poisson2d --variants $V --rounds 9 > gpurun_out/r03x_spmm2d.log 2>&1; echo "2d rc=$?"
grep -v "^{" gpurun_out/r03x_spmm2d.log | tail -8
timeout -k 10 300 python benchmarks/tune_spmm.py --workload sprand --variants $V --rounds 5 --reps 5 > gpurun_out/r03x_sprand.log 2>&1; echo "sprand rc=$?"
grep -v "^{" gpurun_out/r03x_sprand.log | tail -8
true
