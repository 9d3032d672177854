#!/bin/bash
# r03v: why is the shipped SpMV kernel 2.4 % behind the bare harness kernel?  Library variants, interleaved.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python benchmarks/tune_spmv_lib.py --variants 3,897,899,1923 > gpurun_out/r03v_lib_variants.log 2>&1; echo "rc=$?"
grep -v "^{" gpurun_out/r03v_lib_variants.log | tail -44
timeout -k 10 200 python benchmarks/tune_spmv.py --dim 2 --size 4096 --rounds 7 --reps 20 --variants 3,897,899,1923 > gpurun_out/r03v_spmv2d.log 2>&1
grep -v "^{" gpurun_out/r03v_spmv2d.log | tail -4
true
