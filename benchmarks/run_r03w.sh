#!/bin/bash
# r03w: the straight-line SpMV pass in the shipped library: GPU suite, then old (exp0) vs new (prod) in one process
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03w_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r03w_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python benchmarks/tune_spmv_lib.py --variants 0 > gpurun_out/r03w_old_vs_new.log 2>&1; echo "rc=$?"
grep -v "^{" gpurun_out/r03w_old_vs_new.log | tail -20
true
