#!/bin/bash
# r03o: k-adaptive SpMM (two lanes per row for k <= 8, general C staging, k = 1 through the SpMV kernel, narrow generic groups)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "spmm or panel" > gpurun_out/r03o_pytest_spmm.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03o_pytest_spmm.log
timeout -k 10 300 python benchmarks/bench_spmm_k.py > gpurun_out/r03o_spmm_k.log 2>&1; echo "rc=$?"; cat gpurun_out/r03o_spmm_k.log | tail -12
HPCLA_SPMM_LPR=4 timeout -k 10 300 python benchmarks/bench_spmm_k.py 4 8 > gpurun_out/r03o_spmm_k_lpr4.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r03o_spmm_k_lpr4.log
timeout -k 10 300 python benchmarks/tune_spmm.py --variants 100,0,21 --rounds 7 > gpurun_out/r03o_tune.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r03o_tune.log | cut -c1-150
timeout -k 10 300 python benchmarks/tune_spmm.py --workload sprand --variants 100,0 --rounds 5 --reps 5 > gpurun_out/r03o_tune_sprand.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r03o_tune_sprand.log | cut -c1-150
true
