#!/bin/bash
# r03r: even k that is not a multiple of 4 on the 16-byte-lane SpMM kernel (TAIL2): tests + rate sweep
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "spmm or panel" > gpurun_out/r03r_pytest_spmm.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03r_pytest_spmm.log
timeout -k 10 300 python benchmarks/bench_spmm_k.py > gpurun_out/r03r_spmm_k.log 2>&1; echo "rc=$?"; tail -14 gpurun_out/r03r_spmm_k.log
timeout -k 10 300 python benchmarks/tune_spmm.py --variants 100,0,21 --rounds 7 > gpurun_out/r03r_tune.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r03r_tune.log | cut -c1-150
true
