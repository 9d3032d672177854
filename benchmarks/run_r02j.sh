#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python benchmarks/check_cg_two_rank_residual.py > gpurun_out/r02j_cg_check.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r02j_cg_check.log
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r02j_multirank.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r02j_multirank.log
