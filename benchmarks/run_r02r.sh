#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1100 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "exchange" > gpurun_out/r02r_multirank.log 2>&1; echo "rc=$?"; tail -40 gpurun_out/r02r_multirank.log
