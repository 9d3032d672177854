#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m pytest tests/test_cabi_from_c.py tests/test_gpu_parity.py -x -q -m gpu -k "cabi or self_exchange" -s > gpurun_out/r02y_cabi.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r02y_cabi.log
timeout -k 10 120 tests/cabi/_build/cabi_window_pair > gpurun_out/r02y_pair.log 2>&1; echo "rc=$?"; cat gpurun_out/r02y_pair.log
