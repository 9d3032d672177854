#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r02s_pytest.log 2>&1; echo "rc=$?"; tail -22 gpurun_out/r02s_pytest.log
