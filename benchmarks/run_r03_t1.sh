#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_cabi_from_c.py -x -q -m gpu -k "pass_boundaries or cabi or block_order" > gpurun_out/r03_t1.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r03_t1.log
./tests/cabi/_build/cabi_smoke 4096 2>&1 | tail -3
true
