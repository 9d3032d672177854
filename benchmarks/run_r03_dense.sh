#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or row_vector or widened or scalar_ops" > gpurun_out/r03_dense_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r03_dense_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python benchmarks/bench_rank4.py > gpurun_out/r03_dense_bench.log 2>&1; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/r03_dense_bench.log | head -14 | cut -c1-200
true
