#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spgemm" > gpurun_out/r03_spgemm_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r03_spgemm_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python benchmarks/bench_spgemm.py > gpurun_out/r03_spgemm_bench.log 2>&1; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/r03_spgemm_bench.log | cut -c1-250
true
