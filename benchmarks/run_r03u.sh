#!/bin/bash
# r03u: where do the 42 s before the second config-5 record go (2 ranks sharing the GPU)?  Then the XCD-group experiment.
set -o pipefail
mkdir -p gpurun_out
export HPCLA_ALLOW_SHARED_GPU=1 HPCLA_BENCH_VERBOSE=1
HPCLA_BENCH_EXTRAS=sprand_spmm,poisson2d_spmm,sprand_spmm_panel_order timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03u_a.log 2>&1; echo "a rc=$?"
grep "extra +\|bench +" gpurun_out/r03u_a.log | grep -v "transport" | cut -c1-160
HPCLA_BENCH_EXTRAS=sprand_spmm,sprand_spmm_panel_order timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03u_b.log 2>&1; echo "b rc=$?"
grep "extra +" gpurun_out/r03u_b.log | cut -c1-160
unset HPCLA_BENCH_VERBOSE
./benchmarks/run_r03t.sh
