#!/bin/bash
# r03t: config 4's per-GPU slab (512 x 512 x 64, 7-point): does an XCD-grouped block order (blocks b and b +- 2 on one
# L2) and/or streaming-hinted A loads cut the x re-reads (1.26 x algorithmic reads in the natural order)?
set -o pipefail
mkdir -p gpurun_out
python benchmarks/tune_spmv.py --build-only
timeout -k 10 500 python benchmarks/tune_spmv.py --dim 3 --size 512 --nz 64 --rounds 7 --reps 20 \
    --variants 100,16,21,22,23,24,25,26,27,28,29 > gpurun_out/r03t_spmv3d_xcd.log 2>&1; echo "3d rc=$?"
cat gpurun_out/r03t_spmv3d_xcd.log | grep -v "^{" | tail -16
timeout -k 10 300 python benchmarks/tune_spmv.py --dim 2 --size 4096 --rounds 7 --reps 20 \
    --variants 100,16,21,23,25,27 > gpurun_out/r03t_spmv2d_xcd.log 2>&1; echo "2d rc=$?"
cat gpurun_out/r03t_spmv2d_xcd.log | grep -v "^{" | tail -10
true
