#!/bin/bash
# the 4-rank stall again with a 3 s spin bound (does the stall scale with it?) and a rocm-smi watcher
mkdir -p gpurun_out
export HPCLA_ALLOW_SHARED_GPU=1 HPCLA_BENCH_VERBOSE=1 HPCLA_BENCH_OUTER_LIMIT_S=400 HPCLA_PUSH_TIMEOUT_S=3 HPCLA_BENCH_DUMP_S=60
( for i in $(seq 1 16); do sleep 15; echo "--- t=$((i*15))s"; rocm-smi --showuse --showmemuse 2>/dev/null | grep -i "GPU\[0\]" | head -4; done ) > gpurun_out/r03_reh4_smi.log 2>&1 &
W=$!
HPCLA_BENCH_EXTRAS=sprand_spmm,poisson2d_spmm,sprand_spmm_panel_order timeout -k 10 380 python bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r03_reh4.log 2>&1; echo "rc=$?"
kill $W 2>/dev/null
grep "bench +" gpurun_out/r03_reh4.log | grep -v "rank [1-9]" | cut -c1-140 | tail -6
grep "extra +" gpurun_out/r03_reh4.log | grep "rank 0" | tail -8 | cut -c1-120
cat gpurun_out/r03_reh4_smi.log | head -60
true
