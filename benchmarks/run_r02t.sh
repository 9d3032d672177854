#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python benchmarks/bench_stream_mix.py > gpurun_out/r02t_stream.log 2>&1; echo "rc=$?"; cat gpurun_out/r02t_stream.log
timeout -k 10 300 python bench.py --workload poisson2d_spmm --steps 30 --warmup 3 > gpurun_out/r02t_spmm.log 2>&1; tail -1 gpurun_out/r02t_spmm.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("poisson2d_spmm", r["ms_per_step"], "ms frac", r["roofline"]["frac"])'
