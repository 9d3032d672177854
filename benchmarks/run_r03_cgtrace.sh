#!/bin/bash
# kernel timeline of the CG iteration on the final kernels (block order fixed to 64 so that the plan's measurement launches
# do not sit in the per-kernel means)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export HPCLA_BLOCK_ORDER=64
rm -rf gpurun_out/r03_cgtrace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_cgtrace -- python3 bench.py --workload poisson3d_cg --steps 40 --warmup 8 > gpurun_out/r03_cgtrace.log 2>&1; echo "rc=$?"
t=$(find gpurun_out/r03_cgtrace -name '*kernel_trace.csv' | head -1)
python benchmarks/trace_gaps.py "$t" "CG 512x512x64, eager, final kernels of round 3 (XCD groups of 64)" > gpurun_out/r03_cg_gaps_final.txt 2>&1
rm -rf gpurun_out/r03_cgtrace
cat gpurun_out/r03_cg_gaps_final.txt | head -20
true
