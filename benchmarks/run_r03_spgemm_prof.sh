#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/r03_spgemm_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_spgemm_prof -- python3 benchmarks/bench_spgemm.py > gpurun_out/r03_spgemm_prof.log 2>&1; echo "rc=$?"
find gpurun_out/r03_spgemm_prof -type f ! -name '*kernel_stats.csv' -delete
f=$(find gpurun_out/r03_spgemm_prof -name '*kernel_stats.csv' | head -1)
grep -i "spgemm\|gather_kernel" "$f" | cut -c1-220
true
