#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python benchmarks/tune_spmv_lib.py --variants "" --orders 1,2,4,8,16,32 --dims 2,8 --rounds 9 > gpurun_out/r03z3_orders.log 2>&1; echo "rc=$?"
grep -v "^{" gpurun_out/r03z3_orders.log | grep "plain\|^#" | tail -50
true
