#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "test_cg_matches_oracle or test_cg_fused_equals_unfused" > gpurun_out/r02o_cg.log 2>&1; echo "rc=$?"; grep -E "CG |passed|failed" gpurun_out/r02o_cg.log
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "push_transport_ranks_exchange and 4" > gpurun_out/r02o_mr4.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r02o_mr4.log
timeout -k 10 600 python benchmarks/bench_halo_overhead.py --dim3 > gpurun_out/r02o_halo3d.log 2>&1; echo "rc=$?"; grep -E "3-D|plain|halo \+|overhead|timed_out" gpurun_out/r02o_halo3d.log
