#!/bin/bash
# r02u: LDS-tiled SpMM
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 600 gpurun_out/r02u_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -m gpu -k "spmm or SpMM or (push_transport and 2)" || { tail -40 gpurun_out/r02u_pytest.log; exit 1; }
tail -2 gpurun_out/r02u_pytest.log
OUT=gpurun_out/r02u_spmm_tile.log; : > $OUT
for T in 0 1 0 1; do
  export HPCLA_SPMM_TILE=$T
  step 300 gpurun_out/r02u_tmp.log python bench.py --workload poisson2d_spmm --steps 30 --warmup 3 || { tail -5 gpurun_out/r02u_tmp.log; exit 1; }
  echo "TILE=$T poisson2d_spmm: $(tail -1 gpurun_out/r02u_tmp.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], "ms", r["value"], "GFLOP/s frac", r["roofline"]["frac"])')" | tee -a $OUT
done
