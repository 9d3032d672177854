#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
HPCLA_SPMM_LINE=2 step 600 gpurun_out/r02w_pytest.log python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spmm or SpMM" || { tail -40 gpurun_out/r02w_pytest.log; exit 1; }
tail -2 gpurun_out/r02w_pytest.log
OUT=gpurun_out/r02w_spmm_line.log; : > $OUT
for T in 0 2 3 0 2 3; do
  export HPCLA_SPMM_LINE=$T
  step 300 gpurun_out/r02w_tmp.log python bench.py --workload poisson2d_spmm --steps 30 --warmup 3 || { tail -5 gpurun_out/r02w_tmp.log; exit 1; }
  echo "LINE=$T poisson2d_spmm: $(tail -1 gpurun_out/r02w_tmp.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], "ms", r["value"], "GFLOP/s frac", r["roofline"]["frac"])')" | tee -a $OUT
  HPCLA_SPMM_COLS_MULT=8 step 300 gpurun_out/r02w_tmp.log python bench.py --workload sprand_spmm --steps 20 --warmup 3 || { tail -5 gpurun_out/r02w_tmp.log; exit 1; }
  echo "LINE=$T sprand_spmm x8:  $(tail -1 gpurun_out/r02w_tmp.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], "ms frac", r["roofline"]["frac"], "gather GB/s", r["roofline"]["gather_gbs"])')" | tee -a $OUT
done
