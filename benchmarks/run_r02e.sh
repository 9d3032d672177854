#!/bin/bash
# r02e: SpMM kernel diet -- tuning knobs on the three SpMM regimes (stencil; random, B in cache; random, 2.1 GB B)
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
OUT=gpurun_out/r02e_spmm_tune.log; : > $OUT
for vu in 2 4; do for ch in 512 1536; do
  export HPCLA_SPMM_VU=$vu HPCLA_SPMM_CHUNK=$ch
  step 300 gpurun_out/r02e_tmp.log python bench.py --workload poisson2d_spmm --steps 30 --warmup 3 || { tail -5 gpurun_out/r02e_tmp.log; exit 1; }
  echo "VU=$vu CHUNK=$ch poisson2d_spmm: $(tail -1 gpurun_out/r02e_tmp.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], "ms", r["value"], "GFLOP/s frac", r["roofline"]["frac"])')" | tee -a $OUT
  step 300 gpurun_out/r02e_tmp.log python bench.py --workload sprand_spmm --steps 20 --warmup 3 || { tail -5 gpurun_out/r02e_tmp.log; exit 1; }
  echo "VU=$vu CHUNK=$ch sprand_spmm x1:  $(tail -1 gpurun_out/r02e_tmp.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], "ms", r["value"], "GFLOP/s frac", r["roofline"]["frac"], "gather GB/s", r["roofline"]["gather_gbs"], "setup_s", r["setup_s"])')" | tee -a $OUT
  HPCLA_SPMM_COLS_MULT=8 step 300 gpurun_out/r02e_tmp.log python bench.py --workload sprand_spmm --steps 20 --warmup 3 || { tail -5 gpurun_out/r02e_tmp.log; exit 1; }
  echo "VU=$vu CHUNK=$ch sprand_spmm x8:  $(tail -1 gpurun_out/r02e_tmp.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], "ms", r["value"], "GFLOP/s frac", r["roofline"]["frac"], "gather GB/s", r["roofline"]["gather_gbs"], "setup_s", r["setup_s"])')" | tee -a $OUT
done; done
unset HPCLA_SPMM_VU HPCLA_SPMM_CHUNK
step 600 gpurun_out/r02e_pytest_spmm.log python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spmm or SpMM"; tail -3 gpurun_out/r02e_pytest_spmm.log
step 900 gpurun_out/r02e_pytest_new.log python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -x -q -m gpu -s -k "full_size or config3 or config4 or packed or to_backend or golden_host_layer or device_side or widened"; tail -12 gpurun_out/r02e_pytest_new.log
