set -o pipefail
mkdir -p gpurun_out
run() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?; echo "rc=$rc :: $*"; tail -4 "$log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
run 300 gpurun_out/r05b_colmajor.log python -m pytest tests/test_gpu_colmajor.py -q -x
run 300 gpurun_out/r05b_nan.log python -m pytest tests/test_gpu_parity.py tests/test_float32.py -q -x -k "nan or dot_norm or reductions_and_updates"
run 600 gpurun_out/r05b_mr.log python -m pytest tests/test_gpu_multirank.py -q -x -k "exchange[5] or expired or bench"
HPCLA_SPMM_COLS_MULT=8 run 600 gpurun_out/r05b_sprand8.log python bench.py --workload sprand_spmm --steps 10 --warmup 5
run 600 gpurun_out/r05b_spmm2d.log python bench.py --workload poisson2d_spmm --steps 20 --warmup 5
