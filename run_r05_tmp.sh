set -o pipefail
mkdir -p gpurun_out
run() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?; echo "rc=$rc :: $*"; tail -3 "$log" | cut -c1-300; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
run 600 gpurun_out/r05g_tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_colmajor.py tests/test_float32.py -q -x -k "transpose or block_order or spmm or colmajor"
HPCLA_SPMM_COLS_MULT=8 run 600 gpurun_out/r05g_sprand8.log python bench.py --workload sprand_spmm --steps 10 --warmup 5
