set -o pipefail
mkdir -p gpurun_out
run() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?; echo "rc=$rc :: $*"; tail -3 "$log" | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
run 300 gpurun_out/r05h_tiled_test.log python -m pytest tests/test_gpu_parity.py -q -x -k "tile_stream"
grep -q "passed" gpurun_out/r05h_tiled_test.log && ! grep -q "failed" gpurun_out/r05h_tiled_test.log || { tail -60 gpurun_out/r05h_tiled_test.log; exit 1; }
HPCLA_SPRAND_SPMV=1 HPCLA_SPMM_COLS_MULT=8 run 400 gpurun_out/r05h_sprandv8.log python bench.py --workload sprand_spmm --steps 20 --warmup 5
HPCLA_SPRAND_SPMV=1 HPCLA_SPMM_COLS_MULT=1 run 400 gpurun_out/r05h_sprandv1.log python bench.py --workload sprand_spmm --steps 20 --warmup 5
