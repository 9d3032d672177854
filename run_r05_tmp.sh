set -o pipefail
mkdir -p gpurun_out
run() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?; echo "rc=$rc :: $*"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then tail -5 "$log"; exit $rc; fi; }
run 600 gpurun_out/r05f_slabs.log python benchmarks/bench_halo_overhead.py --slabs 8192x4096,8192x2048,8192x1024,4096x2048,4096x1024,4096x512
grep SLAB gpurun_out/r05f_slabs.log
REHEARSE_FLAGS="--steps 20 --warmup 5" ./run_gpu_checks.sh r05f rehearse6 2>&1 | tail -25 | cut -c1-300
run 900 gpurun_out/r05f_pytest.log python -m pytest tests -m gpu -q -x --durations=25
tail -40 gpurun_out/r05f_pytest.log
