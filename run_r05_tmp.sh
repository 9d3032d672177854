set -o pipefail
mkdir -p gpurun_out
run() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?; echo "rc=$rc :: $*"; tail -2 "$log" | cut -c1-400; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
LIB=linearalgebrampi.jl_amd/libhpcla_rocm.so
cp $LIB /tmp/new.so
run 600 gpurun_out/r05e_parity.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -q -x -k "spmv or sweep"
run 400 gpurun_out/r05e_arrow.log python benchmarks/bench_arrow.py
HEAD="--steps 100 --warmup 20 --no-cpu-baseline --no-strong --no-extras --no-packed"
for i in 1 2 3; do
  for v in new prev; do
    if [ $v = prev ]; then cp benchmarks/_build/libhpcla_rocm_prev.so $LIB; else cp /tmp/new.so $LIB; fi
    timeout -k 10 200 python bench.py $HEAD 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v headline', r['ms_per_step'], r['roofline']['launch_ms_timed_region'], r['roofline']['block_order_group'])" | tee -a gpurun_out/r05e_ab.log
    timeout -k 10 200 python bench.py --workload poisson3d_cg --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cg', r['ms_per_step'], r['device_ms_per_iter'])" | tee -a gpurun_out/r05e_ab.log
  done
done
cp /tmp/new.so $LIB
