set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r06c_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r06c_pytest.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python benchmarks/bench_spmm_generic.py --ks 15,13,7,5,3 2>&1 | tee gpurun_out/r06c_spmm_odd_k.log
