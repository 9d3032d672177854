set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_cabi_from_c.py tests/test_gpu_multirank.py -m gpu -q -x -k "long_rows or cabi or push_transport" > gpurun_out/r06d_pytest_sel.log 2>&1; rc=$?
tail -3 gpurun_out/r06d_pytest_sel.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
./run_gpu_checks.sh r06d smoke driverbench cgtrace prof
