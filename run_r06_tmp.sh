set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_cabi_from_c.py -m gpu -q -x > gpurun_out/r06e_pytest_cabi.log 2>&1; rc=$?
tail -3 gpurun_out/r06e_pytest_cabi.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r06e_pytest.log 2>&1; rc=$?
tail -8 gpurun_out/r06e_pytest.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
exit 0
