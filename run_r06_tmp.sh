set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r06k_summary.log
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24; do
  t0=$(date +%s)
  GPU_MAX_HW_QUEUES=32 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 120 tests/cabi/_build/cabi_ranks_threads 8 > gpurun_out/r06k_run.log 2>&1; rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - t0 )) s $(grep -c 'timed out' gpurun_out/r06k_run.log) timeouts $(grep -m1 stages gpurun_out/r06k_run.log | cut -c1-120) $(grep -m1 -i 'failed\|error' gpurun_out/r06k_run.log | cut -c1-100)" | tee -a gpurun_out/r06k_summary.log
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
done
exit 0
