#!/bin/bash
# Helper for gpurun calls: GPU tests, then bench, then (optionally) a rocprofv3 kernel trace.
# Stops at the first step that is killed by its timeout (never start a GPU step after a hang).
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-r01}
run() {  # run <timeout_s> <logfile> <cmd...>
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/${TAG}_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out/killed -- stopping"; exit $rc; fi
  return 0
}
run 900 gpurun_out/${TAG}_pytest.log python -m pytest tests -m gpu -q -x
tail -5 gpurun_out/${TAG}_pytest.log
run 600 gpurun_out/${TAG}_bench.log python bench.py --steps 100 --warmup 10
tail -3 gpurun_out/${TAG}_bench.log
