#!/bin/bash
# r03h: non-temporal C stores / cacheable A loads for the SpMM kernel; the bench line with the cross-check in front of the
# contract region; the new panel-accumulate test
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03h_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 400 gpurun_out/r03h_tune_spmm.log python benchmarks/tune_spmm.py --variants 100,0,21,22 --rounds 9
tail -7 gpurun_out/r03h_tune_spmm.log | cut -c1-200
step 400 gpurun_out/r03h_tune_spmm_sprand.log python benchmarks/tune_spmm.py --workload sprand --variants 100,0,21,22 --rounds 5 --reps 5
tail -7 gpurun_out/r03h_tune_spmm_sprand.log | cut -c1-200
step 300 gpurun_out/r03h_pytest_panel.log python -m pytest tests/test_gpu_parity.py -m gpu -q -k "panel_accumulate or spmm"
tail -3 gpurun_out/r03h_pytest_panel.log
for i in 1 2 3; do
  step 300 gpurun_out/r03h_bench$i.log python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-strong --no-cpu-baseline --no-packed
  grep "^{" gpurun_out/r03h_bench$i.log | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); f=r['roofline']
print('headline', r['value'], r['ms_per_step'], f['frac'], f['launch_ms_timed_region'], f['launch_ms_event_pairs'], f['launch_ms_back_to_back'], f['launch_ms_min'], f.get('launches_before_timed_region'))"
done
