#!/bin/bash
# guarded RCCL init + breakdown-last restructure: self-exchange tests, N=1 with the breakdown self-test,
# 2-rank rehearsal, and the watchdog itself (a 1 s limit must still print the finished line)
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 600 gpurun_out/r02t_halo.log python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "halo or rccl"; tail -2 gpurun_out/r02t_halo.log
step 600 gpurun_out/r02t_b1.log env HPCLA_BENCH_BREAKDOWN_SELFTEST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r02t_b1.log") if l.startswith('{"metric"')][-1])
print("N=1", d["value"], d["roofline"]["frac"], d["verified_vs_closed_form"], d.get("step_breakdown_ms_max_over_ranks"))
PY
step 600 gpurun_out/r02t_b2.log env HPCLA_ALLOW_SHARED_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r02t_b2.log") if l.startswith('{"metric"')][-1])
print("N=2", d["value"], d["verified_vs_closed_form"], d["strong_scaling"].get("speedup_vs_n1"), d.get("step_breakdown_ms_max_over_ranks"))
PY
step 600 gpurun_out/r02t_b3.log env HPCLA_ALLOW_SHARED_GPU=1 HPCLA_BENCH_BREAKDOWN_TIMEOUT_S=0.05 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-strong
python - <<'PY'
import json
lines = [l for l in open("gpurun_out/r02t_b3.log") if l.startswith('{"metric"')]
print("watchdog: result lines", len(lines))
d = json.loads(lines[-1])
print("watchdog", d["value"], d["verified_vs_closed_form"], d.get("step_breakdown_ms_max_over_ranks"))
PY
