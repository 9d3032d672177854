#!/bin/bash
# r03f: rehearsal of the N > 1 bench line with two ranks on the one GPU (timings meaningless, every code path real):
# full line (top-level strong-scaling fields, roofline_xgmi in the SpMM record, budget record), the same under an
# artificially small budget (optional stages must be skipped, the line must still come out), and panel-order SpMM.
set -o pipefail
mkdir -p gpurun_out
export HPCLA_ALLOW_SHARED_GPU=1
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03f_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
summ() { python3 - "$1" <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print("NO JSON LINE in", sys.argv[1]); sys.exit(0)
r = json.loads(lines[-1])
keys = ("n_gpus", "value", "ms_per_step", "halo_mode", "n_ranks_rccl", "peer_windows", "strong_scaling_speedup_vs_n1", "verified_vs_closed_form", "budget")
print({k: r.get(k) for k in keys})
print("strong:", {k: r.get("strong_scaling", {}).get(k) for k in ("ms_per_step", "speedup_vs_n1", "n1_ms_per_step_rank0_alone", "skipped", "n1_skipped")})
for name, rec in (r.get("other_configs") or {}).items():
    print(name, {k: rec.get(k) for k in ("ms_per_step", "device_ms_per_step", "wall_ms_per_iter", "device_ms_per_iter", "skipped", "error")}, rec.get("roofline_xgmi"))
bd = r.get("step_breakdown_ms_max_over_ranks")
print("breakdown:", (bd if not isinstance(bd, dict) else {k: bd[k] for k in list(bd)[:6]}))
PY
}
step 900 gpurun_out/r03f_bench2.log python bench.py --gpus 2 --steps 5 --warmup 2
grep "bench +" gpurun_out/r03f_bench2.log | tail -20; summ gpurun_out/r03f_bench2.log
HPCLA_BENCH_OUTER_LIMIT_S=70 step 300 gpurun_out/r03f_bench2_small_budget.log python bench.py --gpus 2 --steps 5 --warmup 2
grep "bench +" gpurun_out/r03f_bench2_small_budget.log | tail -20; summ gpurun_out/r03f_bench2_small_budget.log
HPCLA_SPMM_ORDER=panel step 600 gpurun_out/r03f_spmm_panel.log python bench.py --gpus 2 --workload sprand_spmm --steps 5 --warmup 2
tail -1 gpurun_out/r03f_spmm_panel.log | cut -c1-1500
step 600 gpurun_out/r03f_spmm_seq.log python bench.py --gpus 2 --workload sprand_spmm --steps 5 --warmup 2
tail -1 gpurun_out/r03f_spmm_seq.log | cut -c1-600
step 600 gpurun_out/r03f_cabi.log python -m pytest tests/test_cabi_from_c.py -m gpu -q
tail -3 gpurun_out/r03f_cabi.log
