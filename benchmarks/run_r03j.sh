#!/bin/bash
# r03j: 4 ranks on the one GPU: the whole N > 1 bench line (weak headline, strong 8192^2 + rank-0-alone point, CG, both SpMMs,
# transport comparison) -- more than two ranks through every sub-record; timings meaningless
set -o pipefail
mkdir -p gpurun_out
export HPCLA_ALLOW_SHARED_GPU=1
timeout -k 10 900 python bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r03j_bench4.log 2>&1
echo "rc=$?"
grep "bench +" gpurun_out/r03j_bench4.log | tail -16
grep "^{" gpurun_out/r03j_bench4.log | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print({k:r.get(k) for k in ('n_gpus','value','ms_per_step','halo_mode','n_ranks_rccl','peer_windows','strong_scaling_speedup_vs_n1','verified_vs_closed_form','exchange_timed_out','budget')})
print('strong', {k:r['strong_scaling'].get(k) for k in ('ms_per_step','speedup_vs_n1','verified_vs_closed_form','timed_out','halo_mode','error')})
for k,v in r['other_configs'].items():
    print(k, {kk:v.get(kk) for kk in ('ms_per_step','error','skipped','residual_last','exchange_timed_out')}, (v.get('roofline_xgmi') or {}).get('bytes_in_per_gpu_per_step'), (v.get('roofline_xgmi') or {}).get('peers_in'))
bd=r['step_breakdown_ms_max_over_ranks']; print('modes', bd.get('modes'), bd.get('timed_out'), bd.get('error'))"
grep -i "error\|traceback" gpurun_out/r03j_bench4.log | head -5
