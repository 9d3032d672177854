#!/bin/bash
# r03s: 2-rank rehearsal of the N > 1 line with the extra panel-order SpMM record, then the validation of the tree
set -o pipefail
mkdir -p gpurun_out
HPCLA_ALLOW_SHARED_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03s_bench2.log 2>&1; echo "bench2 rc=$?"
grep "bench +" gpurun_out/r03s_bench2.log | tail -12
grep "^{" gpurun_out/r03s_bench2.log | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
for k,v in r['other_configs'].items():
    x=v.get('roofline_xgmi') or {}
    print(k, v.get('ms_per_step'), v.get('error'), v.get('skipped'), x.get('order'), x.get('bytes_in_per_gpu_per_step'), (v.get('config') or {}).get('workload','')[-40:])"
./benchmarks/run_r03_validate.sh
