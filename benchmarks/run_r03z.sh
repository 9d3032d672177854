#!/bin/bash
# r03z: block orders over more structures: config 3's 8192^2, a 1000-wide 2-D grid, a 256^3 cube, config 5's random pattern
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python benchmarks/tune_spmv_lib.py --variants "" --orders 1,8,32,64 --dims 2,3,8,1,4 --rounds 7 > gpurun_out/r03z_orders.log 2>&1; echo "rc=$?"
grep -v "^{" gpurun_out/r03z_orders.log | grep "plain\|^#" | tail -50

timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block_order or spmv or cg" > gpurun_out/r03z_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03z_pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03z_bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r03z_bench.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('config'))
for k,v in r['other_configs'].items(): print(k, v.get('ms_per_step'), v.get('device_ms_per_iter', v.get('device_ms_per_step')), (v.get('roofline') or {}).get('frac'))
print('strong', r['strong_scaling'].get('ms_per_step'), r['strong_scaling'].get('roofline',{}).get('frac'))"

true
