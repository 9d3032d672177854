#!/bin/bash
# r03y: block-order hint: new tests, then natural vs grouped orders of the shipped library in one process; SpMM with late row bounds
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block_order or spmv or cg or spmm" > gpurun_out/r03y_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r03y_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python benchmarks/tune_spmv_lib.py --variants "" --orders 1,4,16,32,64,128 > gpurun_out/r03y_orders.log 2>&1; echo "rc=$?"
grep -v "^{" gpurun_out/r03y_orders.log | tail -50
timeout -k 10 200 python bench.py --workload poisson3d_cg --steps 100 > gpurun_out/r03y_cg.log 2>&1; echo "cg rc=$?"
tail -1 gpurun_out/r03y_cg.log | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r.get('ms_per_step'), r.get('device_ms_per_iter'), r['roofline']['frac'])"
HPCLA_BLOCK_ORDER=natural timeout -k 10 200 python bench.py --workload poisson3d_cg --steps 100 > gpurun_out/r03y_cg_nat.log 2>&1; echo "cg nat rc=$?"
tail -1 gpurun_out/r03y_cg_nat.log | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r.get('ms_per_step'), r.get('device_ms_per_iter'), r['roofline']['frac'])"
true
