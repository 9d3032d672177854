#!/bin/bash
# one-rank CG loop with the consumers finishing the reductions: parity tests, then A/B against the launches (separate processes, alternating)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_cabi_from_c.py -x -q -m gpu -k "cg or cabi" > gpurun_out/r03_cgp_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r03_cgp_pytest.log
[ $rc -eq 0 ] || exit 1
for i in 1 2 3; do
  for m in 0 1; do
    HPCLA_CG_SCALAR_LAUNCHES=$m timeout -k 10 200 python bench.py --workload poisson3d_cg --steps 100 > gpurun_out/r03_cgp_$m.log 2>&1
    echo "launches=$m: $(tail -1 gpurun_out/r03_cgp_$m.log | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["wall_ms_per_iter"], r["device_ms_per_iter"], r["roofline"]["frac"], r["residual_last"])')"
  done
done
true
