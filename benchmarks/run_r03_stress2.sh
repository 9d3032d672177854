#!/bin/bash
# long-run stress of the push protocol with the straight-line kernel, once in the natural block order and once with every
# launch in XCD groups of 4 (2 ranks on the shared GPU, 3000 dependent steps each with a window all-reduce, no host sync)
export HSA_ENABLE_IPC_MODE_LEGACY=0
export HPCLA_PUSH_TIMEOUT_S=30
export STRESS_STEPS=3000
mkdir -p gpurun_out
for g in 0 4; do
  if [ $g = 0 ]; then unset HPCLA_SPMV_XCD_GROUP; else export HPCLA_SPMV_XCD_GROUP=$g; fi
  timeout -k 10 500 python -c "
import sys, importlib.util
spec = importlib.util.spec_from_file_location('l', 'linearalgebrampi.jl_amd/launch.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
sys.exit(m.spawn_ranks(['benchmarks/stress_push_epochs.py'], 2, timeout=450))" > gpurun_out/r03_stress2_g$g.log 2>&1
  rc=$?; echo "group=$g rc=$rc"; grep -E "OK|Error|assert|differ" gpurun_out/r03_stress2_g$g.log | tail -6
  if [ $rc -ne 0 ]; then tail -20 gpurun_out/r03_stress2_g$g.log; exit 1; fi
done
true
