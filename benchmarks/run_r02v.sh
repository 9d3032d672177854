#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4; do
  export STRESS_STEPS=$([ $n = 2 ] && echo 3000 || echo 1200)
  timeout -k 10 500 python -c "
import sys, importlib.util
spec = importlib.util.spec_from_file_location('l', 'linearalgebrampi.jl_amd/launch.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
sys.exit(m.spawn_ranks(['benchmarks/stress_push_epochs.py'], $n, timeout=450))" > gpurun_out/r02v_stress_$n.log 2>&1
  echo "ranks=$n rc=$?"; grep -E "OK|Error|assert|differ" gpurun_out/r02v_stress_$n.log | tail -10
done
