#!/bin/bash
# r03q: long-run stress of the push protocol after round 3's changes to the wait / push / all-reduce code (2 ranks, 3000 dependent
# steps each with a window all-reduce, no host sync; 4 ranks, 1200 steps), then the new 3-rank panel-order test
export HSA_ENABLE_IPC_MODE_LEGACY=0
export HPCLA_PUSH_TIMEOUT_S=30
mkdir -p gpurun_out
for n in 2 4; do
  export STRESS_STEPS=$([ $n = 2 ] && echo 3000 || echo 1200)
  timeout -k 10 500 python -c "
import sys, importlib.util
spec = importlib.util.spec_from_file_location('l', 'linearalgebrampi.jl_amd/launch.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
sys.exit(m.spawn_ranks(['benchmarks/stress_push_epochs.py'], $n, timeout=450))" > gpurun_out/r03q_stress_$n.log 2>&1
  rc=$?; echo "ranks=$n rc=$rc"; grep -E "OK|Error|assert|differ" gpurun_out/r03q_stress_$n.log | tail -10
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; exit $rc; fi
done
timeout -k 10 500 python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "panel" > gpurun_out/r03q_pytest_panel.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03q_pytest_panel.log
true
