#!/bin/bash
# r03l (same script as r03g): validation of the tree as it stands: GPU suite, smoke(), the bench line at the driver's flags, and the N > 1 line under
# the DRIVER's launcher (torch.distributed.run) with two ranks on the one GPU.
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03l_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 1000 gpurun_out/r03l_pytest.log python -m pytest tests -m gpu -q
tail -4 gpurun_out/r03l_pytest.log
step 300 gpurun_out/r03l_smoke.log python -c "import __graft_entry__ as g; g.smoke()"
tail -1 gpurun_out/r03l_smoke.log
step 500 gpurun_out/r03l_bench.log python bench.py --gpus 1 --steps 20 --warmup 5
grep "^{" gpurun_out/r03l_bench.log | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['launch_ms_timed_region'])
for k,v in r['other_configs'].items():
    print(k, v.get('ms_per_step'), v.get('device_ms_per_step', v.get('device_ms_per_iter')), v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('traffic'))
print('strong', r['strong_scaling']['ms_per_step'], r['budget'])"
export HPCLA_ALLOW_SHARED_GPU=1
step 600 gpurun_out/r03l_torchrun2.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-extras
grep "^{" gpurun_out/r03l_torchrun2.log | tail -1 | cut -c1-700
