#!/bin/bash
# 2-rank and 4-rank rehearsal of the full N > 1 bench line on the shared GPU (final tree)
mkdir -p gpurun_out
export HPCLA_ALLOW_SHARED_GPU=1
for n in 4 2; do
  timeout -k 10 200 python bench.py --gpus $n --steps 5 --warmup 2 > gpurun_out/r03_reh_$n.log 2>&1; rc=$?; echo "ranks=$n rc=$rc"
  grep "bench +" gpurun_out/r03_reh_$n.log | grep -v "rank [1-9]" | cut -c1-140 | tail -14
  grep "^{" gpurun_out/r03_reh_$n.log | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value', r['value'], 'ms', r['ms_per_step'], 'verified', r['verified_vs_closed_form'], 'halo', r.get('halo_mode'), 'speedup', r.get('strong_scaling_speedup_vs_n1'), 'group', r['roofline'].get('block_order_group'))
for k,v in r['other_configs'].items(): print(' ', k, v.get('ms_per_step'), v.get('setup_s'), v.get('error'), v.get('skipped'))
print(' budget', r['budget'])"
  [ $rc -eq 0 ] || exit 1
done
true
