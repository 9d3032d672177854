#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 HPCLA_ALLOW_SHARED_GPU=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --size 2048 --strong-size 2048 --no-extras > gpurun_out/r02z_torchrun.log 2>&1; echo "rc=$?"
grep -c '^{"metric"' gpurun_out/r02z_torchrun.log; grep '^{"metric"' gpurun_out/r02z_torchrun.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["n_gpus"], r["verified_vs_closed_form"], r["config"]["parallelism"], r["strong_scaling"]["speedup_vs_n1"])'
