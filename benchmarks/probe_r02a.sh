#!/bin/bash
# r02a: feasibility probes (IPC push between two processes on one GPU; RCCL with a duplicate GPU)
export HSA_ENABLE_IPC_MODE_LEGACY=0
for alloc in 2 1 0; do
  d=$(mktemp -d)
  timeout -k 5 120 benchmarks/_build/probe_ipc_push 0 $d $alloc 4096 300 > gpurun_out/r02a_ipc_a${alloc}_r0.log 2>&1 &
  p0=$!
  timeout -k 5 120 benchmarks/_build/probe_ipc_push 1 $d $alloc 4096 300 > gpurun_out/r02a_ipc_a${alloc}_r1.log 2>&1
  rc1=$?
  wait $p0; rc0=$?
  echo "alloc=$alloc rc0=$rc0 rc1=$rc1"; cat gpurun_out/r02a_ipc_a${alloc}_r0.log gpurun_out/r02a_ipc_a${alloc}_r1.log
  if [ $rc0 -eq 124 ] || [ $rc1 -eq 124 ]; then echo "timed out - stopping"; exit 1; fi
done
# larger payload (3-D plane: 262144 doubles)
d=$(mktemp -d)
timeout -k 5 120 benchmarks/_build/probe_ipc_push 0 $d 2 262144 100 > gpurun_out/r02a_ipc_big_r0.log 2>&1 &
p0=$!
timeout -k 5 120 benchmarks/_build/probe_ipc_push 1 $d 2 262144 100 > gpurun_out/r02a_ipc_big_r1.log 2>&1
wait $p0
cat gpurun_out/r02a_ipc_big_r0.log gpurun_out/r02a_ipc_big_r1.log
timeout -k 5 180 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29811 benchmarks/probe_rccl_same_gpu.py > gpurun_out/r02a_rccl_same_gpu.log 2>&1
echo "rccl probe rc=$?"; tail -5 gpurun_out/r02a_rccl_same_gpu.log
