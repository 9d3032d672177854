"""Probe: can two RCCL ranks share ONE GPU?  (They cannot on NCCL: 'Duplicate GPU detected'.)
Launched by torch.distributed.run with 2 processes; both use device 0."""
import os, sys
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl")
try:
    t = torch.ones(4, device="cuda") * (dist.get_rank() + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {dist.get_rank()}: all_reduce on a shared GPU OK -> {t.tolist()}", flush=True)
except Exception as exc:
    print(f"rank {dist.get_rank()}: RCCL on a shared GPU failed: {type(exc).__name__}: {str(exc)[:400]}", flush=True)
    sys.exit(0)
