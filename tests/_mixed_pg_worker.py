"""Worker for tests/test_gpu_parity.py::test_host_collectives_on_mixed_gloo_nccl_group: the process group
bench.py creates at N > 1 ("cpu:gloo,cuda:nccl") with two processes; only HOST collectives are issued (the
plan-time ones), so both ranks may share the single GPU of the box -- the NCCL half is never used."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("cpu:gloo,cuda:nccl")
rank, world = dist.get_rank(), dist.get_world_size()
import hpcla_amd as hp
from hpcla_amd import backends as B
comm = hp.CommTorch()
print(rank, "backend", dist.get_backend(), "host dev", B._host_device(comm))
a2a = B.comm_alltoall_counts(comm, np.array([10 * rank + q for q in range(world)]))
assert a2a.tolist() == [10 * q + rank for q in range(world)], a2a
got = B.comm_exchange_indices(comm, [1 - rank], [np.arange(3) + 100 * rank], [1 - rank], [3])
assert got[0].tolist() == (np.arange(3) + 100 * (1 - rank)).tolist(), got
vals = B.comm_exchange_arrays(comm, [1 - rank], [np.ones(4) * rank], [1 - rank], [4], np.float64)
assert vals[0].tolist() == [float(1 - rank)] * 4
hs = B.comm_allgather_bytes(comm, bytes([rank]) * 32)
assert hs == [bytes([r]) * 32 for r in range(world)]
blob = B.comm_bcast_bytes(comm, bytes(range(128)) if rank == 0 else None, 128)
assert blob == bytes(range(128))
from oracle import oracle as orc
n = 400
xp = orc.uniform_partition(n, world)
cis = [orc.compress_columns(orc.sprand_rows(n, 0.03, int(xp[r]), int(xp[r + 1])))[0] for r in range(world)]
pl = hp.build_host_vector_plan(cis[rank], xp, comm)
want = orc.vector_plans(cis, xp)[rank]
assert pl.send_rank_ids == want.send_rank_ids and pl.recv_rank_ids == want.recv_rank_ids
print(rank, "mixed-backend host collectives OK")
dist.destroy_process_group()
