"""world_size-2, 3 and 8 gloo tests of the multi-rank host path on CPU (8 = the rank count of BASELINE's multi-GPU
configurations: 7-neighbour plans of config 5's shape, slab neighbours of configs 3/4)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpcla_amd  # noqa: E402,F401  (makes hpcla_amd.launch importable)


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_vector_plan_across_processes_gloo(nranks):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    from hpcla_amd.launch import free_port
    port = free_port()                       # picked at run time: fixed ports collide when sessions share a host
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count(": OK") == nranks
