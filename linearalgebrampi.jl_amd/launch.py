"""Self-launcher: one process per GPU, started by the program itself.

The reference's distributed entry points launch themselves (test/runtests.jl:16-35 shells out
``mpiexec -n N julia ...``).  This is the DeviceROCm twin: ``spawn_ranks`` starts N copies of a script
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (the environment contract of
``torch.distributed.run``, so the same script also runs under torchrun), forwards rank 0's standard
output, and fails if any rank fails.

stdlib only, and it must stay that way: the PARENT never imports torch or loads the HIP library --
a process that has initialised the GPU must not fork/exec others, and the children need the devices.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    """A TCP port that is free right now on 127.0.0.1 (picked at run time: fixed ports collide when two
    sessions share a host)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def already_launched() -> bool:
    """True inside a rank started by torchrun / spawn_ranks (the launcher's environment is present)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def spawn_ranks(argv: Sequence[str], nranks: int, env_extra: Optional[Dict[str, str]] = None,
                timeout: Optional[float] = None, python: Optional[str] = None,
                forward_rank0_stdout: bool = True) -> int:
    """Run ``python argv...`` as ``nranks`` processes and wait.  Rank 0's stdout is this process's stdout
    (the one JSON line of a benchmark); the other ranks' stdout goes to stderr; stderr is shared.
    Returns 0 if every rank exited 0, else the first non-zero exit code; when one rank fails or the
    timeout expires the remaining ranks are terminated (exact PIDs, never by pattern)."""
    if nranks < 1:
        raise ValueError("nranks must be >= 1")
    port = free_port()
    procs: List[subprocess.Popen] = []
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: needed by RCCL and by the peer windows
    if env_extra:
        base.update(env_extra)
    for r in range(nranks):
        env = dict(base)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks),
                    "LOCAL_WORLD_SIZE": str(nranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HPCLA_SELF_LAUNCHED": "1"})
        out = None if (r == 0 and forward_rank0_stdout) else sys.stderr
        procs.append(subprocess.Popen([python or sys.executable, *argv], env=env, stdout=out))
    deadline = time.monotonic() + timeout if timeout else None
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
        expired = deadline is not None and time.monotonic() > deadline
        if (rc != 0 or expired) and alive:
            if expired and rc == 0:
                rc = 124
            # a rank failed (or the job overran): the others are blocked in a collective -- stop them
            grace = time.monotonic() + 5.0
            for p in alive:
                p.terminate()
            for p in alive:
                try:
                    p.wait(timeout=max(0.1, grace - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            alive = []
    return rc
