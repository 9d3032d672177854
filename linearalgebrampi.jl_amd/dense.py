"""HPCMatrix (dense, row-partitioned) and ``HPCSparseMatrix * HPCMatrix`` on DeviceROCm.

Reference: container + ``HPCMatrix_local`` (src/dense.jl:59-69, 125-156), ``HPCMatrix(M, backend)``
(:185-201) and the SpMM column loop (src/sparse.jl:2391-2413).  Only what SpMM touches is built.

Layout: the reference's local block is a column-major Julia ``Matrix`` (src/dense.jl:63).  On the
device the local block is stored ROW-major (a ``(rows_local, k)`` torch tensor): at k=16 one row is
exactly one 128-byte line, so the gather of a B row per stored entry of A is a single full-line
read, and ghost rows travel as contiguous ``count*k`` RCCL messages.  Column-major callers convert
with ``hpcla_transpose_f64``.

SpMM runs ONE kernel over all k columns and ONE halo exchange of ``count*k`` doubles per
neighbour (the reference: k exchanges, k kernel launches, 17 Allgathers, A streamed k times).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np

from . import _capi
from .backends import (HPCBackend, assert_backends_compatible, attach_halo_windows, comm_allgather, comm_rank,
                       comm_size)
from .partition import compute_partition_hash, uniform_partition
from .vectors import current_stream_ptr, dptr


def _torch():
    import torch
    return torch


class HPCMatrix:
    """``HPCMatrix{T,B}`` (src/dense.jl:59-69): ``row_partition``, ``col_partition``, local block
    ``A`` (device, row-major ``(rows_local, ncols)``), ``backend``."""

    def __init__(self, row_partition, col_partition, A_dev, backend: HPCBackend):
        self.structural_hash = None
        self.row_partition = np.asarray(row_partition, dtype=np.int64)
        self.col_partition = np.asarray(col_partition, dtype=np.int64)
        self.A = A_dev
        self.backend = backend

    @property
    def shape(self):
        return int(self.row_partition[-1]), int(self.A.shape[1])

    def __matmul__(self, x):
        from .vectors import HPCVector
        if isinstance(x, HPCVector):
            return dense_matvec(self, x)
        return NotImplemented

    def __mul__(self, other):
        """``A * x`` or ``A * a`` (scalar: src/dense.jl:1317-1327, 1818-1838), on the device."""
        if isinstance(other, (int, float, np.floating, np.integer)):
            return self._scaled(float(other), divide=False)
        return self.__matmul__(other)

    def __rmul__(self, a):
        if isinstance(a, (int, float, np.floating, np.integer)):
            return self._scaled(float(a), divide=False)
        return NotImplemented

    def __truediv__(self, a):
        return self._scaled(float(a), divide=True)

    def _scaled(self, a: float, divide: bool) -> "HPCMatrix":
        src = self.A if self.A.is_contiguous() else self.A.contiguous()
        out = _torch().empty_like(src)
        from .vectors import sfx_of
        if divide:
            _capi.call(f"hpcla_divide_{sfx_of(self.backend)}", dptr(src), a, dptr(out), src.numel(), current_stream_ptr())
        else:
            _capi.call(f"hpcla_scale_{sfx_of(self.backend)}", a, dptr(src), dptr(out), src.numel(), current_stream_ptr())
        return HPCMatrix(self.row_partition, self.col_partition, out, self.backend)

    def norm(self, p: float = 2) -> float:
        """``norm(A, p)`` (src/dense.jl:1399-1420): the entries as one long vector (p = 2: Frobenius)."""
        from .vectors import HPCVector, norm as vnorm
        flat = (self.A if self.A.is_contiguous() else self.A.contiguous()).view(-1)
        sizes = comm_allgather(self.backend.comm, np.array([flat.numel()], dtype=np.int64))
        part = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        return vnorm(HPCVector(compute_partition_hash(part), part, flat, self.backend), p)

    def __getitem__(self, key):
        """``A[:, k]`` (src/indexing.jl:385-393): column k (0-based here) as an HPCVector on A's row
        partition -- the operand of the reference's SpMM column loop.  Stays on the device."""
        from .vectors import HPCVector
        if not (isinstance(key, tuple) and len(key) == 2 and key[0] == slice(None) and isinstance(key[1], (int, np.integer))):
            raise TypeError("HPCMatrix indexing supports A[:, k] only")
        k, n = int(key[1]), int(self.A.shape[1])
        if k < 0 or k >= n:
            raise IndexError(f"HPCMatrix column index out of bounds: k={k}, ncols={n}")
        return HPCVector(compute_partition_hash(self.row_partition), self.row_partition,
                         self.A[:, k].contiguous(), self.backend)

    def local_values(self) -> np.ndarray:
        return self.A.detach().cpu().numpy()

    def gather(self) -> np.ndarray:
        """``Matrix(A)`` (src/HPCLinearAlgebra.jl:840-880): parity checks only."""
        from .backends import CommSerial, _dist, _host_device
        loc = self.local_values()
        comm = self.backend.comm
        if isinstance(comm, CommSerial):
            return loc
        torch = _torch()
        dev = _host_device(comm)
        sizes = np.diff(self.row_partition)
        nmax, k = int(sizes.max()), loc.shape[1]
        pad = torch.zeros((nmax, k), dtype=self.A.dtype, device=dev)
        pad[:loc.shape[0]] = torch.from_numpy(loc).to(dev)
        outs = [torch.empty_like(pad) for _ in range(comm_size(comm))]
        _dist().all_gather(outs, pad, group=comm.group)
        return np.concatenate([o[:int(s)].cpu().numpy() for o, s in zip(outs, sizes)], axis=0)

    @classmethod
    def from_global(cls, M, backend: HPCBackend, row_partition=None, col_partition=None):
        """``HPCMatrix(M, backend; row_partition, col_partition)`` (src/dense.jl:185-201)."""
        torch = _torch()
        M = np.asarray(M, dtype=backend.T)
        nranks, rank = comm_size(backend.comm), comm_rank(backend.comm)
        if row_partition is None:
            row_partition = uniform_partition(M.shape[0], nranks)
        if col_partition is None:
            col_partition = uniform_partition(M.shape[1], nranks)
        lo, hi = int(row_partition[rank]), int(row_partition[rank + 1])
        loc = torch.from_numpy(np.ascontiguousarray(M[lo:hi, :])).to(backend.torch_device)
        return cls(row_partition, col_partition, loc, backend)


def HPCMatrix_local(A_local, backend: HPCBackend, col_partition=None) -> HPCMatrix:
    """src/dense.jl:125-156: row partition inferred by Allgather of ``[nrows, ncols]``."""
    torch = _torch()
    if isinstance(A_local, np.ndarray):
        A_local = torch.from_numpy(np.ascontiguousarray(A_local, dtype=backend.T))
    from .vectors import torch_dtype_of
    A_local = A_local.to(device=backend.torch_device, dtype=torch_dtype_of(backend)).contiguous()
    nranks = comm_size(backend.comm)
    info = comm_allgather(backend.comm, np.array(list(A_local.shape), dtype=np.int64)).reshape(nranks, 2)
    if not np.all(info[:, 1] == info[0, 1]):                       # src/dense.jl:139-143
        raise ValueError("HPCMatrix_local: All ranks must have the same number of columns. "
                         f"Got column counts: {info[:, 1].tolist()}")
    row_partition = np.concatenate([[0], np.cumsum(info[:, 0])]).astype(np.int64)
    if col_partition is None:
        col_partition = uniform_partition(int(A_local.shape[1]), nranks)
    return HPCMatrix(row_partition, col_partition, A_local, backend)


# ---- dense A * x (src/dense.jl:397-658) -------------------------------------------
_dense_vector_plan_cache: Dict[tuple, object] = {}


def clear_dense_plan_cache() -> None:
    for halo in _dense_vector_plan_cache.values():
        if halo:
            _capi.call("hpcla_halo_plan_destroy", halo)
    _dense_vector_plan_cache.clear()


def _dense_vector_plan(A: HPCMatrix, x):
    """DenseMatrixVectorPlan (src/dense.jl:424-538): every rank needs the WHOLE of x, so each rank
    sends its slice to all others.  Memoized on (A's partitions, x's partition) like
    _dense_vector_plan_cache (src/dense.jl:596-606)."""
    torch = _torch()
    backend = A.backend
    rank, nranks = comm_rank(backend.comm), comm_size(backend.comm)
    key = (compute_partition_hash(A.row_partition), compute_partition_hash(A.col_partition), x.structural_hash)
    if key in _dense_vector_plan_cache:
        return _dense_vector_plan_cache[key]
    halo = ctypes.c_void_p()
    if nranks > 1:
        others = [r for r in range(nranks) if r != rank]
        n_own = int(x.partition[rank + 1] - x.partition[rank])
        sizes = np.diff(x.partition)
        send_to = [r for r in others if n_own > 0]
        recv_from = [r for r in others if sizes[r] > 0]
        send_ranks = (ctypes.c_int32 * max(len(send_to), 1))(*send_to)
        send_counts = (ctypes.c_int64 * max(len(send_to), 1))(*([n_own] * len(send_to)))
        recv_ranks = (ctypes.c_int32 * max(len(recv_from), 1))(*recv_from)
        recv_counts = (ctypes.c_int64 * max(len(recv_from), 1))(*[int(sizes[r]) for r in recv_from])
        idx = (torch.arange(n_own, dtype=torch.int64, device=backend.torch_device).repeat(len(send_to))
               if send_to else None)
        torch.cuda.current_stream().synchronize()
        _capi.check("hpcla_halo_plan_create", _capi.load().hpcla_halo_plan_create(
            ctypes.byref(halo), backend.rccl, len(send_to), send_ranks, send_counts, dptr(idx), 1,
            len(recv_from), recv_ranks, recv_counts, 1))
    _dense_vector_plan_cache[key] = halo
    return halo


def dense_matvec(A: HPCMatrix, x, y=None):
    """``A * x`` / ``mul!(y, A, x)`` for a dense row-partitioned A (src/dense.jl:614-658)."""
    from .vectors import HPCVector, f64_only
    f64_only(A.backend, "dense A*x")
    torch = _torch()
    assert_backends_compatible(A.backend, x.backend)
    backend = A.backend
    rank = comm_rank(backend.comm)
    nloc, ncols = int(A.A.shape[0]), int(A.A.shape[1])
    if int(x.partition[-1]) != ncols:
        raise ValueError(f"dimension mismatch: A has {ncols} columns, x has length {int(x.partition[-1])}")
    if y is None:
        y = HPCVector(compute_partition_hash(A.row_partition), A.row_partition,
                      torch.empty(nloc, dtype=torch.float64, device=backend.torch_device), backend)
    elif y.local_length != nloc:
        raise ValueError("mul!: y has the wrong local length")
    halo = _dense_vector_plan(A, x)
    s = current_stream_ptr()
    n_lo, n_own = int(x.partition[rank]), int(x.partition[rank + 1] - x.partition[rank])
    n_hi = ncols - n_lo - n_own
    ghost = ctypes.c_void_p()
    if halo:
        _capi.call("hpcla_halo_begin", halo, dptr(x.v), s)
        _capi.call("hpcla_halo_end", halo, s)
        ng = ctypes.c_int64()
        _capi.call("hpcla_halo_ghost_ptr", halo, ctypes.byref(ghost), ctypes.byref(ng))
    Ac = A.A if A.A.is_contiguous() else A.A.contiguous()
    x_lo = ghost if n_lo else None
    x_hi = ctypes.c_void_p(ghost.value + 8 * n_lo) if (n_hi and ghost.value) else None
    _capi.call("hpcla_gemv_rowmajor_f64", dptr(Ac), ncols, nloc, x_lo, n_lo, dptr(x.v), n_own, x_hi, n_hi,
               dptr(y.v), s)
    return y


class TransposedHPCMatrix:
    """Lazy ``transpose(A)`` of a dense HPCMatrix (``Transpose(A)``, src/dense.jl:952)."""

    def __init__(self, parent: HPCMatrix):
        self.parent = parent

    @property
    def shape(self):
        m, n = self.parent.shape
        return n, m

    def __matmul__(self, x):
        from .vectors import HPCVector
        if isinstance(x, HPCVector):
            return dense_matvec_t(self.parent, x)
        return NotImplemented

    __mul__ = __matmul__


def dense_matvec_t(A: HPCMatrix, x):
    """``transpose(A) * x`` without materialising the transpose (src/dense.jl:1210-1261).  The reference
    gathers x onto A's row partition through a DenseTransposeVectorPlan (CPU-staged), multiplies the
    local block transposed, all-reduces the ncols partial sums on the host and keeps the own column
    slice.  Here: x is aligned to ``A.row_partition`` device to device (repartition.py; a no-op when
    it already is), ``hpcla_gemv_t_rowmajor_f64`` forms the partial column sums, RCCL all-reduces them
    in place, and the own slice of ``A.col_partition`` is the result."""
    from .vectors import f64_only
    f64_only(A.backend, "transpose(A)*x for dense A")
    from .repartition import repartition_vector
    from .vectors import HPCVector
    torch = _torch()
    assert_backends_compatible(A.backend, x.backend)
    backend = A.backend
    rank = comm_rank(backend.comm)
    nloc, ncols = int(A.A.shape[0]), int(A.A.shape[1])
    if int(x.partition[-1]) != int(A.row_partition[-1]):
        raise ValueError(f"dimension mismatch: transpose(A) has {int(A.row_partition[-1])} columns, "
                         f"x has length {int(x.partition[-1])}")
    xa = repartition_vector(x, A.row_partition)
    dev = backend.torch_device
    full = torch.empty(ncols, dtype=torch.float64, device=dev)
    work = torch.empty(max(1, _capi.load().hpcla_gemv_t_work_bytes(nloc, ncols) // 8), dtype=torch.float64, device=dev)
    Ac = A.A if A.A.is_contiguous() else A.A.contiguous()
    s = current_stream_ptr()
    _capi.call("hpcla_gemv_t_rowmajor_f64", dptr(Ac), ncols, nloc, ncols, dptr(xa.v), dptr(full), dptr(work), s)
    _capi.call("hpcla_allreduce_f64", backend.rccl, dptr(full), ncols, 0, s)
    lo, hi = int(A.col_partition[rank]), int(A.col_partition[rank + 1])
    return HPCVector(compute_partition_hash(A.col_partition), A.col_partition.copy(), full[lo:hi].clone(), backend)


# width-k halo plans hang off the same key as the vector plan, plus k
_spmm_halo_cache: Dict[tuple, object] = {}


_spmm_backends: Dict[int, HPCBackend] = {}
_spmm_panel_cache: Dict[tuple, "SpmmPanelPlan"] = {}


def clear_spmm_cache() -> None:
    """Collective, like clear_plan_cache! (the ranks meet before ghost windows are unmapped)."""
    from .sparse import _quiesce
    _quiesce(list(_spmm_backends.values()))
    _spmm_backends.clear()

    def _dead(h) -> bool:
        if not h:
            return False
        flag = ctypes.c_int(0)
        _capi.call("hpcla_halo_status", h, ctypes.byref(flag))
        return bool(flag.value)
    n_dead = sum(1 for ent in _spmm_halo_cache.values() if _dead(ent[0])) + \
        sum(1 for pp in _spmm_panel_cache.values() if any(_dead(h) for h in pp.halos))
    if n_dead:
        import warnings
        warnings.warn(f"clear_spmm_cache: {n_dead} SpMM exchange plan(s) had a timed-out exchange (their results were NaN)")
    for pp in _spmm_panel_cache.values():
        for h in reversed(pp.halos):                  # chained plans before their leader (hpcla_halo_plan_chain)
            if h:
                _capi.call("hpcla_halo_plan_destroy", h)
    _spmm_panel_cache.clear()
    for h in _spmm_halo_cache.values():
        if h[0]:
            _capi.call("hpcla_halo_plan_destroy", h[0])
    _spmm_halo_cache.clear()


def spmm_pitch(A, k: int) -> int:
    """Row pitch (in values) of the row-major B / ghost / C blocks of ``A * B``: k -- or k + 1 for an ODD k >= 3 of the
    sequential Float64 product (round 6): the 16-byte vector kernel owns column pairs, so an odd k runs on the even pitch
    with the last pair's second half masked (csrc/spmm.hip; 5-point matrix x 15: 0.946 ms on the one-column-per-lane kernel
    the odd pitch falls to, against 0.475 for 16).  Ghost rows travel on the same pitch (one padding double per row).  A
    function of (element type, k, HPCLA_SPMM_ORDER) only, so every rank of a communicator computes the same width."""
    if k >= 3 and (k & 1) and A.T == np.dtype(np.float64) and spmm_order() != "panel":
        return k + 1
    return k


def _rows_on_pitch(M, pitch: int):
    """The rows of the (n, k) block ``M`` as a device tensor whose row stride is exactly ``pitch`` and whose base is 16-byte
    aligned: M itself when it already is (the products of this module allocate their results that way), else one copy."""
    torch = _torch()
    n, k = int(M.shape[0]), int(M.shape[1])
    if pitch == k:
        return M.contiguous()
    if n > 0 and M.stride(1) == 1 and M.stride(0) == pitch and M.data_ptr() % 16 == 0:
        return M
    buf = torch.empty((n, pitch), dtype=M.dtype, device=M.device)
    buf[:, :k] = M
    return buf


def _spmm_plan(A, B: HPCMatrix, width=None):
    """(vector plan, exchange entry) for ``A * B``; ``width`` = values per exchanged row (default: ``spmm_pitch(A, k)`` --
    k, or k + 1 for an odd k; a caller that drives the exchange from column-major blocks passes k). the vector plan for (A, B's row partition) provides the
    neighbour lists, the split colval and the blocks; the width-k halo plan hangs off the same key plus k.
    Entry = (halo handle | None, interior blocks, boundary blocks, send_idx, colval_split, ghost pointer, n_ghost rows,
    send rows, peers, lists, entry_is_i64).  ``entry_is_i64``: index type of THIS entry's kernel arrays -- the vector
    plan's (Int32 for a narrowed Int64 matrix, sparse.can_narrow_indices) unless whole slices made the ghost row space
    outgrow Int32, in which case the entry falls back to the matrix's own Int64 arrays.  Collective on first use."""
    from .sparse import get_vector_plan
    from .vectors import HPCVector
    torch = _torch()
    backend = A.backend
    dev = backend.torch_device
    k = int(B.A.shape[1])
    kw = spmm_pitch(A, k) if width is None else int(width)     # values per exchanged row: k, or the padded pitch of an odd k
    probe = HPCVector(compute_partition_hash(B.row_partition), B.row_partition,
                      B.A[:, 0] if k > 0 else torch.empty(0, dtype=torch.float64, device=dev), backend)
    plan = get_vector_plan(A, probe)
    if int(B.A.shape[0]) != plan.n_own:
        raise ValueError("A*B: B's local rows do not match its row partition")
    nranks = comm_size(backend.comm)
    if nranks == 1 or k == 0:
        return plan, None
    s = current_stream_ptr()
    key = (A._ensure_hash(), probe.structural_hash, kw, plan.is_i64, str(A.T))
    _spmm_backends[id(backend)] = backend
    ent = _spmm_halo_cache.get(key)
    if ent is None:
        # plan time, collective (every rank, with or without neighbours): who gets whole slices
        from .backends import comm_alltoall_counts
        from .sparse import can_narrow_indices, split_colval, whole_slice_lists, whole_slice_wishes
        h = plan.host
        wish = whole_slice_wishes(h, B.row_partition, nranks)
        granted = comm_alltoall_counts(backend.comm, wish)
        if not plan.has_halo:
            attach_halo_windows(backend, None)          # collective: the other ranks' plans are attaching
            ent = _spmm_halo_cache[key] = (None, None, None, None, plan.colval_split, None, 0, 0, (0, 0), None, plan.is_i64)
        else:
            send_indices, recv_counts_l, cmap = whole_slice_lists(h, A.col_indices, B.row_partition, wish, granted)
            # index type of this entry's kernel arrays: the vector plan's, unless whole slices outgrow a narrowed plan
            ent_i64 = plan.is_i64
            if plan.narrowed and not can_narrow_indices(A.nnz, A.nrows_local, plan.n_own, sum(recv_counts_l)):
                ent_i64 = True
            sfx = "i64" if ent_i64 else "i32"
            Ti = np.int64 if ent_i64 else np.int32
            n_send, n_recv = len(h.send_rank_ids), len(h.recv_rank_ids)
            send_ranks = (ctypes.c_int32 * max(n_send, 1))(*h.send_rank_ids)
            send_counts = (ctypes.c_int64 * max(n_send, 1))(*[len(i) for i in send_indices])
            recv_ranks = (ctypes.c_int32 * max(n_recv, 1))(*h.recv_rank_ids)
            recv_counts = (ctypes.c_int64 * max(n_recv, 1))(*recv_counts_l)
            send_idx = (torch.from_numpy(np.concatenate(send_indices).astype(Ti)).to(dev)
                        if n_send else None)
            if plan.n_own + sum(recv_counts_l) > np.iinfo(Ti).max:
                raise OverflowError("split column space does not fit the index type")
            if wish.any() or ent_i64 != plan.is_i64:
                # ghost positions (or the index type) differ from the vector plan's: a split colval copy of its own
                colval_split, _ = split_colval(A, cmap, to_i32=not ent_i64)
            else:
                colval_split = plan.colval_split
            halo = ctypes.c_void_p()
            torch.cuda.current_stream().synchronize()
            # SINGLE_BUFFER: this plan is driven through halo_begin / halo_end and its consumers take the ghost
            # pointer from the host while the exchange is still in flight -- a width-1 plan (one-column B) must
            # not be double-buffered like the fused SpMV's vector plans (round-2 defect: k == 1 read the buffer
            # of the PREVIOUS exchange)
            _capi.check("hpcla_halo_plan_create_ex", _capi.load().hpcla_halo_plan_create_ex(
                ctypes.byref(halo), backend.rccl, n_send, send_ranks, send_counts, dptr(send_idx),
                1 if ent_i64 else 0, n_recv, recv_ranks, recv_counts, kw, _capi.HALO_SINGLE_BUFFER))
            bp = np.asarray(B.row_partition, dtype=np.int64)
            wprobe = (plan.n_own, kw, [(r, np.arange(bp[r + 1] - bp[r]) if wish[r] else A.col_indices[perm] - bp[r])
                                     for r, perm in zip(h.recv_rank_ids, h.recv_perm)])
            attach_halo_windows(backend, halo, wprobe)  # collective: push transport when all ranks share a node
            # SpMM row blocks are smaller than SpMV row blocks: classify at SpMM granularity (the Float32 product, csrc/f32.hip,
            # runs on the SpMV's 256-row blocks)
            rpb = (_capi.load().hpcla_spmv_rows_per_block() if A.T == np.dtype(np.float32)
                   else _capi.load().hpcla_spmm_rows_per_block())
            nblk = (A.nrows_local + rpb - 1) // rpb
            flags_i = torch.empty(nblk, dtype=torch.int32, device=dev)
            _capi.call(f"hpcla_classify_blocks_{sfx}", dptr(_entry_rowptr(A, plan, ent_i64)), dptr(colval_split),
                       A.nrows_local, 0, plan.n_own, rpb, dptr(flags_i), s)
            flags = flags_i != 0
            interior = torch.nonzero(~flags).flatten().to(torch.int32).contiguous()
            boundary = torch.nonzero(flags).flatten().to(torch.int32).contiguous()
            # the ghost buffer of a single-buffered plan is a constant: fetched once, at plan time
            ghost, ng = ctypes.c_void_p(), ctypes.c_int64()
            _capi.call("hpcla_halo_ghost_ptr", halo, ctypes.byref(ghost), ctypes.byref(ng))
            peers = (sum(1 for c in recv_counts_l if c), sum(1 for i in send_indices if len(i)))
            ent = (halo, interior, boundary, send_idx, colval_split, ghost, int(ng.value),
                   int(sum(len(i) for i in send_indices)), peers,
                   {"send_indices": send_indices, "recv_counts": list(recv_counts_l), "wish": wish}, ent_i64)
            _spmm_halo_cache[key] = ent
    return plan, ent


def _entry_rowptr(A, plan, ent_i64: bool):
    """``rowptr`` for the kernels of an SpMM entry: the vector plan's (its Int32 copy when narrowed), or the matrix's
    own Int64 array when the entry could not stay narrowed."""
    return plan.rowptr_of(A) if ent_i64 == plan.is_i64 else A.rowptr_target


def spmm_order() -> str:
    """``HPCLA_SPMM_ORDER``: "sequential" (default: the reference's bits) or "panel" (exchange overlapped chunk by
    chunk; the same sums in a different order)."""
    o = os.environ.get("HPCLA_SPMM_ORDER", "sequential").strip().lower()
    if o not in ("sequential", "panel"):
        raise ValueError("HPCLA_SPMM_ORDER must be 'sequential' or 'panel'")
    return o


class SpmmPanelPlan:
    """Opt-in PANEL order of the distributed ``A * B`` (``HPCLA_SPMM_ORDER=panel``; default stays sequential).

    Why: with uniformly random columns (BASELINE config 5) no 64-row block of A is interior, so the sequential
    form is exchange + kernel (~1.8 GB per GPU ~ 1.7 ms at link rate, plus ~1.4 ms of kernel), not
    the larger of the two.  Here every sender's slice travels in ``n_chunks`` chunk-sets, one after the other on
    ONE exchange stream (chained halo plans), and the consumer multiplies panel by panel as the chunk-sets land:
    first the own columns (overlapping chunk-set 0), then the panel of chunk-set c as soon as it has arrived,
    each panel CONTINUING the sums in C (``hpcla_spmm_panel_*``, accumulate).  Every C(r, c) is therefore the
    reference's sum (src/sparse.jl:2391-2413) with its terms taken own-columns-first, then chunk by chunk --
    one running sum, never separately rounded partials; it differs from the sequential result by reassociation
    only (tests: <= 1e-12 relative and within 1e-12 * (|A||B|) componentwise, BASELINE's tolerance).

    Plan time: the split-column CSR is cut into 1 + n_chunks panels (torch index ops -- setup plumbing); the
    values are a snapshot taken through the panel permutations and refreshed when ``A.nzval`` has changed
    (torch's version counter), like the packed copy."""

    def __init__(self, A, plan, ent, k: int, B_row_partition, n_chunks: int):
        torch = _torch()
        backend = A.backend
        dev = backend.torch_device
        self.k, self.n_chunks = k, n_chunks
        self.is_i64 = plan.is_i64 if ent is None else bool(ent[10])
        self.halos, self.ghosts, self._keep, self.panels, self.vals = [], [], [], [], []
        if ent is None or ent[0] is None:
            # a rank without neighbours: nothing to cut, but the other ranks' chunk-set plans attach COLLECTIVELY
            for _ in range(n_chunks):
                attach_halo_windows(backend, None)
            return
        Ti = np.int64 if self.is_i64 else np.int32
        tdt = torch.int64 if self.is_i64 else torch.int32
        h = plan.host
        info = ent[9]
        send_indices, recv_counts = info["send_indices"], info["recv_counts"]
        colval_split = ent[4]
        n_own = plan.n_own
        from .sparse import panel_chunk_lists
        cut = lambda n, c: (n * c) // n_chunks                      # both ends of a link cut its list the same way
        send_chunks, recv_chunk_counts, chunk_of, newpos = panel_chunk_lists(send_indices, recv_counts, n_chunks)
        n_ghost = len(chunk_of)
        # one chained halo plan per chunk-set
        n_send, n_recv = len(h.send_rank_ids), len(h.recv_rank_ids)
        send_ranks = (ctypes.c_int32 * max(n_send, 1))(*h.send_rank_ids)
        recv_ranks = (ctypes.c_int32 * max(n_recv, 1))(*h.recv_rank_ids)
        bp = np.asarray(B_row_partition, dtype=np.int64)
        wish = info["wish"]
        torch.cuda.current_stream().synchronize()
        for c in range(n_chunks):
            s_lists, r_counts = send_chunks[c], recv_chunk_counts[c]
            send_counts = (ctypes.c_int64 * max(n_send, 1))(*[len(i) for i in s_lists])
            recv_cnt_c = (ctypes.c_int64 * max(n_recv, 1))(*r_counts)
            send_idx = torch.from_numpy(np.concatenate(s_lists).astype(Ti)).to(dev) if n_send else None
            halo = ctypes.c_void_p()
            _capi.check("hpcla_halo_plan_create_ex", _capi.load().hpcla_halo_plan_create_ex(
                ctypes.byref(halo), backend.rccl, n_send, send_ranks, send_counts, dptr(send_idx),
                1 if self.is_i64 else 0, n_recv, recv_ranks, recv_cnt_c, k, _capi.HALO_SINGLE_BUFFER))
            # the rows this chunk-set must deliver, for the plan's connection test (owner-local row numbers)
            seg = []
            for r, perm, cnt in zip(h.recv_rank_ids, h.recv_perm, recv_counts):
                rows_all = (np.arange(bp[r + 1] - bp[r]) if wish[r] else A.col_indices[perm] - bp[r])
                seg.append((r, rows_all[cut(cnt, c):cut(cnt, c + 1)]))
            attach_halo_windows(backend, halo, (n_own, k, seg))      # collective, like the sequential plan's
            if self.halos:
                _capi.call("hpcla_halo_plan_chain", halo, self.halos[0])
            g, ng = ctypes.c_void_p(), ctypes.c_int64()
            _capi.call("hpcla_halo_ghost_ptr", halo, ctypes.byref(g), ctypes.byref(ng))
            self.halos.append(halo)
            self.ghosts.append(g)
            self._keep.append(send_idx)
        # the panels of A: panel 0 = own columns, panel 1 + c = ghost columns that arrive with chunk-set c
        cs = colval_split.to(torch.int64)
        is_ghost = cs >= n_own
        gpos = torch.clamp(cs - n_own, min=0)
        chunk_dev = torch.from_numpy(chunk_of.astype(np.int64)).to(dev)
        newpos_dev = torch.from_numpy(newpos).to(dev)
        panel_id = torch.where(is_ghost, 1 + chunk_dev[gpos] if n_ghost else torch.zeros_like(cs), torch.zeros_like(cs))
        newcol = torch.where(is_ghost, newpos_dev[gpos] if n_ghost else cs, cs)
        counts = (A.rowptr_target[1:] - A.rowptr_target[:-1]).to(torch.int64)
        rowid = torch.repeat_interleave(torch.arange(A.nrows_local, device=dev, dtype=torch.int64), counts)
        self.panels = []                                             # (rowptr, colval, perm, nnz)
        for q in range(1 + n_chunks):
            perm = torch.nonzero(panel_id == q).flatten()            # ascending: stored order survives inside a row
            rp = torch.zeros(A.nrows_local + 1, dtype=torch.int64, device=dev)
            if perm.numel():
                torch.cumsum(torch.bincount(rowid[perm], minlength=A.nrows_local), 0, out=rp[1:])
            self.panels.append((rp.to(tdt).contiguous(), newcol[perm].to(tdt).contiguous(), perm.to(tdt).contiguous(),
                                int(perm.numel())))
        del cs, is_ghost, gpos, panel_id, newcol, rowid, counts
        self.vals = [torch.empty(pn[3], dtype=torch.float64, device=dev) for pn in self.panels]
        self._vals_version = None
        self.n_own = n_own

    def refresh_values(self, A) -> None:
        ver = (A.nzval.data_ptr(), A.nzval._version)
        if ver == self._vals_version:
            return
        sfx = "i64" if self.is_i64 else "i32"
        s = current_stream_ptr()
        for (rp, cv, perm, n), v in zip(self.panels, self.vals):
            if n:
                _capi.call(f"hpcla_gather_f64_{sfx}", dptr(A.nzval), dptr(perm), None, dptr(v), n, 0, s)
        self._vals_version = ver

    def multiply(self, A, Bc, C, plan=None) -> None:
        sfx = "i64" if self.is_i64 else "i32"
        s = current_stream_ptr()
        k = self.k
        if not self.halos:                                           # no neighbours: every column is owned
            psfx = "i64" if plan.is_i64 else "i32"
            _capi.call(f"hpcla_spmm_csr_f64_{psfx}", dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval),
                       dptr(Bc), k, _capi.LAYOUT_ROW, dptr(C), k, _capi.LAYOUT_ROW, A.nrows_local, A.nnz, k, 0, s)
            return
        self.refresh_values(A)
        for halo in self.halos:                                      # chunk-sets leave in order on one exchange stream
            _capi.call("hpcla_halo_begin", halo, dptr(Bc), s)
        rp, cv, _, n = self.panels[0]                                # own columns: overlaps chunk-set 0
        _capi.call(f"hpcla_spmm_panel_f64_{sfx}", dptr(rp), dptr(cv), dptr(self.vals[0]), dptr(Bc), k, None, k,
                   self.n_own, dptr(C), k, A.nrows_local, n, k, 0, 0, s)
        for c, halo in enumerate(self.halos):
            _capi.call("hpcla_halo_end", halo, s)
            rp, cv, _, n = self.panels[1 + c]
            # every column of this panel is a position in chunk-set c's ghost buffer (n_own = 0)
            _capi.call(f"hpcla_spmm_panel_f64_{sfx}", dptr(rp), dptr(cv), dptr(self.vals[1 + c]), dptr(Bc), k,
                       self.ghosts[c], k, 0, dptr(C), k, A.nrows_local, n, k, 0, 1, s)


def _spmm_panel_plan(A, B, plan, ent) -> SpmmPanelPlan:
    k = int(B.A.shape[1])
    n_chunks = max(1, int(os.environ.get("HPCLA_SPMM_PANELS", "4")))
    key = (A._ensure_hash(), compute_partition_hash(B.row_partition), k, n_chunks, plan.is_i64)
    pp = _spmm_panel_cache.get(key)
    if pp is None:
        pp = _spmm_panel_cache[key] = SpmmPanelPlan(A, plan, ent, k, B.row_partition, n_chunks)
    return pp


def spmm_exchange_bytes(A, B: HPCMatrix):
    """(bytes received, bytes sent, peers received from, peers sent to) by THIS rank per ``A * B``: the ghost rows of
    B that cross xGMI (8k bytes per row) -- the communication side of config 5's roofline."""
    k = int(B.A.shape[1])
    _, ent = _spmm_plan(A, B)
    if ent is None or ent[0] is None:
        return 0, 0, 0, 0
    kw = spmm_pitch(A, k)                      # what travels per row: k values, or the padded pitch of an odd k
    return ent[6] * kw * 8, ent[7] * kw * 8, ent[8][0], ent[8][1]


def _spmm_apply_order(plan, rowptr, k: int) -> None:
    """The library keeps ONE SpMM block-order hint per rowptr array (csrc/spmm.hip g_mm_order) while the host measures
    one per (k, rowptr): before a product whose (k, rowptr) has a measured order, put THAT order in force if the last
    measurement or product on this structure left another one (a later k's tuning silently overwrote the order
    earlier k's launches ran under).  Every order is a bijection of the row blocks: results never depend on it."""
    cache = plan.__dict__.get("_spmm_order")
    if not cache:
        return
    ptr = rowptr.data_ptr()
    want = cache.get((k, ptr))
    in_force = plan.__dict__.setdefault("_spmm_order_in_force", {})
    if want is not None and in_force.get(ptr) != want:
        _capi.call("hpcla_spmm_block_order_hint", ptr, 0 if want <= 1 else want)
        in_force[ptr] = want


def _spmm_block_order(A, plan, rowptr, colval_split, is_i64, Bc, ghost, C, k, blocks, pitch=None) -> int:
    """Block order of the SpMM launches over this structure, MEASURED once per (plan, k) after the first product
    (``hpcla_spmm_tune_block_order_*``: the plan's own launch -- contiguous, or the larger of its two block lists --
    under natural / 16 / 64 / 256-block XCD groups; every timed launch rewrites C with the same complete product).
    ``HPCLA_SPMM_BLOCK_ORDER`` = auto (default) | natural | <G>.  Returns the group (1 = natural).  The hint is keyed
    by the rowptr array the kernels read (the matrix's, or the narrowed plan's copy)."""
    cache = plan.__dict__.setdefault("_spmm_order", {})
    key = (k, rowptr.data_ptr())
    if key in cache:
        return cache[key]
    pitch = k if pitch is None else pitch      # row pitch of Bc / the ghost segment / C (k + 1 for an odd k)
    want = os.environ.get("HPCLA_SPMM_BLOCK_ORDER", "auto").strip().lower()
    group = 1
    if want.isdigit():
        group = max(1, int(want))
        _capi.call("hpcla_spmm_block_order_hint", rowptr.data_ptr(), group)
    elif want != "natural":
        sfx = "i64" if is_i64 else "i32"
        chosen = ctypes.c_int(1)
        try:
            _capi.call(f"hpcla_spmm_tune_block_order_f64_{sfx}", dptr(rowptr), dptr(colval_split), dptr(A.nzval),
                       dptr(Bc), pitch, ghost, pitch, plan.n_own, dptr(C), pitch, A.nrows_local, A.nnz, k, 0,
                       dptr(blocks) if blocks is not None else None, int(blocks.numel()) if blocks is not None else 0,
                       current_stream_ptr(), ctypes.byref(chosen))
            group = int(chosen.value)
        except _capi.HPCLAError as exc:            # an optional performance step must not take A*B down with it
            import sys
            sys.stderr.write(f"hpcla: SpMM block-order measurement failed ({exc}); natural order\n")
    cache[key] = group
    plan.__dict__.setdefault("_spmm_order_in_force", {})[rowptr.data_ptr()] = group     # what the library holds now
    if group > 1:
        import weakref
        from .sparse import _unhint_spmm_block_order
        owner = plan if plan.narrowed else A
        keep = owner.__dict__.setdefault("_spmm_order_finalizers", {})
        if rowptr.data_ptr() not in keep:
            keep[rowptr.data_ptr()] = weakref.finalize(owner, _unhint_spmm_block_order, rowptr.data_ptr())
    return group


RUNS_MIN_FIT = 0.99            # use the run-tile kernel when at least this share of the 64-row blocks fits its tile


def _spmm_runs(A, plan, rowptr, colval_split, is_i64):
    """Run descriptors of the SpMM row blocks for this (structure, split column space), built ONCE (plan time,
    ``hpcla_spmm_runs_build_*``: sorts every 64-row block's columns on the device and cuts them into <= 4 contiguous
    runs), or None when the run-tile kernel should not be used: fewer than RUNS_MIN_FIT of the blocks fit (unstructured
    columns: config 5), or ``HPCLA_SPMM_RUNS=0``.  Cached on the vector plan, keyed by the colval array it describes."""
    cache = plan.__dict__.setdefault("_spmm_runs", {})
    key = (colval_split.data_ptr(), rowptr.data_ptr())
    if key in cache:
        return cache[key]
    desc = None
    if os.environ.get("HPCLA_SPMM_RUNS", "1").strip().lower() not in ("0", "off", "false", "no") and A.nrows_local > 0 and A.nnz > 0:
        torch = _torch()
        nb = (A.nrows_local + 63) // 64
        buf = torch.empty(_capi.load().hpcla_spmm_runs_desc_bytes(A.nrows_local), dtype=torch.uint8, device=A.backend.torch_device)
        n_fit = ctypes.c_int64(0)
        sfx = "i64" if is_i64 else "i32"
        try:
            _capi.call(f"hpcla_spmm_runs_build_{sfx}", dptr(rowptr), dptr(colval_split), A.nrows_local, A.nnz, 0, plan.n_own,
                       dptr(buf), ctypes.byref(n_fit), current_stream_ptr())
            if n_fit.value >= RUNS_MIN_FIT * nb:
                desc = buf
        except _capi.HPCLAError as exc:            # an optional performance step must not take A*B down with it
            import sys
            sys.stderr.write(f"hpcla: SpMM run descriptors not built ({exc}); gather kernel stays\n")
        cache[(key, "fit")] = (int(n_fit.value), nb)
    cache[key] = desc
    return desc


def spmm_runs_fit_of(A, B: HPCMatrix):
    """(blocks that fit the run tile, blocks) of the plan for ``A * B``, or None before the first k = 16 product."""
    plan, ent = _spmm_plan(A, B)
    for k_, v in plan.__dict__.get("_spmm_runs", {}).items():
        if isinstance(k_, tuple) and len(k_) == 2 and k_[1] == "fit":
            return v
    return None


def spmm_block_order_of(A, B: HPCMatrix) -> int:
    """The block-order group the plan measured for ``A * B`` (1 = natural; 0 = not measured yet)."""
    plan, ent = _spmm_plan(A, B)
    k = int(B.A.shape[1])
    rowptr = plan.rowptr_of(A) if (ent is None or ent[0] is None) else _entry_rowptr(A, plan, bool(ent[10]))
    return plan.__dict__.get("_spmm_order", {}).get((k, rowptr.data_ptr()), 0)


def _spmm_f32(A, plan, ent, Bc, C, k: int, s) -> None:
    """The Float32 product (csrc/f32.hip): the same exchange entry and block lists as the Float64 one; the exchange widens
    the B rows it sends into a staging block, the boundary blocks narrow the ghost rows they gather."""
    if ent is None or ent[0] is None:
        sfx = "i64" if plan.is_i64 else "i32"
        _capi.call(f"hpcla_spmm_split_f32_{sfx}", dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval), dptr(Bc), k,
                   None, k, plan.n_own, dptr(C), k, A.nrows_local, A.nnz, k, 0, None, 0, s)
        return
    halo, interior, boundary, _, colval_split, ghost = ent[:6]
    sfx = "i64" if ent[10] else "i32"
    rowptr = _entry_rowptr(A, plan, bool(ent[10]))

    def blocks_launch(blocks, g):
        _capi.call(f"hpcla_spmm_split_f32_{sfx}", dptr(rowptr), dptr(colval_split), dptr(A.nzval), dptr(Bc), k, g, k,
                   plan.n_own, dptr(C), k, A.nrows_local, A.nnz, k, 0, dptr(blocks), int(blocks.numel()), s)
    _capi.call("hpcla_halo_begin_f32", halo, dptr(Bc), dptr(plan.stage_f32(plan.n_own * k)), s)
    if interior.numel():
        blocks_launch(interior, None)
    _capi.call("hpcla_halo_end", halo, s)
    if boundary.numel():
        blocks_launch(boundary, ghost)


def spmm(A, B: HPCMatrix) -> HPCMatrix:
    """``A * B`` (src/sparse.jl:2391-2413): result has A's row partition and B's backend."""
    torch = _torch()
    assert_backends_compatible(A.backend, B.backend)
    backend = A.backend
    dev = backend.torch_device
    k = int(B.A.shape[1])
    plan, ent = _spmm_plan(A, B)
    # row pitch of B's rows, the ghost rows and C: k, or k + 1 for an odd k (spmm_pitch; the result's block is then the
    # (rows, k) view of a (rows, k + 1) buffer, which the next product takes as it is)
    kw = spmm_pitch(A, k)
    Cbuf = torch.empty((A.nrows_local, kw), dtype=B.A.dtype, device=dev)
    C = Cbuf if kw == k else Cbuf[:, :k]
    out = HPCMatrix(plan.result_partition, uniform_partition(k, comm_size(backend.comm)), C, backend)
    if k == 0:                       # (a rank without local rows still takes part in the exchange below)
        return out
    s = current_stream_ptr()
    Bc = _rows_on_pitch(B.A, kw)
    if A.T == np.dtype(np.float32):
        _spmm_f32(A, plan, ent, Bc, C, k, s)
        return out
    if ent is not None and spmm_order() == "panel":
        # COLLECTIVE choice: every rank must run the same order (the chunk-set plans are separate exchanges, and a
        # rank without neighbours still takes part in their collective attach)
        _spmm_panel_plan(A, B, plan, ent).multiply(A, Bc, C, plan)
        return out
    if ent is None or ent[0] is None:
        # every column owned: split indices == offsets into B's local rows
        sfx = "i64" if plan.is_i64 else "i32"
        runs = _spmm_runs(A, plan, plan.rowptr_of(A), plan.colval_split, plan.is_i64) if k == 16 else None
        if runs is not None:
            # banded / stencil structure: the blocks' B rows are staged as contiguous runs (csrc/spmm.hip, RUN TILES)
            _capi.call(f"hpcla_spmm_runs_k16_f64_{sfx}", dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval),
                       dptr(Bc), None, plan.n_own, dptr(C), A.nrows_local, A.nnz, 0, dptr(runs), None, 0, s)
            return out
        _spmm_apply_order(plan, plan.rowptr_of(A), k)
        _capi.call(f"hpcla_spmm_csr_f64_{sfx}", dptr(plan.rowptr_of(A)), dptr(plan.colval_split),
                   dptr(A.nzval), dptr(Bc), kw, _capi.LAYOUT_ROW, dptr(Cbuf), kw, _capi.LAYOUT_ROW,
                   A.nrows_local, A.nnz, k, 0, s)
        _spmm_block_order(A, plan, plan.rowptr_of(A), plan.colval_split, plan.is_i64, Bc, None, Cbuf, k, None, kw)
        return out
    halo, interior, boundary, _, colval_split, ghost = ent[:6]
    sfx = "i64" if ent[10] else "i32"
    rowptr = _entry_rowptr(A, plan, bool(ent[10]))
    runs = _spmm_runs(A, plan, rowptr, colval_split, bool(ent[10])) if (k == 16 and ghost) else None

    def blocks_launch(blocks):
        if runs is not None:
            _capi.call(f"hpcla_spmm_runs_k16_f64_{sfx}", dptr(rowptr), dptr(colval_split), dptr(A.nzval), dptr(Bc), ghost,
                       plan.n_own, dptr(C), A.nrows_local, A.nnz, 0, dptr(runs), dptr(blocks), int(blocks.numel()), s)
        else:
            _capi.call(f"hpcla_spmm_split_f64_{sfx}", dptr(rowptr), dptr(colval_split),
                       dptr(A.nzval), dptr(Bc), kw, ghost, kw, plan.n_own, dptr(Cbuf), kw, A.nrows_local,
                       A.nnz, k, 0, dptr(blocks), int(blocks.numel()), s)
    if runs is None:
        _spmm_apply_order(plan, rowptr, k)
    _capi.call("hpcla_halo_begin", halo, dptr(Bc), s)
    if interior.numel():
        blocks_launch(interior)
    _capi.call("hpcla_halo_end", halo, s)
    if boundary.numel():
        blocks_launch(boundary)
    if runs is not None:
        return out
    # (local launches only -- no exchange: the ranks need not agree on the order, every order is a bijection)
    _spmm_block_order(A, plan, rowptr, colval_split, bool(ent[10]), Bc, ghost, Cbuf, k,
                      boundary if boundary.numel() >= interior.numel() else interior, kw)
    return out
