"""HPCSparseMatrix on DeviceROCm: construction, VectorPlan, ``A*x`` / ``mul!`` (reference: src/sparse.jl).

Kept bit-for-bit from the reference (host side, numpy): row partition inference, the compressed
local column space ``col_indices`` (sorted unique global columns, src/sparse.jl:501) and local
``colval`` (src/sparse.jl:137-144), the VectorPlan neighbour lists (src/sparse.jl:1875-1984) and the
memoization key ``(hash(A), hash(x.partition), T, Ti, array type)`` (src/sparse.jl:1992-2001).

Replaced (device side, libhpcla_rocm through the C ABI): the execute half of the plan
(src/vectors.jl:394-463 -> GPU-resident RCCL halo, ordering chosen by HPCLA_HALO_MODE), the kernel
(src/sparse.jl:2055-2084 -> row-block stream SpMV), and the CPU ``mul!`` (src/sparse.jl:2019-2037 ->
the same device kernel writing ``y.v`` in place).

Index translation: everything here is 0-based (partitions ``[0..n]``, ``colval`` in
``0:ncols_compressed-1``); the C ABI takes ``index_base`` so Julia callers pass 1-based arrays as is.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _capi
from .backends import (CommSerial, HPCBackend, assert_backends_compatible, attach_halo_windows, comm_allgather,
                       comm_alltoall_counts, comm_barrier, comm_exchange_indices, comm_rank, comm_size,
                       require_device)
from .partition import (compute_partition_hash, compute_structural_hash, owner_of,
                        uniform_partition)
from .vectors import HPCVector, current_stream_ptr, dptr


import os as _os

_PACKED_BY_ENV = _os.environ.get("HPCLA_SPMV_PACKED", "") == "1"


def _torch():
    import torch
    return torch


def _plan_key(plan):
    """Stable identity of a cached plan: its memoization key (get_vector_plan), else the object id."""
    return getattr(plan, "key", None) or id(plan)


# =====================================================================================================
# host plan (pure numpy + comm_*; testable on CPU with gloo)
# =====================================================================================================
@dataclass
class HostVectorPlan:
    """The index half of ``VectorPlan{T,Ti,AV}`` (src/vectors.jl:229-251), 0-based."""
    send_rank_ids: List[int]
    send_indices: List[np.ndarray]      # local indices into x.v
    recv_rank_ids: List[int]
    recv_perm: List[np.ndarray]         # destination positions in `gathered`
    local_src_indices: np.ndarray
    local_dst_indices: np.ndarray
    n_gathered: int
    n_own: int                          # length of x.v on this rank


def build_host_vector_plan(col_indices: np.ndarray, x_partition: np.ndarray, comm) -> HostVectorPlan:
    """``VectorPlan(A, x)`` steps 1-6 (src/sparse.jl:1875-1953)."""
    rank, nranks = comm_rank(comm), comm_size(comm)
    col_indices = np.asarray(col_indices, dtype=np.int64)
    my_x_start = int(x_partition[rank])

    # step 1: group col_indices by owner rank in x's partition (:1888-1896).  col_indices is sorted and
    # owners are contiguous rank ranges (owner(g) = min(searchsortedlast(partition, g) - 1, nranks - 1)), so
    # each owner's entries are one contiguous run whose ends are found by nranks - 1 binary searches in
    # col_indices -- no per-element owner array, no per-element push! (1.6e7 columns at config 5)
    cuts = np.asarray(x_partition[1:nranks], dtype=np.int64)
    bounds = np.concatenate([[0], np.searchsorted(col_indices, cuts, side="left"), [len(col_indices)]]).astype(np.int64)
    send_counts = np.diff(bounds)                                           # :1899 (what I need)

    # step 2: Alltoall of counts (:1899-1900)
    recv_counts = comm_alltoall_counts(comm, send_counts)

    # step 3/4: request indices from owners, receive requests (:1908-1936)
    recv_rank_ids = [r for r in range(nranks) if send_counts[r] > 0 and r != rank]
    send_rank_ids = [r for r in range(nranks) if recv_counts[r] > 0 and r != rank]
    requests = [col_indices[bounds[r]:bounds[r + 1]] for r in recv_rank_ids]
    recv_perm = [np.arange(bounds[r], bounds[r + 1], dtype=np.int64) for r in recv_rank_ids]
    received = comm_exchange_indices(comm, recv_rank_ids, requests, send_rank_ids,
                                     [int(recv_counts[r]) for r in send_rank_ids])

    # step 5: global -> local indices for sending (:1939-1944)
    send_indices = [np.asarray(g, dtype=np.int64) - my_x_start for g in received]

    # step 6: local elements (:1947-1953)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    local_src = col_indices[lo:hi] - my_x_start
    local_dst = np.arange(lo, hi, dtype=np.int64)

    n_own = int(x_partition[rank + 1] - x_partition[rank])
    for idx in send_indices:
        if len(idx) and (idx.min() < 0 or idx.max() >= n_own):
            raise ValueError("VectorPlan: a neighbour requested an index this rank does not own")
    return HostVectorPlan(send_rank_ids, send_indices, recv_rank_ids, recv_perm, local_src,
                          local_dst, len(col_indices), n_own)


def split_column_map(plan: HostVectorPlan) -> np.ndarray:
    """compressed column c -> split column: owned columns map to their offset in x.v
    (``< n_own``), ghosts to ``n_own + position in the ghost buffer``.  The ghost buffer stores the
    receive segments in recv_rank order, which is ascending global column order."""
    m = np.empty(plan.n_gathered, dtype=np.int64)
    m[plan.local_dst_indices] = plan.local_src_indices
    off = plan.n_own
    for perm in plan.recv_perm:
        m[perm] = off + np.arange(len(perm), dtype=np.int64)
        off += len(perm)
    return m


WHOLE_SLICE_FRACTION = 0.5     # a neighbour that needs more than this share of my rows gets all of them


def whole_slice_wishes(plan: HostVectorPlan, x_partition: np.ndarray, nranks: int) -> np.ndarray:
    """Per owner rank r: 1 if this rank wants r's WHOLE slice instead of the requested rows (it needs
    more than WHOLE_SLICE_FRACTION of them).  SURVEY 8e(3): an unstructured A (config 5) touches ~98 % of
    all rows of B, where a gather-and-pack of the requested rows costs a full extra pass over them and
    a send buffer as large as the slice; whole slices go out of B in place (contiguous, no pack)."""
    sizes = np.diff(np.asarray(x_partition, dtype=np.int64))
    wish = np.zeros(nranks, dtype=np.int64)
    for r, perm in zip(plan.recv_rank_ids, plan.recv_perm):
        if len(perm) > WHOLE_SLICE_FRACTION * sizes[r]:
            wish[r] = 1
    return wish


def whole_slice_lists(plan: HostVectorPlan, col_indices: np.ndarray, x_partition: np.ndarray,
                      wish: np.ndarray, granted: np.ndarray):
    """Exchange lists with whole slices where wished.  ``wish[r]``: I receive r's whole slice;
    ``granted[q]``: q receives my whole slice (= q's wish, learnt through an Alltoall of the wishes).
    Returns (send_indices, recv_counts, split column map): the ghost segment of a whole-slice
    neighbour r holds ALL of r's rows in order, so column g sits at ``g - x_partition[r]`` in it."""
    x_partition = np.asarray(x_partition, dtype=np.int64)
    send_indices = [np.arange(plan.n_own, dtype=np.int64) if granted[q] else idx
                    for q, idx in zip(plan.send_rank_ids, plan.send_indices)]
    recv_counts = [int(x_partition[r + 1] - x_partition[r]) if wish[r] else len(perm)
                   for r, perm in zip(plan.recv_rank_ids, plan.recv_perm)]
    m = np.empty(plan.n_gathered, dtype=np.int64)
    m[plan.local_dst_indices] = plan.local_src_indices
    off = plan.n_own
    for r, perm, cnt in zip(plan.recv_rank_ids, plan.recv_perm, recv_counts):
        m[perm] = off + (col_indices[perm] - x_partition[r] if wish[r] else np.arange(len(perm), dtype=np.int64))
        off += cnt
    return send_indices, recv_counts, m


def panel_chunk_lists(send_indices, recv_counts, n_chunks: int):
    """Cut an exchange into ``n_chunks`` chunk-sets for the panel-ordered SpMM (dense.SpmmPanelPlan): every link's list is
    cut at ``floor(c * len / n_chunks)`` on BOTH ends (sender: its index list; receiver: the count it expects), so the
    two ends agree without exchanging anything.  Returns

    * ``send_chunks[c][j]``  -- the part of ``send_indices[j]`` that travels in chunk-set c (order kept),
    * ``recv_chunk_counts[c][i]`` -- entries neighbour i delivers in chunk-set c,
    * ``chunk_of[p]``, ``newpos[p]`` -- for ghost position p of the UNCUT exchange (segments back to back in neighbour
      order): the chunk-set that carries it and its position in that chunk-set's ghost buffer (again the neighbours'
      parts back to back in neighbour order)."""
    cut = lambda n, c: (int(n) * c) // n_chunks
    send_chunks = [[np.asarray(i[cut(len(i), c):cut(len(i), c + 1)], dtype=np.int64) for i in send_indices]
                   for c in range(n_chunks)]
    recv_chunk_counts = [[cut(cnt, c + 1) - cut(cnt, c) for cnt in recv_counts] for c in range(n_chunks)]
    n_ghost = int(sum(int(c) for c in recv_counts))
    chunk_of = np.empty(n_ghost, dtype=np.int64)
    newpos = np.empty(n_ghost, dtype=np.int64)
    fill = [0] * n_chunks
    off = 0
    for cnt in recv_counts:
        cnt = int(cnt)
        for c in range(n_chunks):
            lo, hi = cut(cnt, c), cut(cnt, c + 1)
            chunk_of[off + lo:off + hi] = c
            newpos[off + lo:off + hi] = fill[c] + np.arange(hi - lo, dtype=np.int64)
            fill[c] += hi - lo
        off += cnt
    return send_chunks, recv_chunk_counts, chunk_of, newpos


# =====================================================================================================
# device plan
# =====================================================================================================
INT32_MAX = int(np.iinfo(np.int32).max)


def narrowing_enabled() -> bool:
    """``HPCLA_NARROW_INDICES=0`` keeps Int64 structures on the Int64 kernels (A/B measurements, and the tests that
    must still reach those kernels through the host layer)."""
    return _os.environ.get("HPCLA_NARROW_INDICES", "1").strip().lower() not in ("0", "off", "false", "no")


def can_narrow_indices(nnz: int, nrows_local: int, n_own: int, n_ghost: int, index_base: int = 0) -> bool:
    """May the kernels of a plan over an Int64 matrix stream Int32 indices?  Everything an index array of the plan
    can hold must fit: row pointers reach ``nnz + index_base``, split columns reach ``n_own + n_ghost - 1 +
    index_base`` (own offsets first, then positions in the ghost segment), send indices stay below ``n_own``.
    Indices are never results, so narrowing cannot change a bit of y (reference default ``Ti = Int``,
    src/backends.jl:348,369)."""
    lim = INT32_MAX - int(index_base)
    return (int(nnz) <= lim and int(nrows_local) <= lim and int(n_own) + int(n_ghost) <= lim and int(n_own) <= lim)


def split_colval(A: "HPCSparseMatrix", cmap: np.ndarray, to_i32: bool):
    """Device split-column ``colval`` of A through the compressed-column map ``cmap`` (one remap kernel over the stored
    entries): in A's index type, or -- ``to_i32`` with an Int64 matrix -- narrowed on the fly
    (``hpcla_remap_i64_to_i32``).  Returns (colval_split, cmap_dev)."""
    torch = _torch()
    dev = A.backend.torch_device
    a64 = A.Ti == np.dtype(np.int64)
    out64 = a64 and not to_i32
    cmap_dev = torch.from_numpy(cmap.astype(np.int64 if out64 else np.int32)).to(dev)
    out = torch.empty(A.nnz, dtype=torch.int64 if out64 else torch.int32, device=dev)
    fn = "hpcla_remap_i64" if out64 else ("hpcla_remap_i64_to_i32" if a64 else "hpcla_remap_i32")
    _capi.call(fn, dptr(A.colval_target()), dptr(cmap_dev), dptr(out), A.nnz, 0, current_stream_ptr())
    return out, cmap_dev


def narrowed_rowptr(A: "HPCSparseMatrix"):
    """Int32 copy of an Int64 ``rowptr_target`` (``hpcla_narrow_i64_to_i32``; the overflow word is checked: a caller
    may not reach this with nnz >= 2^31)."""
    torch = _torch()
    dev = A.backend.torch_device
    n = int(A.rowptr_target.numel())
    out = torch.empty(n, dtype=torch.int32, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    _capi.call("hpcla_narrow_i64_to_i32", dptr(A.rowptr_target), dptr(out), n, dptr(ovf), current_stream_ptr())
    if int(ovf.item()):
        raise OverflowError("rowptr does not fit Int32")
    return out


class VectorPlan:
    """Memoized communication + launch plan for ``A*x`` (the cache entry of _vector_plan_cache,
    src/HPCLinearAlgebra.jl:133).  Holds the reference plan's lists (``host``) and their device
    form: the RCCL halo plan handle, the split-column ``colval`` copy, and the interior/boundary
    row-block lists."""

    def __init__(self, A: "HPCSparseMatrix", x: HPCVector):
        assert_backends_compatible(A.backend, x.backend)
        require_device(A.backend, "VectorPlan(A, x)")
        torch = _torch()
        backend = A.backend
        dev = backend.torch_device
        self.backend = backend
        self.host = build_host_vector_plan(A.col_indices, x.partition, backend.comm)
        h = self.host
        self.n_own = h.n_own
        self.result_partition = A.row_partition.copy()                       # src/sparse.jl:2103-2106
        self.result_partition_hash = compute_partition_hash(self.result_partition)
        s = current_stream_ptr()

        # split-column colval (device): one remap kernel over the nonzeros
        cmap = split_column_map(h)
        n_ghost = h.n_gathered - len(h.local_dst_indices)
        if self.n_own + n_ghost > np.iinfo(A.Ti).max:
            raise OverflowError("split column space does not fit the index type")
        # Index type of the KERNEL arrays of this plan (colval_split, rowptr, send lists).  An Int64 matrix -- the
        # reference's default Ti = Int, src/backends.jl:348,369 -- whose nonzero count and split column space fit Int32
        # is NARROWED here, once per structure: every launch over the plan then takes the _i32 kernels (12 instead of
        # 16 bytes per stored entry; indices are not results, the bits of y cannot change).  The matrix keeps its type.
        a64 = A.Ti == np.dtype(np.int64)
        self.narrowed = bool(a64 and narrowing_enabled() and can_narrow_indices(A.nnz, A.nrows_local, self.n_own, n_ghost))
        self.is_i64 = a64 and not self.narrowed
        sfx = "i64" if self.is_i64 else "i32"
        Tk = np.int64 if self.is_i64 else np.int32
        self.colval_split, cmap_dev = split_colval(A, cmap, to_i32=self.narrowed)
        # rowptr of the kernels: the matrix's own array, or the plan's Int32 copy (matrices that share this plan share
        # the structural hash, hence the contents of rowptr)
        self._rowptr32 = narrowed_rowptr(A) if self.narrowed else None

        # halo plan (RCCL)
        self.halo = ctypes.c_void_p()
        self.n_ghost = n_ghost
        n_send, n_recv = len(h.send_rank_ids), len(h.recv_rank_ids)
        send_ranks = (ctypes.c_int32 * max(n_send, 1))(*h.send_rank_ids)
        send_counts = (ctypes.c_int64 * max(n_send, 1))(*[len(i) for i in h.send_indices])
        recv_ranks = (ctypes.c_int32 * max(n_recv, 1))(*h.recv_rank_ids)
        recv_counts = (ctypes.c_int64 * max(n_recv, 1))(*[len(p) for p in h.recv_perm])
        if n_send:
            send_idx = torch.from_numpy(np.concatenate(h.send_indices).astype(Tk)).to(dev)
        else:
            send_idx = None
        self.has_halo = (n_send + n_recv) > 0
        # Float32 plans (csrc/f32.hip) drive the exchange through begin / end with the ghost pointer taken from the host in
        # between: one ghost buffer, so that the pointer is a constant of the plan
        self.is_f32 = A.T == np.dtype(np.float32)
        self._stage32 = None
        if self.has_halo:
            torch.cuda.current_stream().synchronize()
            _capi.check("hpcla_halo_plan_create_ex", _capi.load().hpcla_halo_plan_create_ex(
                ctypes.byref(self.halo), backend.rccl, n_send, send_ranks, send_counts,
                dptr(send_idx), 1 if self.is_i64 else 0, n_recv, recv_ranks, recv_counts, 1,
                _capi.HALO_SINGLE_BUFFER if self.is_f32 else 0))
        # collective (ranks without neighbours take part with an empty descriptor): map the neighbours'
        # ghost windows -> push transport for this plan (csrc/window.hip)
        xp = np.asarray(x.partition, dtype=np.int64)
        probe = (self.n_own, 1, [(r, A.col_indices[perm] - xp[r]) for r, perm in zip(h.recv_rank_ids, h.recv_perm)])
        self.push = attach_halo_windows(backend, self.halo if self.has_halo else None, probe)

        # interior / boundary row blocks
        self.interior = self.boundary = None
        self.n_interior = self.n_boundary = 0
        if self.has_halo and A.nrows_local > 0:
            rpb = _capi.load().hpcla_spmv_rows_per_block()
            nblk = (A.nrows_local + rpb - 1) // rpb
            flags = torch.empty(nblk, dtype=torch.int32, device=dev)
            _capi.call(f"hpcla_classify_blocks_{sfx}", dptr(self.rowptr_of(A)), dptr(self.colval_split),
                       A.nrows_local, 0, self.n_own, rpb, dptr(flags), s)
            self.interior = torch.nonzero(flags == 0).flatten().to(torch.int32).contiguous()
            self.boundary = torch.nonzero(flags != 0).flatten().to(torch.int32).contiguous()
            self.n_interior = int(self.interior.numel())
            self.n_boundary = int(self.boundary.numel())
        self._keep = (cmap_dev, send_idx)
        # block order of the SpMV launches over this structure: measured once, here (hpcla_spmv_tune_block_order_*, a few
        # dozen launches into a scratch vector); HPCLA_BLOCK_ORDER=natural skips it, =<G> forces groups of G row blocks
        self.block_group = 1
        self.block_group_measured = False
        want = _os.environ.get("HPCLA_BLOCK_ORDER", "auto").strip().lower()
        if want.isdigit():
            g = int(want)
            if g > 1024 or (g & (g - 1)) != 0:           # the hint takes 0 / 1 or a power of two <= 1024
                import sys
                sys.stderr.write(f"hpcla: HPCLA_BLOCK_ORDER={want} is not a power of two <= 1024; natural order\n")
                g = 1
            self.block_group = max(1, g)
        elif (want != "natural" and not self.is_f32 and A.nrows_local > 0 and A.nnz > 0
              and int(x.v.numel()) >= max(self.n_own, 1)):
            # (x only has to be readable at the plan's own columns: the products are discarded; a width-0 SpMM probe is not)
            scratch = torch.empty(A.nrows_local, dtype=torch.float64, device=dev)
            ghost, _ng = self.ghost_tensor_ptr()
            chosen = ctypes.c_int(1)
            try:
                _capi.call(f"hpcla_spmv_tune_block_order_f64_{sfx}", dptr(self.rowptr_of(A)), dptr(self.colval_split),
                           dptr(A.nzval), dptr(x.v), ghost, self.n_own, dptr(scratch), A.nrows_local, A.nnz, 0, s,
                           ctypes.byref(chosen))
            except _capi.HPCLAError as exc:        # an optional performance step must not take A*x down with it
                import sys
                sys.stderr.write(f"hpcla: block-order measurement failed ({exc}); natural order\n")
                chosen = ctypes.c_int(1)
            self.block_group = int(chosen.value)
            self.block_group_measured = (A.nrows_local + 255) // 256 >= 4096      # the tuner's own threshold (64 launches)
            # the tuner left it registered for the rowptr array it was given: the matrix's, or this plan's Int32 copy
            owner = self if self.narrowed else A
            owner._block_order_hint = A._block_order_hint = self.block_group
            if self.block_group > 1:
                import weakref
                owner._block_order_finalizer = weakref.finalize(owner, _unhint_block_order, self.rowptr_of(A).data_ptr())

    def rowptr_of(self, A: "HPCSparseMatrix"):
        """The ``rowptr`` the kernels of this plan read for matrix A: A's own device array, or -- narrowed plan -- the
        plan's Int32 copy (equal contents for every matrix that shares the plan: the structural hash covers rowptr)."""
        return self._rowptr32 if self.narrowed else A.rowptr_target

    def stage_f32(self, n: int):
        """Staging vector of a Float32 exchange (hpcla_halo_begin_f32): n doubles, written at the send positions only."""
        if self._stage32 is None or int(self._stage32.numel()) < n:
            self._stage32 = _torch().empty(max(n, 1), dtype=_torch().float64, device=self.backend.torch_device)
        return self._stage32

    def ghost_tensor_ptr(self) -> Tuple[ctypes.c_void_p, int]:
        g = ctypes.c_void_p()
        n = ctypes.c_int64()
        if self.has_halo:
            _capi.call("hpcla_halo_ghost_ptr", self.halo, ctypes.byref(g), ctypes.byref(n))
        return g, n.value

    def timed_out(self) -> bool:
        """True if a push-mode exchange of this plan gave up waiting for a neighbour (results invalid)."""
        if not self.halo:
            return False
        flag = ctypes.c_int(0)
        _capi.call("hpcla_halo_status", self.halo, ctypes.byref(flag))
        return bool(flag.value)

    def destroy(self) -> None:
        if self.halo:
            _capi.call("hpcla_halo_plan_destroy", self.halo)
            self.halo = ctypes.c_void_p()


# memoization (src/HPCLinearAlgebra.jl:126-164; clear_plan_cache! :181-201)
_vector_plan_cache: Dict[tuple, VectorPlan] = {}


def get_vector_plan(A: "HPCSparseMatrix", x: HPCVector) -> VectorPlan:
    """src/sparse.jl:1992-2001."""
    key = (A._ensure_hash(), x.structural_hash, str(A.T), str(A.Ti), "ROCArray")
    if A.Ti == np.dtype(np.int64) and not narrowing_enabled():
        key += ("wide",)                       # HPCLA_NARROW_INDICES=0: a plan of its own, on the Int64 kernels
    plan = _vector_plan_cache.get(key)
    if plan is None:
        plan = VectorPlan(A, x)
        plan.key = key
        _vector_plan_cache[key] = plan
    if getattr(plan if plan.narrowed else A, "_block_order_hint", 1) != plan.block_group:
        _hint_block_order(A, plan)
    A._block_order_hint = plan.block_group
    return plan


def _unhint_block_order(ptr: int) -> None:
    try:
        _capi.load().hpcla_spmv_block_order_hint(ctypes.c_void_p(ptr), 0)
    except Exception:                          # interpreter shutdown
        pass


def _unhint_spmm_block_order(ptr: int) -> None:
    try:
        _capi.load().hpcla_spmm_block_order_hint(ctypes.c_void_p(ptr), 0)
    except Exception:                          # interpreter shutdown
        pass


def _hint_block_order(A, plan: VectorPlan) -> None:
    """Tell the library the block order of SpMV launches over THIS matrix under this plan (keyed by the rowptr device
    pointer the kernels read: the matrix's own array, which several matrices of one structure do not share, or a
    narrowed plan's Int32 copy, which they do); the hint goes when the owner of that array does (a host-side table
    entry: no device work in the finaliser)."""
    import weakref
    group = int(plan.block_group)
    owner = plan if plan.narrowed else A
    ptr = plan.rowptr_of(A).data_ptr()
    _capi.call("hpcla_spmv_block_order_hint", ptr, group)
    owner._block_order_hint = group
    if group > 1 and not getattr(owner, "_block_order_finalizer", None):
        owner._block_order_finalizer = weakref.finalize(owner, _unhint_block_order, ptr)


def clear_plan_cache() -> None:
    """``clear_plan_cache!`` (src/HPCLinearAlgebra.jl:181-201): plans are freed only here, never
    by a finaliser (a finaliser must not issue device/collective work)."""
    _quiesce([p.backend for p in _vector_plan_cache.values()])
    dead = [p for p in _vector_plan_cache.values() if p.timed_out()]
    for p in _vector_plan_cache.values():
        p.destroy()
    _vector_plan_cache.clear()
    if dead:
        import warnings
        warnings.warn(f"clear_plan_cache: {len(dead)} plan(s) had a timed-out exchange (their results were NaN)")


class ExchangeTimeout(RuntimeError):
    """A peer-window exchange or all-reduce gave up waiting for a neighbour (HPCLA_PUSH_TIMEOUT_S): every result
    that depended on it has been poisoned with NaN by the kernels; the plans involved are dead."""


def check_exchange_health(backend=None, always: bool = False) -> None:
    """Raise ExchangeTimeout if the backend's communicator or any cached plan reports an expired spin.
    Called wherever a NaN scalar reaches the host (dot / norm / CG history read-backs): the device poisons the
    result of an expired wait, this turns the poison into an error that says what happened -- the reference's
    MPI exchange would have blocked instead (src/vectors.jl:446).  Each status is a synchronising 4-byte read,
    so the good path never pays for it."""
    bad = []
    if backend is not None and getattr(backend, "rccl", None) and comm_size(backend.comm) > 1:
        flag = ctypes.c_int(0)
        _capi.call("hpcla_comm_status", backend.rccl, ctypes.byref(flag))
        if flag.value:
            bad.append("window all-reduce")
    for plan in _vector_plan_cache.values():
        if plan.timed_out():
            bad.append("halo plan of a sparse matrix")
    from . import dense, matmat

    def _status(handle) -> bool:
        if not handle:
            return False
        flag = ctypes.c_int(0)
        _capi.call("hpcla_halo_status", handle, ctypes.byref(flag))
        return bool(flag.value)

    for ent in dense._spmm_halo_cache.values():
        if _status(ent[0]):
            bad.append("SpMM ghost-row plan")
    for pp in dense._spmm_panel_cache.values():           # the chained chunk-set plans of the panel-ordered SpMM
        if any(_status(h) for h in pp.halos):
            bad.append("SpMM chunk-set plan (panel order)")
    for mp in matmat._plan_cache.values():                # sparse x sparse: the value exchange of the gathered rows
        if _status(getattr(mp, "halo", None)):
            bad.append("MatrixPlan value exchange")
    if bad:
        raise ExchangeTimeout("exchange timed out (" + ", ".join(sorted(set(bad))) + "): a neighbour did not publish its "
                              "values within HPCLA_PUSH_TIMEOUT_S; the affected results are NaN")


def _quiesce(backends) -> None:
    """Before plans are freed: drain this device and meet the other ranks, so that no peer is still
    storing into (or polling) a ghost window that is about to be unmapped.  Collective, like
    clear_plan_cache! itself."""
    seen = set()
    for b in backends:
        if id(b.comm) in seen:
            continue
        seen.add(id(b.comm))
        _torch().cuda.synchronize()
        if b.peer_windows and comm_size(b.comm) > 1:
            comm_barrier(b.comm)


def cache_sizes() -> Dict[str, int]:
    return {"vector_plan_cache": len(_vector_plan_cache)}


def execute_plan(plan: VectorPlan, x: HPCVector):
    """``execute_plan!(plan, x)`` (src/vectors.jl:394-463): returns ``gathered = x[col_indices]``
    as a device tensor.  API-parity form only -- ``A*x`` never materialises ``gathered``: it reads
    x.v and the ghost segment in place."""
    torch = _torch()
    h = plan.host
    dev = x.v.device
    s = current_stream_ptr()
    from .vectors import f64_only
    f64_only(x.backend, "execute_plan (the materialised `gathered`; A*x itself reads x.v and the ghosts in place)")
    gathered = torch.empty(h.n_gathered, dtype=torch.float64, device=dev)
    if plan.has_halo:
        _capi.call("hpcla_halo_begin", plan.halo, dptr(x.v), s)
        _capi.call("hpcla_halo_end", plan.halo, s)
    if len(h.local_src_indices):
        src = torch.from_numpy(h.local_src_indices).to(dev)
        dst = torch.from_numpy(h.local_dst_indices).to(dev)
        _capi.call("hpcla_gather_f64_i64", dptr(x.v), dptr(src), dptr(dst), dptr(gathered),
                   len(h.local_src_indices), 0, s)
    if plan.has_halo:
        gptr, n = plan.ghost_tensor_ptr()
        ghost_pos = torch.from_numpy(np.concatenate(h.recv_perm)).to(dev)
        ident = torch.arange(n, dtype=torch.int64, device=dev)
        _capi.call("hpcla_gather_f64_i64", gptr, dptr(ident), dptr(ghost_pos), dptr(gathered), n, 0, s)
    return gathered


# =====================================================================================================
# HPCSparseMatrix
# =====================================================================================================
class HPCSparseMatrix:
    """``HPCSparseMatrix{T,Ti,B}`` (src/sparse.jl:319-337).  Host: partitions, ``col_indices``,
    ``rowptr``/``colval`` (compressed local columns).  Device: ``nzval``, ``rowptr_target`` and --
    uploaded lazily, the hot path uses the plan's split copy -- ``colval_target``."""

    def __init__(self, row_partition, col_partition, col_indices, rowptr, colval, nzval_dev,
                 rowptr_dev, backend: HPCBackend):
        self.structural_hash: Optional[bytes] = None
        self.row_partition = np.asarray(row_partition, dtype=np.int64)
        self.col_partition = np.asarray(col_partition, dtype=np.int64)
        self.col_indices = np.asarray(col_indices, dtype=np.int64)
        self._rowptr = rowptr                            # host copies ("always CPU" in the reference);
        self._colval = colval                            # None when built on the device -> lazy D2H
        self.nzval = nzval_dev
        self.rowptr_target = rowptr_dev
        self._colval_target = None
        self.cached_transpose = None                     # src/sparse.jl:331, filled by transpose()
        self._packed = {}                                # plan cache key -> packed handle (opt-in)
        self.packed_reason = ""
        self._long_rows = None                           # (device row list, min length, work buffer): enable_long_rows()
        self.nrows_local = int(rowptr_dev.numel()) - 1
        self.ncols_compressed = len(self.col_indices)
        self.backend = backend
        self.T = backend.T
        self.Ti = backend.Ti

    @property
    def nnz(self) -> int:
        return int(self.nzval.numel())

    @property
    def rowptr(self) -> np.ndarray:
        if self._rowptr is None:
            self._rowptr = self.rowptr_target.cpu().numpy()
        return self._rowptr

    @property
    def colval(self) -> np.ndarray:
        if self._colval is None:
            self._colval = self._colval_target.cpu().numpy()
        return self._colval

    @property
    def shape(self) -> Tuple[int, int]:                  # src/sparse.jl:2151-2155
        return int(self.row_partition[-1]), int(self.col_partition[-1])

    # -- OPT-IN packed copy (csrc/packed.hip).  It snapshots the VALUES, so it lives on the matrix
    #    object, never on the structure-keyed VectorPlan (two matrices with the same structure share
    #    a plan).  If nzval is modified in place afterwards, call disable_packed() first.
    def enable_packed(self, x: HPCVector) -> bool:
        """Build the packed copy (16-bit block-relative columns + 8-bit value codes: 3 B per stored
        entry instead of 12) of the interior row blocks of the plan for ``(A, x.partition)``.
        Returns False, leaving the CSR path in place, when the matrix is not packable (> 256
        distinct values, column window > 16 bit, Int64 indices); the reason is in
        ``self.packed_reason``.  Results are bit-identical either way."""
        return self._packed_create(get_vector_plan(self, x))

    def _packed_create(self, plan) -> bool:
        if self._packed.get(_plan_key(plan)) is not None:
            return True
        self._packed[_plan_key(plan)] = None
        if plan.is_i64:
            self.packed_reason = "Int64 indices"
            return False
        if plan.is_f32:
            self.packed_reason = "Float32 values (the packed copy is a Float64 kernel)"
            return False
        h = ctypes.c_void_p()
        rc = _capi.load().hpcla_packed_create_i32(
            ctypes.byref(h), dptr(plan.rowptr_of(self)), dptr(plan.colval_split), dptr(self.nzval),
            self.nrows_local, self.nnz, plan.n_own, 0, dptr(plan.interior) if plan.has_halo else None,
            plan.n_interior if plan.has_halo else 0, current_stream_ptr())
        if rc == -5:        # HPCLA_ERR_UNSUPPORTED: not packable, keep CSR
            self.packed_reason = _capi.last_error()
            return False
        _capi.check("hpcla_packed_create_i32", rc)
        self._packed[_plan_key(plan)] = h
        return True

    def disable_packed(self) -> None:
        for h in self._packed.values():
            if h is not None:
                _capi.call("hpcla_packed_destroy", h)
        self._packed.clear()

    def _packed_for(self, plan):
        """Packed handle for this plan or None; HPCLA_SPMV_PACKED=1 opts every matrix in."""
        key = _plan_key(plan)
        if key not in self._packed and _PACKED_BY_ENV:
            self._packed_create(plan)
        return self._packed.get(key)

    # -- OPT-IN long rows (csrc/spmv.hip LONGR).  NOT the reference's bits: the listed rows are summed in tree order.
    def enable_long_rows(self, min_len: int = 4096) -> int:
        """Sum rows of at least ``min_len`` stored entries in TREE order (1024 pieces per row, wave shuffles) instead of on one
        lane: an arrow matrix's dense row is a sequential pass over n entries in the default kernel -- the reference's order
        (src/sparse.jl:2059-2064) and its cliff.  Results of those rows then agree with the sequential sum to
        1e-12 * (|A||x|)_r, every other row keeps its bits.  Single-rank plans (no halo), Float64.  Returns the number of
        long rows (0: nothing changes).  The list is derived HERE from the device rowptr (one pass), so it is exactly the set
        the kernels leave out; a product over a plan with neighbours, the packed copy and Float32 do not honour the switch
        (they keep the default order) -- said once, on stderr, when the matrix's backend has more than one rank."""
        torch = _torch()
        if comm_size(self.backend.comm) > 1 or self.T != np.dtype(np.float64):
            import sys
            sys.stderr.write("hpcla: enable_long_rows applies to single-rank Float64 products only; this matrix's products over "
                             "plans with neighbours (and Float32 / packed products) keep the default row order\n")
        if min_len < 928:
            raise ValueError("enable_long_rows: min_len must be at least 928 (two passes of a wave)")
        rp = self.rowptr_target
        rows = torch.nonzero((rp[1:] - rp[:-1]) >= min_len).flatten().to(torch.int64).contiguous()
        n_long = int(rows.numel())
        if n_long > 65535:
            raise ValueError("enable_long_rows: more than 65535 rows qualify; raise min_len")
        if n_long == 0:
            self._long_rows = None
            return 0
        work = torch.empty(_capi.load().hpcla_spmv_longrows_work_bytes(n_long) // 8, dtype=torch.float64, device=rp.device)
        self._long_rows = (rows, int(min_len), work)
        return n_long

    def disable_long_rows(self) -> None:
        self._long_rows = None

    def transpose(self):
        """Lazy ``transpose(A)`` (src/sparse.jl:2254-2258); ``transpose(A) @ x`` materialises and
        caches A^T (linearalgebrampi.jl_amd/transpose.py)."""
        from .transpose import TransposedHPCSparseMatrix
        return TransposedHPCSparseMatrix(self)

    @property
    def T_(self):
        return self.transpose()

    def colval_target(self):
        if self._colval_target is None:
            torch = _torch()
            self._colval_target = torch.from_numpy(self.colval).to(self.backend.torch_device)
        return self._colval_target

    def _ensure_hash(self) -> bytes:                     # src/HPCLinearAlgebra.jl:759-764
        """Structural hash (memoization key, src/sparse.jl:97-121).  The array passes run where the arrays
        live: rowptr / colval are digested ON THE DEVICE (``hpcla_digest_*``) when their device copies exist
        -- a device-built matrix never copies colval to the host for this -- else by the numpy twin; both give
        the same words."""
        if self.structural_hash is None:
            from .partition import array_digest, structural_hash_from_digests

            def dig(dev_t, host_a):
                if dev_t is not None and dev_t.is_cuda:
                    out = (ctypes.c_uint64 * 4)()
                    sfx = "i64" if dev_t.dtype == _torch().int64 else "i32"
                    _capi.call(f"hpcla_digest_{sfx}", dptr(dev_t), int(dev_t.numel()), out, current_stream_ptr())
                    return int(dev_t.numel()), bytes(out)
                return int(len(host_a)), array_digest(host_a)

            ci = getattr(self, "_col_indices_dev", None)
            cv = self._colval_target
            if self.backend.on_device:               # host-built matrix: the plan needs colval on the device anyway
                cv = self.colval_target()
                if ci is None and len(self.col_indices) > (1 << 16):
                    ci = _torch().from_numpy(self.col_indices).to(self.backend.torch_device)
            self.structural_hash = structural_hash_from_digests(
                self.row_partition,
                [dig(ci, self.col_indices),
                 dig(self.rowptr_target, None if self.rowptr_target.is_cuda else self.rowptr),
                 dig(cv, None if cv is not None and cv.is_cuda else self.colval)],
                self.backend.comm)
        return self.structural_hash

    # -- A * x (src/sparse.jl:2096-2128) and A * B (src/sparse.jl:2391-2413) ---------------------------
    def __matmul__(self, other):
        from .dense import HPCMatrix, spmm
        if isinstance(other, HPCVector):
            plan = get_vector_plan(self, other)
            y = HPCVector(plan.result_partition_hash, plan.result_partition,
                          _torch().empty(self.nrows_local, dtype=other.v.dtype,
                                         device=self.backend.torch_device), self.backend)
            _spmv_into(y, self, other, plan)
            return y
        if isinstance(other, HPCMatrix):
            return spmm(self, other)
        if isinstance(other, HPCSparseMatrix):
            from .matmat import spgemm
            return spgemm(self, other)
        from .transpose import TransposedHPCSparseMatrix
        if isinstance(other, TransposedHPCSparseMatrix):            # A * transpose(B), src/sparse.jl:2364-2368
            return self.__matmul__(other.materialize())
        return NotImplemented

    def __mul__(self, other):
        if isinstance(other, (int, float, np.floating, np.integer)):
            return self._scaled(float(other))
        return self.__matmul__(other)

    def __rmul__(self, a):
        """``a * A`` (src/sparse.jl:2289-2308): new values, the structure arrays (and hash) are shared."""
        if isinstance(a, (int, float, np.floating, np.integer)):
            return self._scaled(float(a))
        return NotImplemented

    def __neg__(self):                                               # src/sparse.jl:2315
        return self._scaled(-1.0)

    def _with_values(self, nzval) -> "HPCSparseMatrix":
        out = HPCSparseMatrix(self.row_partition, self.col_partition, self.col_indices, self._rowptr,
                              self._colval, nzval, self.rowptr_target, self.backend)
        out._colval_target = self._colval_target
        out.structural_hash = self.structural_hash
        return out

    def _scaled(self, a: float) -> "HPCSparseMatrix":
        out = _torch().empty_like(self.nzval)
        from .vectors import sfx_of
        _capi.call(f"hpcla_scale_{sfx_of(self.backend)}", a, dptr(self.nzval), dptr(out), self.nnz, current_stream_ptr())
        return self._with_values(out)

    def copy(self) -> "HPCSparseMatrix":                             # src/sparse.jl:2458
        return self._with_values(self.nzval.clone())

    def norm(self, p: float = 2) -> float:
        """``norm(A, p)`` (src/sparse.jl:2172-2195): over the stored entries (p = 2: Frobenius)."""
        from .vectors import norm as vnorm
        sizes = comm_allgather(self.backend.comm, np.array([self.nnz], dtype=np.int64))
        part = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        return vnorm(HPCVector(compute_partition_hash(part), part, self.nzval, self.backend), p)

    def __add__(self, other):
        if isinstance(other, HPCSparseMatrix):
            from .addition import sparse_add
            return sparse_add(self, other, subtract=False)
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, HPCSparseMatrix):
            from .addition import sparse_add
            return sparse_add(self, other, subtract=True)
        return NotImplemented


def _spmv_into(y: HPCVector, A: HPCSparseMatrix, x: HPCVector, plan: VectorPlan) -> None:
    if y.local_length != A.nrows_local:
        raise ValueError("mul!: y has the wrong local length")
    if x.local_length != plan.n_own:
        raise ValueError("A*x: x does not match the plan's partition")
    if plan.is_f32:
        _spmv_into_f32(y, A, x, plan)
        return
    pk = A._packed_for(plan)
    if pk is not None:
        _capi.call("hpcla_spmv_dist_packed_f64_i32", plan.halo if plan.has_halo else None, A.backend.rccl,
                   pk, dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval), dptr(x.v),
                   plan.n_own, dptr(y.v), A.nrows_local, A.nnz, 0, dptr(plan.interior), plan.n_interior,
                   dptr(plan.boundary), plan.n_boundary, None, None, current_stream_ptr())
        return
    sfx = "i64" if plan.is_i64 else "i32"
    if A._long_rows is not None and not plan.has_halo:
        rows, min_len, work = A._long_rows               # opt-in: the listed rows in tree order (enable_long_rows)
        _capi.call(f"hpcla_spmv_longrows_f64_{sfx}", dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval), dptr(x.v),
                   None, plan.n_own, dptr(y.v), A.nrows_local, A.nnz, 0, dptr(rows), int(rows.numel()), min_len, dptr(work),
                   current_stream_ptr())
        return
    _capi.call(f"hpcla_spmv_dist_f64_{sfx}", plan.halo if plan.has_halo else None,
               dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval), dptr(x.v), plan.n_own,
               dptr(y.v), A.nrows_local, A.nnz, 0, dptr(plan.interior), plan.n_interior,
               dptr(plan.boundary), plan.n_boundary, current_stream_ptr())


def _spmv_into_f32(y: HPCVector, A: HPCSparseMatrix, x: HPCVector, plan: VectorPlan) -> None:
    """The Float32 form of the distributed product (csrc/f32.hip, hpcla_spmv_dist_f32_*): the exchange widens what it sends
    into the plan's staging vector and runs on the plan's side stream / in the peers' windows while the interior row blocks
    multiply; the boundary blocks follow the exchange and narrow the ghost values they gather.  Same transports, same block
    lists as the Float64 product; the plan is single-buffered (a constant ghost pointer)."""
    sfx = "i64" if plan.is_i64 else "i32"
    _capi.call(f"hpcla_spmv_dist_f32_{sfx}", plan.halo if plan.has_halo else None, dptr(plan.rowptr_of(A)),
               dptr(plan.colval_split), dptr(A.nzval), dptr(x.v), plan.n_own, dptr(y.v), A.nrows_local, A.nnz, 0,
               dptr(plan.interior), plan.n_interior, dptr(plan.boundary), plan.n_boundary,
               dptr(plan.stage_f32(plan.n_own)) if plan.has_halo else None, current_stream_ptr())


def mul_(y: HPCVector, A: HPCSparseMatrix, x: HPCVector) -> HPCVector:
    """``LinearAlgebra.mul!(y, A, x)`` (src/sparse.jl:2019-2037), run on the device into ``y.v`` --
    the reference's version multiplies on the CPU and reads a stale buffer on 1-rank GPU backends
    (SURVEY.md section 3.2); neither is reproduced."""
    assert_backends_compatible(A.backend, x.backend)
    assert_backends_compatible(A.backend, y.backend)
    plan = get_vector_plan(A, x)
    if y.structural_hash != plan.result_partition_hash:      # hash cached on the plan (:2103-2106)
        raise ValueError("mul!: y must have A's row partition")
    _spmv_into(y, A, x, plan)
    return y


def mul_dot_(y: HPCVector, A: HPCSparseMatrix, x: HPCVector, out) -> HPCVector:
    """Fused ``mul!(y, A, x)`` and ``out = dot(x, y)`` (CG's p.Ap) in one pass over A: the SpMV
    epilogue leaves per-row-block partial sums that are reduced deterministically and all-reduced;
    the scalar stays in the 1-element device tensor ``out``.  Requires x partitioned like A's rows."""
    assert_backends_compatible(A.backend, x.backend)
    assert_backends_compatible(A.backend, y.backend)
    from .vectors import f64_only
    f64_only(A.backend, "mul_dot_ (the fused p.Ap epilogue)")
    plan = get_vector_plan(A, x)
    if y.structural_hash != plan.result_partition_hash or x.structural_hash != plan.result_partition_hash:
        raise ValueError("mul_dot_: x and y must have A's row partition")
    torch = _torch()
    work = getattr(plan, "_dot_work", None)
    if work is None:
        nbytes = _capi.load().hpcla_spmv_dot_work_bytes(A.nrows_local)
        work = plan._dot_work = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=x.v.device)
    pk = A._packed_for(plan)
    if pk is not None:
        _capi.call("hpcla_spmv_dist_packed_f64_i32", plan.halo if plan.has_halo else None, A.backend.rccl,
                   pk, dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval), dptr(x.v),
                   plan.n_own, dptr(y.v), A.nrows_local, A.nnz, 0, dptr(plan.interior), plan.n_interior,
                   dptr(plan.boundary), plan.n_boundary, dptr(out), dptr(work), current_stream_ptr())
        return y
    sfx = "i64" if plan.is_i64 else "i32"
    _capi.call(f"hpcla_spmv_dist_dot_f64_{sfx}", plan.halo if plan.has_halo else None, A.backend.rccl,
               dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval), dptr(x.v), plan.n_own,
               dptr(y.v), A.nrows_local, A.nnz, 0, dptr(plan.interior), plan.n_interior,
               dptr(plan.boundary), plan.n_boundary, dptr(out), dptr(work), current_stream_ptr())
    return y


def _compress_columns(colidx_global: np.ndarray, ncols_global: int, Ti) -> Tuple[np.ndarray, np.ndarray]:
    """``col_indices = unique!(sort(copy(rowval)))`` + ``compress_AT`` (src/sparse.jl:501-509,
    137-144).  Same result via a presence bitmap (O(nnz + ncols) instead of a sort and a binary
    search per nonzero)."""
    if len(colidx_global) == 0:
        return np.empty(0, dtype=np.int64), np.empty(0, dtype=Ti)
    lo, hi = int(colidx_global.min()), int(colidx_global.max())
    if lo < 0 or hi >= ncols_global:
        raise ValueError("column index out of range")
    present = np.zeros(hi - lo + 1, dtype=bool)
    present[colidx_global - lo] = True
    col_indices = np.flatnonzero(present).astype(np.int64) + lo
    lut = np.cumsum(present, dtype=np.int64) - 1
    colval = lut[colidx_global - lo].astype(Ti)
    return col_indices, colval


def HPCSparseMatrix_local(rowptr, colidx_global, vals, ncols_global: int, backend: HPCBackend,
                          col_partition: Optional[np.ndarray] = None) -> HPCSparseMatrix:
    """``HPCSparseMatrix_local(A_local, backend; col_partition)`` (src/sparse.jl:454-525): this rank's
    rows as CSR with GLOBAL column ids (the reference's ``A_local.parent.rowval``).  Row partition is
    inferred by an Allgather of ``[nrows, ncols]``; all ranks must agree on the column count."""
    torch = _torch()
    comm = backend.comm
    nranks = comm_size(comm)
    Ti = backend.Ti.type
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx_global = np.asarray(colidx_global, dtype=np.int64)
    vals = np.ascontiguousarray(vals, dtype=backend.T)
    nrows = len(rowptr) - 1
    if rowptr[0] != 0 or rowptr[-1] != len(colidx_global) or len(vals) != len(colidx_global):
        raise ValueError("HPCSparseMatrix_local: inconsistent CSR arrays")
    info = comm_allgather(comm, np.array([nrows, ncols_global], dtype=np.int64)).reshape(nranks, 2)
    if not np.all(info[:, 1] == info[0, 1]):                       # src/sparse.jl:487-490
        raise ValueError("HPCSparseMatrix_local: All ranks must have the same number of columns. "
                         f"Got column counts: {info[:, 1].tolist()}")
    row_partition = np.concatenate([[0], np.cumsum(info[:, 0])]).astype(np.int64)
    if col_partition is None:
        col_partition = uniform_partition(ncols_global, nranks)
    if len(colidx_global) > np.iinfo(Ti).max:
        raise OverflowError("nnz does not fit the backend index type")
    col_indices, colval = _compress_columns(colidx_global, ncols_global, Ti)
    rowptr_ti = rowptr.astype(Ti)
    dev = backend.torch_device
    nzval_dev = torch.from_numpy(vals).to(dev)                     # _convert_array hook (:518)
    rowptr_dev = torch.from_numpy(rowptr_ti).to(dev)               # _to_target_device hook (:519)
    return HPCSparseMatrix(row_partition, col_partition, col_indices, rowptr_ti, colval, nzval_dev,
                           rowptr_dev, backend)


def HPCSparseMatrix_local_device(rowptr_dev, colidx_global_dev, vals_dev, ncols_global: int,
                                 backend: HPCBackend, col_partition: Optional[np.ndarray] = None,
                                 col_window: Optional[Tuple[int, int]] = None) -> HPCSparseMatrix:
    """Device-side twin of :func:`HPCSparseMatrix_local`: the local rows arrive
    as DEVICE tensors (``rowptr`` int64 0-based, global column ids int64, values f64) and the
    compressed column space is built on the GPU (presence bitmap + scan, ``hpcla_compress_columns_*``)
    instead of the reference's host sort + binary search (src/sparse.jl:501-509, 137-144).
    ``col_window = (lo, hi)`` bounds the column ids (default: their min/max)."""
    torch = _torch()
    comm = backend.comm
    nranks = comm_size(comm)
    is64 = backend.Ti == np.dtype(np.int64)
    tdt = torch.int64 if is64 else torch.int32
    dev = backend.torch_device
    nrows, nnz = int(rowptr_dev.numel()) - 1, int(vals_dev.numel())
    if nnz > np.iinfo(backend.Ti.type).max:
        raise OverflowError("nnz does not fit the backend index type")
    info = comm_allgather(comm, np.array([nrows, ncols_global], dtype=np.int64)).reshape(nranks, 2)
    if not np.all(info[:, 1] == info[0, 1]):
        raise ValueError("HPCSparseMatrix_local: All ranks must have the same number of columns. "
                         f"Got column counts: {info[:, 1].tolist()}")
    row_partition = np.concatenate([[0], np.cumsum(info[:, 0])]).astype(np.int64)
    if col_partition is None:
        col_partition = uniform_partition(ncols_global, nranks)
    if col_window is None:
        col_window = (int(colidx_global_dev.min().item()), int(colidx_global_dev.max().item())) if nnz else (0, 0)
    lo, hi = int(col_window[0]), int(col_window[1])
    if lo < 0 or hi >= ncols_global:
        raise ValueError("column index out of range")
    window = hi - lo + 1
    work = torch.empty(_capi.load().hpcla_colspace_work_bytes(window), dtype=torch.uint8, device=dev)
    colval_dev = torch.empty(nnz, dtype=tdt, device=dev)
    col_indices_dev = torch.empty(window, dtype=torch.int64, device=dev)
    ncomp = ctypes.c_int64()
    _capi.call("hpcla_compress_columns_i64" if is64 else "hpcla_compress_columns_i32",
               dptr(colidx_global_dev), nnz, lo, window, dptr(colval_dev), 0, dptr(col_indices_dev),
               ctypes.byref(ncomp), dptr(work), current_stream_ptr())
    col_indices = col_indices_dev[:ncomp.value].cpu().numpy()
    from .vectors import torch_dtype_of
    A = HPCSparseMatrix(row_partition, col_partition, col_indices, None, None, vals_dev.to(torch_dtype_of(backend)),
                        rowptr_dev.to(tdt), backend)
    A._colval_target = colval_dev
    A._col_indices_dev = col_indices_dev[:ncomp.value]
    return A


def HPCSparseMatrix_from_global(A, backend: HPCBackend, row_partition=None, col_partition=None) -> HPCSparseMatrix:
    """``HPCSparseMatrix(A::SparseMatrixCSC, backend; row_partition, col_partition)``
    (src/sparse.jl:398-413): every rank passes the same global scipy.sparse matrix and keeps its
    row slice."""
    import scipy.sparse as sp
    comm = backend.comm
    nranks, rank = comm_size(comm), comm_rank(comm)
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    m, n = A.shape
    if row_partition is None:
        row_partition = uniform_partition(m, nranks)
    if col_partition is None:
        col_partition = uniform_partition(n, nranks)
    lo, hi = int(row_partition[rank]), int(row_partition[rank + 1])
    loc = A[lo:hi, :]
    M = HPCSparseMatrix_local(loc.indptr, loc.indices, loc.data, n, backend, col_partition=col_partition)
    if not np.array_equal(M.row_partition, np.asarray(row_partition, dtype=np.int64)):
        raise ValueError("row_partition inconsistent across ranks")
    return M
