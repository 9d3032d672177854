"""``HPCSparseMatrix * HPCSparseMatrix`` (SpGEMM) on DeviceROCm -- SURVEY.md section 8f "next" rank 3.

Reference (src/sparse.jl:991-1059): a memoized ``MatrixPlan`` (src/sparse.jl:554-978) gathers the rows
of B named by ``A.col_indices`` (structure exchanged once at plan time, values with tag-3 messages at
every product), then the local product is Julia's CPU SparseArrays multiply -- for GPU backends too
(``CT = plan.AT * A_csc``, :1011) -- and the result is compressed into a new ``HPCSparseMatrix`` with
``row_partition = A.row_partition`` and ``col_partition = B.col_partition``.

Here the plan (host, numpy + comm_*) keeps the reference's shape: who needs which rows, the
structure of the gathered matrix G, and the value-movement lists.  Per product: B's values travel
device-to-device (own rows: one gather kernel; remote rows: the same RCCL halo plan the SpMV uses,
applied to ``B.nzval`` with nonzero positions as indices) and the local product runs in
``csrc/spgemm.hip`` -- Gustavson per row with an LDS hash table, k ascending, so every C(i,j) is summed in
the reference's order (bit-identical).  The result's column space is compressed on the device
(``HPCSparseMatrix_local_device``).
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List

import numpy as np

from . import _capi
from .backends import (assert_backends_compatible, comm_alltoall_counts, comm_exchange_arrays, comm_rank,
                       comm_size)
from .partition import owner_of
from .vectors import current_stream_ptr, dptr


def _torch():
    import torch
    return torch


@dataclass
class HostRowGatherPlan:
    """Index half of ``MatrixPlan`` (src/sparse.jl:554-576), 0-based.  G row r = B row ``needed[r]``."""
    g_rowptr: np.ndarray            # int64, len(needed)+1
    g_col: np.ndarray               # int64 GLOBAL columns, G row order
    local_src: np.ndarray           # positions in my B.nzval of my own needed rows' entries
    local_dst_start: int            # where they land in G's value array (one contiguous run)
    send_rank_ids: List[int]
    send_pos: List[np.ndarray]      # positions in my B.nzval to send to each requester
    recv_rank_ids: List[int]
    recv_counts: List[int]          # entries received from each owner
    recv_dst_start: List[int]       # where each owner's run lands in G's value array


def build_row_gather_plan(needed: np.ndarray, b_row_partition: np.ndarray, b_rowptr: np.ndarray,
                          b_colidx_global, comm) -> HostRowGatherPlan:
    """``MatrixPlan(row_indices, B)`` (src/sparse.jl:579-898): ``needed`` = sorted global rows of B
    (``A.col_indices``); ``b_colidx_global(positions)`` returns the global columns at nonzero
    positions of my part of B."""
    rank, nranks = comm_rank(comm), comm_size(comm)
    needed = np.asarray(needed, dtype=np.int64)
    b_rowptr = np.asarray(b_rowptr, dtype=np.int64)
    my_start = int(b_row_partition[rank])
    owners = owner_of(b_row_partition, needed)
    bounds = np.searchsorted(owners, np.arange(nranks + 1), side="left")
    want_counts = np.diff(bounds)                                  # rows I need from each owner
    asked_counts = comm_alltoall_counts(comm, want_counts)         # rows each rank needs from me
    owners_out = [r for r in range(nranks) if r != rank and want_counts[r] > 0]
    requesters = [r for r in range(nranks) if r != rank and asked_counts[r] > 0]
    asked_rows = comm_exchange_arrays(comm, owners_out, [needed[bounds[r]:bounds[r + 1]] for r in owners_out],
                                      requesters, [int(asked_counts[r]) for r in requesters], np.int64)

    def rows_to_positions(rows_global):
        loc = rows_global - my_start
        if len(loc) and (loc.min() < 0 or loc.max() >= len(b_rowptr) - 1):
            raise ValueError("MatrixPlan: asked for a row this rank does not own")
        lens = b_rowptr[loc + 1] - b_rowptr[loc]
        starts = np.repeat(b_rowptr[loc], lens)
        within = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        return lens, starts + within

    # structure replies: row lengths, then global columns
    send_pos, reply_lens, reply_cols = [], [], []
    for rows in asked_rows:
        lens, pos = rows_to_positions(np.asarray(rows, dtype=np.int64))
        send_pos.append(pos)
        reply_lens.append(lens)
        reply_cols.append(b_colidx_global(pos))
    got_lens = comm_exchange_arrays(comm, requesters, reply_lens, owners_out,
                                    [int(want_counts[r]) for r in owners_out], np.int64)
    nnz_from = [int(np.sum(l)) for l in got_lens]
    got_cols = comm_exchange_arrays(comm, requesters, reply_cols, owners_out, nnz_from, np.int64)

    own_lens, own_pos = rows_to_positions(needed[bounds[rank]:bounds[rank + 1]])
    own_cols = b_colidx_global(own_pos)
    lens_all, cols_all, recv_start, local_dst_start, off = [], [], [], 0, 0
    it = iter(zip(got_lens, got_cols))
    for r in range(nranks):
        if r == rank:
            lens_all.append(own_lens); cols_all.append(own_cols)
            local_dst_start = off
            off += int(own_lens.sum())
        elif want_counts[r] > 0:
            l, c = next(it)
            lens_all.append(np.asarray(l, dtype=np.int64)); cols_all.append(np.asarray(c, dtype=np.int64))
            recv_start.append(off)
            off += int(np.sum(l))
    lens_cat = np.concatenate(lens_all) if lens_all else np.zeros(0, dtype=np.int64)
    g_rowptr = np.concatenate([[0], np.cumsum(lens_cat)]).astype(np.int64)
    g_col = np.concatenate(cols_all).astype(np.int64) if cols_all else np.zeros(0, dtype=np.int64)
    return HostRowGatherPlan(g_rowptr, g_col, own_pos.astype(np.int64), local_dst_start, requesters, send_pos,
                             owners_out, nnz_from, recv_start)


class MatrixPlan:
    """Cached per (hash(A), hash(B)) like ``_plan_cache`` (src/sparse.jl:900-910)."""

    def __init__(self, A, B):
        torch = _torch()
        backend = A.backend
        dev = backend.torch_device
        b_colval = B.colval.astype(np.int64)
        self.host = build_row_gather_plan(A.col_indices, B.row_partition, B.rowptr,
                                          lambda pos: B.col_indices[b_colval[pos]], backend.comm)
        h = self.host
        self.g_rowptr = torch.from_numpy(h.g_rowptr).to(dev)
        self.g_col = torch.from_numpy(h.g_col).to(dev)
        self.nnz_g = int(h.g_rowptr[-1])
        self.local_src = torch.from_numpy(h.local_src).to(dev)
        self.cache = {}
        self.halo = ctypes.c_void_p()
        self.has_halo = bool(h.send_rank_ids or h.recv_rank_ids)
        # every stored entry of B is needed, in B's own order, and none from another rank (A*A on one rank, say): the
        # gathered value array IS B.nzval -- no copy per product
        self.identity = (not self.has_halo and len(h.local_src) == self.nnz_g and int(h.local_dst_start) == 0 and
                         bool(np.array_equal(h.local_src, np.arange(self.nnz_g, dtype=h.local_src.dtype))))
        if self.has_halo:
            n_send, n_recv = len(h.send_rank_ids), len(h.recv_rank_ids)
            send_ranks = (ctypes.c_int32 * max(n_send, 1))(*h.send_rank_ids)
            send_counts = (ctypes.c_int64 * max(n_send, 1))(*[len(p) for p in h.send_pos])
            recv_ranks = (ctypes.c_int32 * max(n_recv, 1))(*h.recv_rank_ids)
            recv_counts = (ctypes.c_int64 * max(n_recv, 1))(*h.recv_counts)
            self._send_idx = (torch.from_numpy(np.concatenate(h.send_pos).astype(np.int64)).to(dev)
                              if n_send else None)
            torch.cuda.current_stream().synchronize()
            _capi.check("hpcla_halo_plan_create", _capi.load().hpcla_halo_plan_create(
                ctypes.byref(self.halo), backend.rccl, n_send, send_ranks, send_counts, dptr(self._send_idx), 1,
                n_recv, recv_ranks, recv_counts, 1))
            self._ident = torch.arange(max(h.recv_counts) if h.recv_counts else 0, dtype=torch.int64, device=dev)

    def gather_values(self, B):
        """execute_plan!(plan, B) (src/sparse.jl:922-978): G's value array on the device."""
        torch = _torch()
        h = self.host
        s = current_stream_ptr()
        if self.identity and int(B.nzval.numel()) == self.nnz_g:
            return B.nzval
        g_val = torch.empty(self.nnz_g, dtype=torch.float64, device=B.nzval.device)
        if self.has_halo:
            _capi.call("hpcla_halo_begin", self.halo, dptr(B.nzval), s)
        if len(h.local_src):
            _capi.call("hpcla_gather_f64_i64", dptr(B.nzval), dptr(self.local_src), None,
                       ctypes.c_void_p(g_val.data_ptr() + 8 * h.local_dst_start), len(h.local_src), 0, s)
        if self.has_halo:
            _capi.call("hpcla_halo_end", self.halo, s)
            ghost = ctypes.c_void_p()
            ng = ctypes.c_int64()
            _capi.call("hpcla_halo_ghost_ptr", self.halo, ctypes.byref(ghost), ctypes.byref(ng))
            off = 0
            for cnt, dst in zip(h.recv_counts, h.recv_dst_start):
                if cnt:
                    _capi.call("hpcla_gather_f64_i64", ctypes.c_void_p(ghost.value + 8 * off), dptr(self._ident),
                               None, ctypes.c_void_p(g_val.data_ptr() + 8 * dst), cnt, 0, s)
                off += cnt
        return g_val

    def destroy(self):
        if self.halo:
            _capi.call("hpcla_halo_plan_destroy", self.halo)
            self.halo = ctypes.c_void_p()


_plan_cache: Dict[tuple, MatrixPlan] = {}


def get_matrix_plan(A, B) -> MatrixPlan:
    key = (A._ensure_hash(), B._ensure_hash(), str(A.T), str(A.Ti))
    plan = _plan_cache.get(key)
    if plan is None:
        plan = _plan_cache[key] = MatrixPlan(A, B)
    return plan


def clear_matrix_plan_cache() -> None:
    for p in _plan_cache.values():
        p.destroy()
    _plan_cache.clear()


def build_product_map(A, g_rowptr, g_col, c_rowptr64, c_col, max_products: int):
    """``_build_product_map`` behind a safety net: the lists are an OPTIONAL speed-up of repeated products, built
    automatically by the third product over a structure, and the build needs ~64 B of transient device memory per
    product (several int64 temporaries of length `total`, the sort's key / permutation and its scratch).  A product that
    worked twice must not fail on its third call for that: the cap is lowered to what the device has free right now, and
    an out-of-memory (or any runtime) error during the build returns None -- the numeric kernels stay in use."""
    torch = _torch()
    try:
        free_b, _total_b = torch.cuda.mem_get_info()
        max_products = min(int(max_products), int(free_b // 64))
        return _build_product_map(A, g_rowptr, g_col, c_rowptr64, c_col, max_products)
    except (torch.cuda.OutOfMemoryError, RuntimeError) as exc:
        import sys
        sys.stderr.write(f"hpcla: SpGEMM product lists not built ({type(exc).__name__}: {str(exc)[:120]}); numeric kernels stay\n")
        torch.cuda.empty_cache()
        return None


def _build_product_map(A, g_rowptr, g_col, c_rowptr64, c_col, max_products: int):
    """Per result entry, the list of its products as (index into A.nzval, index into the gathered B values), in
    ascending A entry (= ascending k, the accumulation order of the numeric kernels): expand every A entry over its B
    row, STABLE sort by (row, result column), run lengths.  Device tensor ops (plan-time plumbing, once per structure);
    returns (pair_ptr, pairs int32 [n, 2], ptr_is_i64) or None when the lists would not fit ``max_products`` / int32, or
    -- a safety net -- when the runs do not reproduce the result structure the numeric kernels produced."""
    torch = _torch()
    dev = A.backend.torch_device
    nrows, nnz_a = A.nrows_local, A.nnz
    nnz_c = int(c_col.numel())
    if nnz_a == 0 or nnz_c == 0 or nnz_a > np.iinfo(np.int32).max or int(g_col.numel()) > np.iinfo(np.int32).max:
        return None
    a_rp = A.rowptr_target.to(torch.int64)
    k = A.colval_target().to(torch.int64)
    starts = g_rowptr[k]
    lens = g_rowptr[k + 1] - starts
    total = int(lens.sum().item())
    if total == 0 or total > max_products:
        return None
    ai = torch.repeat_interleave(torch.arange(nnz_a, device=dev, dtype=torch.int64), lens)
    excl = torch.cumsum(lens, 0) - lens
    gi = starts[ai] + (torch.arange(total, device=dev, dtype=torch.int64) - excl[ai])
    a_row = torch.repeat_interleave(torch.arange(nrows, device=dev, dtype=torch.int64), a_rp[1:] - a_rp[:-1])
    width = int(max(int(g_col.max().item()), int(c_col.max().item())) + 1)
    key = a_row[ai] * width + g_col[gi]
    del a_row, excl, starts, lens
    pairs_unsorted = torch.stack([ai, gi], dim=1).to(torch.int32)       # (the int64 lists go as soon as the pairs exist)
    del ai, gi
    key, perm = torch.sort(key, stable=True)
    uniq, counts = torch.unique_consecutive(key, return_counts=True)
    del key
    c_row = torch.repeat_interleave(torch.arange(nrows, device=dev, dtype=torch.int64), c_rowptr64[1:] - c_rowptr64[:-1])
    if int(uniq.numel()) != nnz_c or not bool(torch.equal(uniq, c_row * width + c_col)):
        return None
    del uniq, c_row
    ptr64 = total > np.iinfo(np.int32).max
    ptr = torch.zeros(nnz_c + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=ptr[1:])
    if not ptr64:
        ptr = ptr.to(torch.int32)
    pairs = pairs_unsorted[perm].contiguous()
    return ptr, pairs, ptr64


def spgemm_local(A, g_rowptr, g_col, g_val, ncols_global: int, col_partition, cache=None):
    """C_local = A_local * G on the device -> HPCSparseMatrix (rows of A, global columns of G).
    ``cache`` (a dict on the MatrixPlan) keeps everything that depends on structure only -- row bins,
    slot offsets, the result's rowptr / col_indices / colval -- so a repeated product with the same
    sparsity patterns only gathers values and reruns the numeric + compaction kernels."""
    from .sparse import HPCSparseMatrix, HPCSparseMatrix_local_device
    torch = _torch()
    lib = _capi.load()
    dev = A.backend.torch_device
    s = current_stream_ptr()
    sfx = "i64" if A.Ti == np.dtype(np.int64) else "i32"
    nrows = A.nrows_local
    a_col = A.colval_target()
    sym = cache.get("symbolic") if cache is not None else None
    if sym is None:
        ub = torch.zeros(max(nrows, 1), dtype=torch.int64, device=dev)
        _capi.call(f"hpcla_spgemm_ub_{sfx}", dptr(A.rowptr_target), dptr(a_col), nrows, 0, dptr(g_rowptr),
                   dptr(ub), s)
        ub_h = ub[:nrows].cpu().numpy()
        caps = []
        while lib.hpcla_spgemm_bin_cap(len(caps)) >= 0:          # bins are the library's to define
            caps.append(lib.hpcla_spgemm_bin_cap(len(caps)))
        # the limit is checked COLLECTIVELY (A*B is collective): every rank learns the worst row of any rank and
        # all raise together -- a rank raising alone would leave the others waiting in the next collective
        from .backends import comm_allgather
        worst = int(comm_allgather(A.backend.comm, np.array([int(ub_h.max()) if nrows else 0], dtype=np.int64)).max())
        if worst > caps[-1]:
            raise NotImplementedError(f"SpGEMM: an output row has up to {worst} candidate entries; "
                                      f"this build handles {caps[-1]}")
        ub_prefix_h = np.concatenate([[0], np.cumsum(ub_h)]).astype(np.int64)
        bins, lo = [], -1
        for b, cap in enumerate(caps):
            rows = np.flatnonzero((ub_h > lo) & (ub_h <= cap)).astype(np.int32)
            lo = cap
            if len(rows):
                bins.append((b, torch.from_numpy(rows).to(dev), len(rows)))
        sym = dict(bins=bins, ub_prefix=torch.from_numpy(ub_prefix_h).to(dev), total_ub=int(ub_prefix_h[-1]),
                   cnt=torch.zeros(max(nrows, 1), dtype=torch.int64, device=dev), result=None)
        sym["c_col_tmp"] = torch.empty(max(sym["total_ub"], 1), dtype=torch.int64, device=dev)
        sym["c_val_tmp"] = torch.empty(max(sym["total_ub"], 1), dtype=torch.float64, device=dev)
        if cache is not None:
            cache["symbolic"] = sym
    def numeric(offsets, col_out, val_out):
        for b, row_list, n in sym["bins"]:
            _capi.call(f"hpcla_spgemm_numeric_{sfx}", b, dptr(A.rowptr_target), dptr(a_col), dptr(A.nzval), 0,
                       dptr(g_rowptr), dptr(g_col), dptr(g_val), dptr(row_list), n, dptr(offsets),
                       dptr(col_out), dptr(val_out), dptr(sym["cnt"]), s)

    if sym["result"] is None:
        numeric(sym["ub_prefix"], sym["c_col_tmp"], sym["c_val_tmp"])
        cnt_h = sym["cnt"][:nrows].cpu().numpy()
        c_rowptr_h = np.concatenate([[0], np.cumsum(cnt_h)]).astype(np.int64)
        nnz_c = int(c_rowptr_h[-1])
        c_rowptr = torch.from_numpy(c_rowptr_h).to(dev)
        c_col = torch.empty(nnz_c, dtype=torch.int64, device=dev)
        c_val = torch.empty(nnz_c, dtype=torch.float64, device=dev)
        _capi.call("hpcla_spgemm_compact", dptr(c_rowptr), dptr(sym["ub_prefix"]), nrows, dptr(sym["c_col_tmp"]),
                   dptr(sym["c_val_tmp"]), dptr(c_col), dptr(c_val), s)
        C = HPCSparseMatrix_local_device(c_rowptr, c_col, c_val, ncols_global, A.backend,
                                         col_partition=col_partition)
        # c_col64: the result's global columns (for the product lists); scratch_col: the same array, where the numeric
        # kernels rewrite the same columns on a repeated product without the lists
        sym["result"] = dict(c_rowptr64=c_rowptr, nnz=nnz_c, template=C, scratch_col=c_col, c_col64=c_col)
        sym["c_col_tmp"] = sym["c_val_tmp"] = sym["ub_prefix"] = None      # upper-bound slots: first product only
        return C
    # structure known: only the values are new (same structure arrays, like conj(A), src/sparse.jl:2261-2270)
    res = sym["result"]
    T = res["template"]
    # the rows' final offsets are known, so the numeric kernels write the compacted arrays directly
    # (offsets = the result's rowptr): no upper-bound slots, no compaction pass
    c_val = torch.empty(res["nnz"], dtype=torch.float64, device=dev)
    # third product on this structure: build the per-entry product lists once (HPCLA_SPGEMM_MAP=0: never; the lists
    # cost 8 B per product of device memory, HPCLA_SPGEMM_MAP_MAX products at most, default 4e8); from then on the
    # numeric product is one streaming pass (hpcla_spgemm_numeric_mapped_f64), same bits
    res["repeats"] = res.get("repeats", 0) + 1
    if "map" not in res and res["repeats"] >= 2:        # a structure multiplied a third time will be multiplied again
        res["map"] = None
        if os.environ.get("HPCLA_SPGEMM_MAP", "1") != "0" and res["nnz"] > 0:
            res["map"] = build_product_map(A, g_rowptr, g_col, res["c_rowptr64"], res["c_col64"],
                                           int(float(os.environ.get("HPCLA_SPGEMM_MAP_MAX", "4e8"))))
    if res.get("map") is not None:
        ptr, pairs, ptr64 = res["map"]
        _capi.call("hpcla_spgemm_numeric_mapped_f64", dptr(ptr), 1 if ptr64 else 0, dptr(pairs), dptr(A.nzval), dptr(g_val),
                   dptr(c_val), res["nnz"], s)
    else:
        numeric(res["c_rowptr64"], res["scratch_col"], c_val)
    C = HPCSparseMatrix(T.row_partition, T.col_partition, T.col_indices, T._rowptr, T._colval, c_val,
                        T.rowptr_target, A.backend)
    C._colval_target = T._colval_target
    C.structural_hash = T.structural_hash
    return C


def spgemm(A, B):
    """``A * B`` for two HPCSparseMatrix (src/sparse.jl:991-1059)."""
    from .vectors import f64_only
    f64_only(A.backend, "sparse * sparse")
    assert_backends_compatible(A.backend, B.backend)
    if A.shape[1] != B.shape[0]:
        raise ValueError(f"dimension mismatch: {A.shape} * {B.shape}")
    plan = get_matrix_plan(A, B)
    g_val = plan.gather_values(B)
    C = spgemm_local(A, plan.g_rowptr, plan.g_col, g_val, B.shape[1], B.col_partition, cache=plan.cache)
    if not np.array_equal(C.row_partition, A.row_partition):
        raise ValueError("A*B: inconsistent row partition across ranks")
    return C
