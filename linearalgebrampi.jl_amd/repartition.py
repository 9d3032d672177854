"""``repartition(x, p)`` for HPCVector, HPCMatrix and HPCSparseMatrix on DeviceROCm.

Reference: ``VectorRepartitionPlan`` + ``execute_plan!`` (src/vectors.jl:470-722),
``DenseRepartitionPlan`` (src/dense.jl:1551-1810), ``SparseRepartitionPlan`` (src/sparse.jl:4069-4600),
memoized in ``_repartition_plan_cache`` (src/HPCLinearAlgebra.jl:143-145).  The reference stages every
repartition through host arrays (``_ensure_cpu`` -> MPI Isend/Irecv -> ``_values_to_backend``).

All three are the same exchange of CONTIGUOUS ranges (a rank's rows that overlap another rank's target
range are one run, and so are their stored entries), so here the data never leaves the GPUs:
``hpcla_exchange_ranges_f64`` issues one group of RCCL send/recv straight between ``x.v`` (or the
row-major dense block, or ``nzval``) and the result buffer, plus one device copy for the part that
stays.  The lists are the reference plan's own fields, 0-based.  The reference obtains the receive
counts with an Alltoall (src/vectors.jl:555-557); both partitions are known on every rank, so the
counts are computed locally -- same lists, no collective.  Only the sparse plan communicates at plan
time (row lengths and global column ids of the moving rows, host p2p like the reference's structure
exchange, src/sparse.jl:4180-4330).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _capi
from .backends import comm_exchange_arrays, comm_rank, comm_size
from .partition import compute_partition_hash
from .vectors import HPCVector, current_stream_ptr, dptr


def _torch():
    import torch
    return torch


def check_partition(p, n: int, nranks: int) -> np.ndarray:
    """A valid target partition: nranks+1 non-decreasing offsets from 0 to n (the reference's
    1-based ``p[1] == 1``, ``p[end] == n+1``, src/vectors.jl:699-700)."""
    p = np.asarray(p, dtype=np.int64)
    if p.ndim != 1 or len(p) != nranks + 1 or p[0] != 0 or p[-1] != n or np.any(np.diff(p) < 0):
        raise ValueError(f"repartition: invalid partition {p.tolist()} for length {n} on {nranks} ranks")
    return p


class RangePlan:
    """The send/recv/local lists shared by the three reference plans, in ROW (or element) units.

    ``send_rank_ids`` / ``send_ranges`` (local start, count): src/vectors.jl:532-550, 574-580;
    ``recv_rank_ids`` / ``recv_counts`` / ``recv_offsets`` (offset into the result): :584-600;
    ``local_src_start`` / ``local_count`` / ``local_dst_offset``: :564-572."""

    def __init__(self, src_partition: np.ndarray, p: np.ndarray, rank: int):
        nranks = len(p) - 1
        s0, s1 = int(src_partition[rank]), int(src_partition[rank + 1])
        d0, d1 = int(p[rank]), int(p[rank + 1])
        self.send_rank_ids: List[int] = []
        self.send_ranges: List[Tuple[int, int]] = []
        self.recv_rank_ids: List[int] = []
        self.recv_counts: List[int] = []
        self.recv_offsets: List[int] = []
        self.local_src_start, self.local_count, self.local_dst_offset = 0, 0, 0
        for r in range(nranks):
            lo, hi = max(s0, int(p[r])), min(s1, int(p[r + 1]))          # what I own of r's target
            if hi > lo:
                if r == rank:
                    self.local_src_start, self.local_count, self.local_dst_offset = lo - s0, hi - lo, lo - d0
                else:
                    self.send_rank_ids.append(r)
                    self.send_ranges.append((lo - s0, hi - lo))
            lo, hi = max(d0, int(src_partition[r])), min(d1, int(src_partition[r + 1]))   # what r owns of mine
            if hi > lo and r != rank:
                self.recv_rank_ids.append(r)
                self.recv_counts.append(hi - lo)
                self.recv_offsets.append(lo - d0)
        self.result_partition = p.copy()
        self.result_partition_hash = compute_partition_hash(p)
        self.result_local_size = d1 - d0


def _carr(ctype, values):
    values = [int(v) for v in values]
    return (ctype * max(1, len(values)))(*values)


def exchange_ranges(backend, src, dst, send_ranks, send_offsets, send_counts, recv_ranks, recv_offsets,
                    recv_counts, local_src, local_dst, local_count, width: int = 1) -> None:
    """One call of ``hpcla_exchange_ranges_f64`` on the current stream (device tensors src/dst)."""
    _capi.call("hpcla_exchange_ranges_f64", backend.rccl, dptr(src), dptr(dst),
               len(send_ranks), _carr(ctypes.c_int, send_ranks), _carr(ctypes.c_int64, send_offsets),
               _carr(ctypes.c_int64, send_counts),
               len(recv_ranks), _carr(ctypes.c_int, recv_ranks), _carr(ctypes.c_int64, recv_offsets),
               _carr(ctypes.c_int64, recv_counts),
               int(local_src), int(local_dst), int(local_count), int(width), current_stream_ptr())


def _execute_rows(plan: RangePlan, backend, src, dst, width: int) -> None:
    exchange_ranges(backend, src, dst, plan.send_rank_ids, [s for s, _ in plan.send_ranges],
                    [c for _, c in plan.send_ranges], plan.recv_rank_ids, plan.recv_offsets,
                    plan.recv_counts, plan.local_src_start, plan.local_dst_offset, plan.local_count, width)


# memoization: (source hash, target hash, kind) -> plan   (src/HPCLinearAlgebra.jl:143-145)
_repartition_plan_cache: Dict[tuple, object] = {}


def clear_repartition_cache() -> None:
    _repartition_plan_cache.clear()


# ---- vectors (src/vectors.jl:470-722) -----------------------------------------------------------------
def get_vector_repartition_plan(x: HPCVector, p: np.ndarray) -> RangePlan:
    key = (x.structural_hash, compute_partition_hash(p), "vector")
    plan = _repartition_plan_cache.get(key)
    if plan is None:
        plan = RangePlan(x.partition, p, comm_rank(x.backend.comm))
        _repartition_plan_cache[key] = plan
    return plan


def repartition_vector(x: HPCVector, p) -> HPCVector:
    from .vectors import f64_only
    f64_only(x.backend, "repartition")
    nranks = comm_size(x.backend.comm)
    p = check_partition(p, len(x), nranks)
    if np.array_equal(x.partition, p):                               # fast path, src/vectors.jl:713-716
        return x
    plan = get_vector_repartition_plan(x, p)
    torch = _torch()
    out = torch.empty(plan.result_local_size, dtype=torch.float64, device=x.v.device)
    _execute_rows(plan, x.backend, x.v, out, 1)
    return HPCVector(plan.result_partition_hash, plan.result_partition, out, x.backend)


# ---- dense rows (src/dense.jl:1551-1810) --------------------------------------------------------------
def repartition_dense(A, p):
    from .vectors import f64_only
    f64_only(A.backend, "repartition")
    from .dense import HPCMatrix
    nranks = comm_size(A.backend.comm)
    p = check_partition(p, int(A.row_partition[-1]), nranks)
    if np.array_equal(A.row_partition, p):
        return A
    key = (compute_partition_hash(A.row_partition), compute_partition_hash(p), "dense")
    plan = _repartition_plan_cache.get(key)
    if plan is None:
        plan = RangePlan(A.row_partition, p, comm_rank(A.backend.comm))
        _repartition_plan_cache[key] = plan
    torch = _torch()
    k = int(A.A.shape[1])
    out = torch.empty((plan.result_local_size, k), dtype=torch.float64, device=A.A.device)
    if k > 0:
        _execute_rows(plan, A.backend, A.A.contiguous(), out, k)     # row-major on the pitch k: a row range is one run
    return HPCMatrix(plan.result_partition, A.col_partition, out, A.backend)


# ---- sparse rows (src/sparse.jl:4069-4600) ------------------------------------------------------------
class SparseRepartitionPlan:
    """Row lists as RangePlan; the result STRUCTURE (rowptr, compressed columns, col_indices) is built
    eagerly at plan time from the exchanged row lengths and global column ids
    (``result_AT`` / ``result_col_indices`` / ``compress_map`` of the reference plan); per execution
    only ``nzval`` moves: the entries of a row range are one contiguous run
    (``send_nnz_counts`` / ``recv_nnz_counts`` / ``recv_value_offsets`` / ``local_value_offset``)."""

    def __init__(self, A, p: np.ndarray):
        from .sparse import _compress_columns
        comm = A.backend.comm
        rank = comm_rank(comm)
        self.rows = RangePlan(A.row_partition, p, rank)
        rp = A.rowptr.astype(np.int64)
        gcol = A.col_indices[A.colval.astype(np.int64)] if len(A.colval) else np.zeros(0, np.int64)
        rows = self.rows
        n_new = rows.result_local_size
        # structure exchange: per moving row its length, then its global column ids
        send_lens = [np.diff(rp[s:s + c + 1]) for s, c in rows.send_ranges]
        recv_lens = comm_exchange_arrays(comm, rows.send_rank_ids, send_lens, rows.recv_rank_ids,
                                         rows.recv_counts, np.int64)
        self.send_value_offsets = [int(rp[s]) for s, _ in rows.send_ranges]
        self.send_nnz_counts = [int(rp[s + c] - rp[s]) for s, c in rows.send_ranges]
        self.recv_nnz_counts = [int(l.sum()) for l in recv_lens]
        send_cols = [gcol[o:o + c] for o, c in zip(self.send_value_offsets, self.send_nnz_counts)]
        recv_cols = comm_exchange_arrays(comm, rows.send_rank_ids, send_cols, rows.recv_rank_ids,
                                         self.recv_nnz_counts, np.int64)
        lens = np.zeros(n_new, dtype=np.int64)
        for off, l in zip(rows.recv_offsets, recv_lens):
            lens[off:off + len(l)] = l
        ls, lc, ld = rows.local_src_start, rows.local_count, rows.local_dst_offset
        lens[ld:ld + lc] = np.diff(rp[ls:ls + lc + 1])
        new_rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        new_gcol = np.empty(int(new_rp[-1]), dtype=np.int64)
        self.recv_value_offsets = [int(new_rp[off]) for off in rows.recv_offsets]
        for voff, cols in zip(self.recv_value_offsets, recv_cols):
            new_gcol[voff:voff + len(cols)] = cols
        self.local_value_src = int(rp[ls])
        self.local_nnz = int(rp[ls + lc] - rp[ls])
        self.local_value_offset = int(new_rp[ld]) if n_new else 0
        new_gcol[self.local_value_offset:self.local_value_offset + self.local_nnz] = \
            gcol[self.local_value_src:self.local_value_src + self.local_nnz]
        Ti = A.backend.Ti.type
        if len(new_gcol) > np.iinfo(Ti).max:
            raise OverflowError("nnz does not fit the backend index type")
        self.result_col_indices, self.result_colval = _compress_columns(new_gcol, int(A.col_partition[-1]), Ti)
        self.result_rowptr = new_rp.astype(Ti)
        self.result_row_partition = rows.result_partition
        self.result_col_partition = A.col_partition.copy()
        self.result_rowptr_dev = None                                # uploaded once, shared by results

    def execute(self, A):
        from .sparse import HPCSparseMatrix
        torch = _torch()
        dev = A.nzval.device
        out = torch.empty(int(self.result_rowptr[-1]), dtype=torch.float64, device=dev)
        rows = self.rows
        exchange_ranges(A.backend, A.nzval, out, rows.send_rank_ids, self.send_value_offsets,
                        self.send_nnz_counts, rows.recv_rank_ids, self.recv_value_offsets,
                        self.recv_nnz_counts, self.local_value_src, self.local_value_offset,
                        self.local_nnz, 1)
        if self.result_rowptr_dev is None:
            self.result_rowptr_dev = torch.from_numpy(self.result_rowptr).to(dev)
        return HPCSparseMatrix(self.result_row_partition, self.result_col_partition,
                               self.result_col_indices, self.result_rowptr, self.result_colval, out,
                               self.result_rowptr_dev, A.backend)


def get_sparse_repartition_plan(A, p: np.ndarray) -> SparseRepartitionPlan:
    key = (A._ensure_hash(), compute_partition_hash(p), "sparse", str(A.Ti))
    plan = _repartition_plan_cache.get(key)
    if plan is None:
        plan = SparseRepartitionPlan(A, p)
        _repartition_plan_cache[key] = plan
    return plan


def repartition_sparse(A, p):
    from .vectors import f64_only
    f64_only(A.backend, "repartition")
    nranks = comm_size(A.backend.comm)
    p = check_partition(p, int(A.row_partition[-1]), nranks)
    if np.array_equal(A.row_partition, p):                           # src/sparse.jl:4591-4594
        return A
    return get_sparse_repartition_plan(A, p).execute(A)


def repartition(obj, p):
    """``repartition(x, p)`` for HPCVector / HPCMatrix / HPCSparseMatrix; ``p`` is the 0-based
    partition (nranks+1 offsets, ``p[0] == 0``, ``p[-1] == n``).  Returns ``obj`` itself when the
    partition is unchanged, like the reference."""
    from .dense import HPCMatrix
    from .sparse import HPCSparseMatrix
    if isinstance(obj, HPCVector):
        return repartition_vector(obj, p)
    if isinstance(obj, HPCMatrix):
        return repartition_dense(obj, p)
    if isinstance(obj, HPCSparseMatrix):
        return repartition_sparse(obj, p)
    raise TypeError(f"repartition: unsupported operand {type(obj).__name__}")
