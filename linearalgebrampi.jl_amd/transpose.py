"""``transpose(A) * x`` ("next" rank 1).

Reference semantics (src/sparse.jl:2375-2379, 2136-2142, 1846-1862): ``transpose(A)`` is a lazy
wrapper; multiplying it materialises ``A^T`` as an ``HPCSparseMatrix`` through a ``TransposePlan``
(src/sparse.jl:1519-1865: every nonzero A[i,j] travels to the rank that owns row j of A^T, i.e. the
owner of column j in ``A.col_partition``), caches it bidirectionally (``A.cached_transpose``), and
then runs the ordinary ``A^T * x`` path.  The materialised transpose has
``row_partition = A.col_partition`` and ``col_partition = A.row_partition``; within a row its entries
are in ascending column order, so the device SpMV sums them in exactly the reference's order.

The STRUCTURE of the redistribution is plan-time host work (numpy + the comm_* primitives), like the
reference's and memoized like its TransposePlan; the VALUES move GPU to GPU (two gather kernels and one
RCCL range exchange); the multiply itself is the DeviceROCm SpMV.
"""
from __future__ import annotations

import numpy as np

from .backends import comm_alltoall_counts, comm_exchange_arrays, comm_rank, comm_size
from .partition import owner_of


class HostTransposeStructure:
    """Structure half of the TransposePlan (src/sparse.jl:1519-1700), host numpy: which stored entries
    travel where and in which order they end up.  Input = this rank's rows of A (CSR, GLOBAL 0-based
    columns); result = this rank's rows of A^T (rows ``col_partition[rank]:col_partition[rank+1]``),
    columns (= global rows of A) ascending in each row.

    ``send_order``  permutation of the local entries into send order (stable by destination rank)
    ``bounds``      send order is cut per destination rank at ``bounds[r]:bounds[r+1]``
    ``peers_out`` / ``peers_in`` / ``recv_counts``   neighbours and message sizes
    ``final_order`` permutation of [own segment | segments received from peers_in, ascending rank] into
                    the result's CSR order;  ``rowptr_t`` / ``col_t``  the result structure"""

    def __init__(self, rowptr, colidx_global, row_partition, col_partition, comm):
        rank, nranks = comm_rank(comm), comm_size(comm)
        rowptr = np.asarray(rowptr, dtype=np.int64)
        colidx_global = np.asarray(colidx_global, dtype=np.int64)
        nloc = len(rowptr) - 1
        gi = np.repeat(np.arange(nloc, dtype=np.int64) + int(row_partition[rank]), np.diff(rowptr))
        dest = owner_of(col_partition, colidx_global)
        self.send_order = np.argsort(dest, kind="stable")
        dest_s, gj_s, gi_s = dest[self.send_order], colidx_global[self.send_order], gi[self.send_order]
        self.bounds = np.searchsorted(dest_s, np.arange(nranks + 1), side="left")
        send_counts = np.diff(self.bounds)
        recv_counts = comm_alltoall_counts(comm, send_counts)
        self.rank = rank
        self.peers_out = [r for r in range(nranks) if r != rank and send_counts[r] > 0]
        self.peers_in = [r for r in range(nranks) if r != rank and recv_counts[r] > 0]
        self.recv_counts = [int(recv_counts[r]) for r in self.peers_in]
        seg = lambda a, r: a[self.bounds[r]:self.bounds[r + 1]]
        got_j = comm_exchange_arrays(comm, self.peers_out, [seg(gj_s, r) for r in self.peers_out], self.peers_in,
                                     self.recv_counts, np.int64)
        got_i = comm_exchange_arrays(comm, self.peers_out, [seg(gi_s, r) for r in self.peers_out], self.peers_in,
                                     self.recv_counts, np.int64)
        tj = np.concatenate([seg(gj_s, rank)] + got_j)
        ti = np.concatenate([seg(gi_s, rank)] + got_i)
        my_start = int(col_partition[rank])
        n_rows_t = int(col_partition[rank + 1]) - my_start
        row_t = tj - my_start
        if len(row_t) and (row_t.min() < 0 or row_t.max() >= n_rows_t):
            raise ValueError("transpose: received an entry this rank does not own")
        self.final_order = np.lexsort((ti, row_t))            # by row of A^T, then ascending column
        self.col_t = ti[self.final_order]
        counts = np.bincount(row_t, minlength=n_rows_t) if n_rows_t else np.zeros(0, dtype=np.int64)
        self.rowptr_t = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.n_own_segment = int(self.bounds[rank + 1] - self.bounds[rank])
        self.n_total = len(ti)


class TransposePlan:
    """``TransposePlan`` (src/sparse.jl:1519-1865) for DeviceROCm: the structure above plus its device
    form.  ``execute`` moves only VALUES, GPU to GPU: gather into send order, one RCCL range exchange
    (``hpcla_exchange_ranges_f64``: every destination's entries are one contiguous run), gather into
    the result order -- the reference stages all three steps through host arrays
    (src/sparse.jl:1702-1845).  With one rank the two gathers fuse into one."""

    def __init__(self, A):
        from .sparse import _compress_columns
        torch = _torch()
        backend = A.backend
        dev = backend.torch_device
        gcol = A.col_indices[A.colval.astype(np.int64)] if A.nnz else np.zeros(0, dtype=np.int64)
        st = HostTransposeStructure(A.rowptr.astype(np.int64), gcol, A.row_partition, A.col_partition,
                                    backend.comm)
        self.st = st
        self.single = not st.peers_out and not st.peers_in
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)
        if self.single:
            self.perm = up(st.send_order[st.bounds[st.rank]:st.bounds[st.rank + 1]][st.final_order])
        else:
            self.send_order = up(st.send_order)
            self.final_order = up(st.final_order)
        Ti = backend.Ti.type
        if st.n_total > np.iinfo(Ti).max:
            raise OverflowError("nnz does not fit the backend index type")
        self.col_indices, self.colval = _compress_columns(st.col_t, int(A.row_partition[-1]), Ti)
        self.rowptr_ti = st.rowptr_t.astype(Ti)
        self.rowptr_dev = torch.from_numpy(self.rowptr_ti).to(dev)
        self.row_partition = A.col_partition.copy()
        self.col_partition = A.row_partition.copy()

    def execute(self, A):
        from .repartition import exchange_ranges
        from .sparse import HPCSparseMatrix
        from .vectors import current_stream_ptr, dptr
        from . import _capi
        torch = _torch()
        st, dev = self.st, A.nzval.device
        out = torch.empty(st.n_total, dtype=torch.float64, device=dev)
        s = current_stream_ptr()
        if self.single:
            _capi.call("hpcla_gather_f64_i64", dptr(A.nzval), dptr(self.perm), None, dptr(out), st.n_total, 0, s)
        else:
            send = torch.empty(A.nnz, dtype=torch.float64, device=dev)
            _capi.call("hpcla_gather_f64_i64", dptr(A.nzval), dptr(self.send_order), None, dptr(send), A.nnz, 0, s)
            recv = torch.empty(st.n_total, dtype=torch.float64, device=dev)
            offs = np.concatenate([[st.n_own_segment], st.n_own_segment + np.cumsum(st.recv_counts)])[:-1]
            exchange_ranges(A.backend, send, recv, st.peers_out, [int(st.bounds[r]) for r in st.peers_out],
                            [int(st.bounds[r + 1] - st.bounds[r]) for r in st.peers_out], st.peers_in,
                            [int(o) for o in offs], st.recv_counts, int(st.bounds[st.rank]), 0, st.n_own_segment, 1)
            _capi.call("hpcla_gather_f64_i64", dptr(recv), dptr(self.final_order), None, dptr(out), st.n_total, 0, s)
        return HPCSparseMatrix(self.row_partition, self.col_partition, self.col_indices, self.rowptr_ti,
                               self.colval, out, self.rowptr_dev, A.backend)


_transpose_plan_cache = {}


def get_transpose_plan(A) -> TransposePlan:
    key = (A._ensure_hash(), str(A.Ti))
    plan = _transpose_plan_cache.get(key)
    if plan is None:
        plan = _transpose_plan_cache[key] = TransposePlan(A)
    return plan


def clear_transpose_plan_cache() -> None:
    _transpose_plan_cache.clear()


def _torch():
    import torch
    return torch


class TransposedHPCSparseMatrix:
    """Lazy ``transpose(A)`` (``Transpose(A)``, src/sparse.jl:2254-2258)."""

    def __init__(self, parent):
        self.parent = parent

    @property
    def shape(self):
        m, n = self.parent.shape
        return n, m

    def materialize(self):
        """``HPCSparseMatrix(transpose(A))`` (src/sparse.jl:1846-1862), cached bidirectionally."""
        from .vectors import f64_only
        f64_only(self.parent.backend, "transpose(A) (materialised)")
        A = self.parent
        if getattr(A, "cached_transpose", None) is not None:
            return A.cached_transpose
        Y = get_transpose_plan(A).execute(A)
        if not np.array_equal(Y.row_partition, A.col_partition):
            raise ValueError("transpose: inconsistent column partition across ranks")
        A.cached_transpose = Y
        Y.cached_transpose = A
        return Y

    def __matmul__(self, x):
        return self.materialize() @ x

    __mul__ = __matmul__


class TransposedHPCVector:
    """``transpose(v)`` / ``v'`` for a real HPCVector (``Transpose{T,HPCVector}``, src/vectors.jl:735-746):
    a lazy row-vector view; ``.parent`` is the column vector.  Supports the reference's row-vector
    algebra: ``vt @ A`` (src/sparse.jl:2136-2142, src/dense.jl:1270-1274), ``vt @ w`` (the inner product ``dot``),
    ``a*vt``, ``vt*a``, ``vt/a``, ``vt ± wt``, ``-vt`` (src/vectors.jl:909-940, 969-987)."""

    def __init__(self, parent):
        self.parent = parent

    @property
    def shape(self):
        return (1, len(self.parent))

    def __matmul__(self, other):
        from .dense import HPCMatrix, dense_matvec_t
        from .sparse import HPCSparseMatrix
        from .vectors import HPCVector, dot
        if isinstance(other, HPCMatrix):                             # src/dense.jl:1270-1274
            return TransposedHPCVector(dense_matvec_t(other, self.parent))
        if isinstance(other, HPCSparseMatrix):                       # transpose(transpose(A) * v)
            return TransposedHPCVector(TransposedHPCSparseMatrix(other) @ self.parent)
        if isinstance(other, TransposedHPCSparseMatrix):             # vt * transpose(A) = transpose(A * v)
            return TransposedHPCVector(other.parent @ self.parent)
        if isinstance(other, HPCVector):
            return dot(self.parent, other)
        return NotImplemented

    def __mul__(self, a):
        if isinstance(a, (int, float, np.floating, np.integer)):
            return TransposedHPCVector(self.parent * float(a))
        return self.__matmul__(a)

    def __rmul__(self, a):
        return TransposedHPCVector(self.parent * float(a))

    def __truediv__(self, a):
        return TransposedHPCVector(self.parent / float(a))

    def __add__(self, other):
        if not isinstance(other, TransposedHPCVector):
            return NotImplemented
        return TransposedHPCVector(self.parent + other.parent)

    def __sub__(self, other):
        if not isinstance(other, TransposedHPCVector):
            return NotImplemented
        return TransposedHPCVector(self.parent - other.parent)

    def __neg__(self):
        return TransposedHPCVector(-self.parent)


def transpose(A):
    """``transpose(x)``: lazy for HPCSparseMatrix and HPCVector; the transpose of a lazy transpose is the
    parent (the element type is real, so ``adjoint`` is the same thing)."""
    from .dense import HPCMatrix, TransposedHPCMatrix
    from .vectors import HPCVector
    if isinstance(A, (TransposedHPCSparseMatrix, TransposedHPCVector, TransposedHPCMatrix)):
        return A.parent
    if isinstance(A, HPCVector):
        return TransposedHPCVector(A)
    if isinstance(A, HPCMatrix):
        return TransposedHPCMatrix(A)
    return TransposedHPCSparseMatrix(A)


adjoint = transpose
