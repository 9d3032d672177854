"""``transpose(A) * x`` ("next" rank 1).

Reference semantics (src/sparse.jl:2375-2379, 2136-2142, 1846-1862): ``transpose(A)`` is a lazy
wrapper; multiplying it materialises ``A^T`` as an ``HPCSparseMatrix`` through a ``TransposePlan``
(src/sparse.jl:1519-1865: every nonzero A[i,j] travels to the rank that owns row j of A^T, i.e. the
owner of column j in ``A.col_partition``), caches it bidirectionally (``A.cached_transpose``), and
then runs the ordinary ``A^T * x`` path.  The materialised transpose has
``row_partition = A.col_partition`` and ``col_partition = A.row_partition``; within a row its entries
are in ascending column order, so the device SpMV sums them in exactly the reference's order.

The redistribution is plan-time host work (numpy + the comm_* primitives), like the reference's; the
multiply itself is the DeviceROCm SpMV.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from .backends import comm_alltoall_counts, comm_exchange_arrays, comm_rank, comm_size
from .partition import owner_of


def transpose_local_rows(rowptr: np.ndarray, colidx_global: np.ndarray, vals: np.ndarray,
                         row_partition: np.ndarray, col_partition: np.ndarray, comm
                         ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Host redistribution of the TransposePlan: input = this rank's rows of A (CSR, GLOBAL 0-based
    columns); output = this rank's rows of A^T (rows ``col_partition[rank]:col_partition[rank+1]``)
    as ``(rowptr, colidx_global, vals)`` with columns (= global rows of A) ascending in each row."""
    rank, nranks = comm_rank(comm), comm_size(comm)
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx_global = np.asarray(colidx_global, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    nloc = len(rowptr) - 1
    gi = np.repeat(np.arange(nloc, dtype=np.int64) + int(row_partition[rank]), np.diff(rowptr))
    dest = owner_of(col_partition, colidx_global)
    order = np.argsort(dest, kind="stable")
    dest_s, gj_s, gi_s, v_s = dest[order], colidx_global[order], gi[order], vals[order]
    bounds = np.searchsorted(dest_s, np.arange(nranks + 1), side="left")
    send_counts = np.diff(bounds)
    recv_counts = comm_alltoall_counts(comm, send_counts)
    peers_out = [r for r in range(nranks) if r != rank and send_counts[r] > 0]
    peers_in = [r for r in range(nranks) if r != rank and recv_counts[r] > 0]
    seg = lambda a, r: a[bounds[r]:bounds[r + 1]]
    got_j = comm_exchange_arrays(comm, peers_out, [seg(gj_s, r) for r in peers_out], peers_in,
                                 [int(recv_counts[r]) for r in peers_in], np.int64)
    got_i = comm_exchange_arrays(comm, peers_out, [seg(gi_s, r) for r in peers_out], peers_in,
                                 [int(recv_counts[r]) for r in peers_in], np.int64)
    got_v = comm_exchange_arrays(comm, peers_out, [seg(v_s, r) for r in peers_out], peers_in,
                                 [int(recv_counts[r]) for r in peers_in], np.float64)
    tj = np.concatenate([seg(gj_s, rank)] + got_j)
    ti = np.concatenate([seg(gi_s, rank)] + got_i)
    tv = np.concatenate([seg(v_s, rank)] + got_v)
    my_start = int(col_partition[rank])
    n_rows_t = int(col_partition[rank + 1]) - my_start
    row_t = tj - my_start
    if len(row_t) and (row_t.min() < 0 or row_t.max() >= n_rows_t):
        raise ValueError("transpose: received an entry this rank does not own")
    order = np.lexsort((ti, row_t))                    # by row of A^T, then ascending column
    row_t, ti, tv = row_t[order], ti[order], tv[order]
    rowptr_t = np.zeros(n_rows_t + 1, dtype=np.int64)
    np.add.at(rowptr_t, row_t + 1, 1)
    return np.cumsum(rowptr_t), ti, tv


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
        from .sparse import HPCSparseMatrix_local
        A = self.parent
        if getattr(A, "cached_transpose", None) is not None:
            return A.cached_transpose
        vals = A.nzval.detach().cpu().numpy()
        colidx_global = A.col_indices[A.colval.astype(np.int64)]
        rp, ci, v = transpose_local_rows(A.rowptr.astype(np.int64), colidx_global, vals, A.row_partition,
                                         A.col_partition, A.backend.comm)
        Y = HPCSparseMatrix_local(rp, ci, v, int(A.row_partition[-1]), A.backend,
                                  col_partition=A.row_partition)
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
