"""``A + B`` / ``A - B`` for two HPCSparseMatrix (src/sparse.jl:1072-1494).

Reference: a memoized ``AdditionPlan`` merges the two sparsity patterns on the host (per-row sorted
union of global columns, structural zeros preserved) and five index-mapped kernels fill the result:
A-only entries copied, B-only entries copied (or negated), shared entries added (or subtracted)
(src/sparse.jl:1258-1375).  Here the plan is numpy (same union) and the value pass is ONE kernel over the
merged pattern, ``hpcla_merge_combine_f64_*`` (a plain axpby when the two patterns are identical) -- entries
present on one side are copied, shared entries take one rounding, so results are bit-identical.

Like the reference (src/sparse.jl:1407, 1456) B is first repartitioned to A's row partition when the
two differ (repartition.py: structure on the host at plan time, values device to device).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import _capi
from .backends import assert_backends_compatible
from .vectors import current_stream_ptr, dptr


def _torch():
    import torch
    return torch


class AdditionPlan:
    def __init__(self, A, B):
        from .sparse import _compress_columns
        torch = _torch()
        dev = A.backend.torch_device
        ncols = int(A.col_partition[-1])
        W = ncols + 1
        ra, rb = A.rowptr.astype(np.int64), B.rowptr.astype(np.int64)
        n = A.nrows_local
        key_a = np.repeat(np.arange(n, dtype=np.int64), np.diff(ra)) * W + A.col_indices[A.colval.astype(np.int64)]
        key_b = np.repeat(np.arange(n, dtype=np.int64), np.diff(rb)) * W + B.col_indices[B.colval.astype(np.int64)]
        union = np.union1d(key_a, key_b)                      # sorted: by row, then global column
        pos_a, pos_b = np.searchsorted(union, key_a), np.searchsorted(union, key_b)
        inv_a = np.full(len(union), -1, dtype=np.int64); inv_a[pos_a] = np.arange(len(key_a))
        inv_b = np.full(len(union), -1, dtype=np.int64); inv_b[pos_b] = np.arange(len(key_b))
        # device lists: position of each merged entry in A.nzval / B.nzval, -1 where absent
        # (the reference keeps three index groups -- A only, B only, both -- for its five kernels,
        # src/sparse.jl:1172-1245; the same information, one pass)
        self.idx64 = max(len(key_a), len(key_b)) > np.iinfo(np.int32).max
        idt = np.int64 if self.idx64 else np.int32
        self.same_pattern = len(key_a) == len(key_b) == len(union)   # identical structure: no lists needed
        if self.same_pattern:
            self.inv_a = self.inv_b = None
        else:
            self.inv_a = torch.from_numpy(inv_a.astype(idt)).to(dev)
            self.inv_b = torch.from_numpy(inv_b.astype(idt)).to(dev)
        self.nnz = len(union)
        rows_c = union // W
        self.rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows_c, minlength=n))]).astype(np.int64)
        cols_glob = union % W
        Ti = A.backend.Ti.type
        self.col_indices, self.colval = _compress_columns(cols_glob, ncols, Ti)
        self.rowptr_ti = self.rowptr.astype(Ti)
        self.rowptr_dev = torch.from_numpy(self.rowptr_ti).to(dev)
        self.colval_dev = None


_addition_plan_cache: Dict[tuple, AdditionPlan] = {}


def clear_addition_plan_cache() -> None:
    _addition_plan_cache.clear()


def _get_addition_plan(A, B) -> AdditionPlan:
    key = (A._ensure_hash(), B._ensure_hash(), str(A.Ti))          # src/sparse.jl:1384-1395
    plan = _addition_plan_cache.get(key)
    if plan is None:
        plan = _addition_plan_cache[key] = AdditionPlan(A, B)
    return plan


def sparse_add(A, B, subtract: bool = False):
    """``A + B`` (src/sparse.jl:1405-1445) / ``A - B`` (:1454-1494)."""
    from .vectors import f64_only
    f64_only(A.backend, "A + B / A - B for sparse operands")
    from .sparse import HPCSparseMatrix
    torch = _torch()
    assert_backends_compatible(A.backend, B.backend)
    if A.shape != B.shape:
        raise ValueError(f"dimension mismatch: {A.shape} vs {B.shape}")
    if not np.array_equal(A.row_partition, B.row_partition):
        from .repartition import repartition_sparse
        B = repartition_sparse(B, A.row_partition)                  # src/sparse.jl:1407, 1456
    plan = _get_addition_plan(A, B)
    s = current_stream_ptr()
    nzval = torch.empty(plan.nnz, dtype=torch.float64, device=A.backend.torch_device)
    if plan.same_pattern:
        # identical patterns (the common case: two operators on one mesh): a plain streaming
        # 1*a + (+-1)*b, 24 B per entry; multiplication by +-1 is exact, so the bits equal a +- b
        _capi.call("hpcla_axpby_f64", 1.0, dptr(A.nzval), -1.0 if subtract else 1.0, dptr(B.nzval), dptr(nzval),
                   plan.nnz, s)
    else:
        _capi.call("hpcla_merge_combine_f64_i64" if plan.idx64 else "hpcla_merge_combine_f64_i32", dptr(nzval),
                   dptr(A.nzval), dptr(plan.inv_a), dptr(B.nzval), dptr(plan.inv_b), plan.nnz,
                   1 if subtract else 0, s)
    C = HPCSparseMatrix(A.row_partition, A.col_partition, plan.col_indices, plan.rowptr_ti, plan.colval, nzval,
                        plan.rowptr_dev, A.backend)
    return C


# ---- A + lambda*I / A - lambda*I (src/sparse.jl:3925-4015) ---------------------------------------------
_identity_cache: Dict[tuple, object] = {}


def add_scaled_identity(A, lam: float, subtract: bool = False):
    """``A + lam*I`` (``A + J::UniformScaling``, src/sparse.jl:3925-3985; ``A - J`` :3987-3995).  The
    reference walks the entries on the CPU with scalar indexing (its comment: "not efficient on GPU");
    here the identity is an ordinary HPCSparseMatrix on A's row partition (unit diagonal, cached per
    structure), scaled on the device and merged by the same one-pass AdditionPlan as ``A + B`` -- diagonal
    entries missing from A are added structurally, exactly as in the reference's IdentityAdditionPlan."""
    from .vectors import f64_only
    f64_only(A.backend, "A + lambda*I")
    from .backends import comm_rank
    from .sparse import HPCSparseMatrix_local
    m, n = A.shape
    if m != n:
        raise ValueError(f"A + lambda*I needs a square matrix, got {A.shape}")
    key = (A.backend.torch_device, tuple(A.row_partition.tolist()), tuple(A.col_partition.tolist()), str(A.Ti))
    eye = _identity_cache.get(key)
    if eye is None:
        r = comm_rank(A.backend.comm)
        lo, hi = int(A.row_partition[r]), int(A.row_partition[r + 1])
        eye = HPCSparseMatrix_local(np.arange(hi - lo + 1, dtype=np.int64), np.arange(lo, hi, dtype=np.int64),
                                    np.ones(hi - lo), n, A.backend, col_partition=A.col_partition)
        _identity_cache[key] = eye
    return sparse_add(A, float(lam) * eye, subtract=subtract)


def clear_identity_cache() -> None:
    _identity_cache.clear()
