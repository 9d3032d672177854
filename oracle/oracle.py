"""CPU parity oracle for the HPCLinearAlgebra.jl SpMV / SpMM / halo hot path.

TEST INFRASTRUCTURE ONLY -- importable from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product package never imports this module.

Two halves:

* floating-point arithmetic and matrix generators live in ``hpcla_oracle.c`` (built by
  ``oracle/Makefile`` with ``-ffp-contract=off``) and are reached through ctypes;
* integer/index bookkeeping (column compression, the VectorPlan neighbour lists, the gather
  semantics of ``execute_plan!``) is restated here in numpy, simulating all ranks in one
  process.  Each function cites the reference file:line it follows (relative to
  ``/root/reference``).  The only deliberate change is 1-based -> 0-based indexing.

Parity pin: see the header of ``hpcla_oracle.c``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import List

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HPCLA_ORACLE_SANITIZE=1: the AddressSanitizer + UBSan build (`make -C oracle asan`); the process must have libasan
# preloaded (tests/test_sanitizers.py does that for a child pytest)
_SANITIZE = os.environ.get("HPCLA_ORACLE_SANITIZE", "") == "1"
_LIB_PATH = os.path.join(_HERE, "_build", "libhpcla_oracle_asan.so" if _SANITIZE else "libhpcla_oracle.so")

SEED_STRUCT = 0xA11CE
SEED_VALS = 0xB0B
SEED_X = 0xC0FFEE
SEED_RHS = 0xBEEF


def build(force: bool = False) -> str:
    """Compile hpcla_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "hpcla_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _SANITIZE else []) + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, f64, vp, i32, u64 = (ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_int,
                                  ctypes.c_uint64)
        L.orc_fill_uniform.argtypes = [vp, i64, i64, u64]
        L.orc_uniform_partition.argtypes = [i64, i32, vp]
        L.orc_poisson2d_rows.argtypes = [i64, i64, i64, i64, vp, vp, vp]
        L.orc_poisson2d_rows.restype = i64
        L.orc_poisson3d_rows.argtypes = [i64, i64, i64, i64, i64, vp, vp, vp]
        L.orc_poisson3d_rows.restype = i64
        L.orc_sprand_rows.argtypes = [i64, f64, u64, u64, i64, i64, vp, vp, vp]
        L.orc_sprand_rows.restype = i64
        for nm in ("orc_spmv_i32", "orc_spmv_i64", "orc_spmv_f32_i32", "orc_spmv_f32_i64"):
            getattr(L, nm).argtypes = [vp, vp, vp, vp, vp, i64, i32, i32]
        L.orc_abs_spmv_i32.argtypes = [vp, vp, vp, vp, vp, i64, i32]
        for nm in ("orc_spmm_i32", "orc_spmm_i64", "orc_spmm_f32_i32", "orc_spmm_f32_i64"):
            getattr(L, nm).argtypes = [vp, vp, vp, vp, i64, i64, vp, i64, i64, i64, i32, i32]
        L.orc_dot_local.argtypes = [vp, vp, i64]
        L.orc_dot_local.restype = f64
        L.orc_norm_local.argtypes = [vp, i64, f64]
        L.orc_norm_local.restype = f64
        L.orc_axpy.argtypes = [f64, vp, vp, i64]
        L.orc_xpay.argtypes = [vp, f64, vp, i64]
        L.orc_gather.argtypes = [vp, vp, vp, vp, i64]
        L.orc_set_threads.argtypes = [i32]
        L.orc_max_threads.restype = i32
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------------------------------------
# inputs
# ---------------------------------------------------------------------------------------------
def fill_uniform(start: int, count: int, seed: int) -> np.ndarray:
    v = np.empty(count, dtype=np.float64)
    lib().orc_fill_uniform(_p(v), start, count, seed)
    return v


def uniform_partition(n: int, nranks: int) -> np.ndarray:
    """src/HPCLinearAlgebra.jl:279-289, 0-based boundaries."""
    part = np.empty(nranks + 1, dtype=np.int64)
    lib().orc_uniform_partition(n, nranks, _p(part))
    return part


@dataclass
class LocalRows:
    """Local rows of a matrix with GLOBAL 0-based column ids (the input of
    HPCSparseMatrix_local, src/sparse.jl:454: ``A_local.parent.rowval`` = global columns)."""
    rowptr: np.ndarray   # int64, nloc+1, 0-based
    colidx: np.ndarray   # int64, global columns, ascending within a row
    vals: np.ndarray     # float64
    ncols_global: int

    @property
    def nrows(self) -> int:
        return len(self.rowptr) - 1

    @property
    def nnz(self) -> int:
        return int(self.rowptr[-1])


def _gen(count_fill, nloc, ncols_global) -> LocalRows:
    rowptr = np.empty(nloc + 1, dtype=np.int64)
    nnz = count_fill(rowptr, None, None)
    colidx = np.empty(nnz, dtype=np.int64)
    vals = np.empty(nnz, dtype=np.float64)
    count_fill(rowptr, colidx, vals)
    return LocalRows(rowptr, colidx, vals, ncols_global)


def poisson2d_rows(nx: int, ny: int, row_start: int, row_end: int) -> LocalRows:
    """create_2d_laplacian, test/test_factorization.jl:60-102 (rows [row_start,row_end))."""
    L = lib()

    def cf(rp, ci, va):
        return L.orc_poisson2d_rows(nx, ny, row_start, row_end, _p(rp),
                                    _p(ci) if ci is not None else None,
                                    _p(va) if va is not None else None)
    return _gen(cf, row_end - row_start, nx * ny)


def poisson3d_rows(nx: int, ny: int, nz: int, row_start: int, row_end: int) -> LocalRows:
    L = lib()

    def cf(rp, ci, va):
        return L.orc_poisson3d_rows(nx, ny, nz, row_start, row_end, _p(rp),
                                    _p(ci) if ci is not None else None,
                                    _p(va) if va is not None else None)
    return _gen(cf, row_end - row_start, nx * ny * nz)


def sprand_rows(ncols: int, p: float, row_start: int, row_end: int,
                seed_struct: int = SEED_STRUCT, seed_vals: int = SEED_VALS) -> LocalRows:
    L = lib()

    def cf(rp, ci, va):
        return L.orc_sprand_rows(ncols, p, seed_struct, seed_vals, row_start, row_end, _p(rp),
                                 _p(ci) if ci is not None else None,
                                 _p(va) if va is not None else None)
    return _gen(cf, row_end - row_start, ncols)


def rows_from_coo(I, J, V, m: int, n: int, row_start: int = 0, row_end: int | None = None) -> LocalRows:
    """Julia ``sparse(I,J,V,m,n)`` (1-based I,J; duplicates summed) restricted to rows
    [row_start,row_end) (0-based) -- the slice ``A[row_start:row_end, :]`` of
    src/sparse.jl:409.  Columns ascending within a row, as Julia's CSC-of-transpose gives."""
    import scipy.sparse as sp
    A = sp.coo_matrix((np.asarray(V, dtype=np.float64),
                       (np.asarray(I) - 1, np.asarray(J) - 1)), shape=(m, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    row_end = m if row_end is None else row_end
    A = A[row_start:row_end, :]
    return LocalRows(A.indptr.astype(np.int64), A.indices.astype(np.int64),
                     A.data.astype(np.float64), n)


# ---------------------------------------------------------------------------------------------
# HPCSparseMatrix_local column compression (src/sparse.jl:501-509, 137-144)
# ---------------------------------------------------------------------------------------------
def compress_columns(rows: LocalRows):
    """col_indices = unique!(sort(copy(rowval))) (sparse.jl:501);
    colval = searchsortedfirst(col_indices, r) per nonzero (sparse.jl:142), 0-based here."""
    col_indices = np.unique(rows.colidx)                      # sorted unique
    colval = np.searchsorted(col_indices, rows.colidx)        # left = searchsortedfirst
    return col_indices.astype(np.int64), colval.astype(np.int64)


# ---------------------------------------------------------------------------------------------
# VectorPlan(A, x)  (src/sparse.jl:1875-1984), all ranks simulated in one process
# ---------------------------------------------------------------------------------------------
@dataclass
class OraclePlan:
    rank: int
    send_rank_ids: List[int] = field(default_factory=list)
    send_indices: List[np.ndarray] = field(default_factory=list)   # local idx into x.v (0-based)
    recv_rank_ids: List[int] = field(default_factory=list)
    recv_perm: List[np.ndarray] = field(default_factory=list)      # positions in gathered
    local_src_indices: np.ndarray = None
    local_dst_indices: np.ndarray = None
    n_gathered: int = 0


def owner_of(partition: np.ndarray, gidx: np.ndarray) -> np.ndarray:
    """searchsortedlast(x.partition, global_idx) - 1, clamped to nranks-1 (sparse.jl:1890-1894).
    0-based boundaries: owner = (#boundaries <= g) - 1."""
    nranks = len(partition) - 1
    own = np.searchsorted(partition, gidx, side="right") - 1
    return np.minimum(own, nranks - 1)


def vector_plans(col_indices_per_rank: List[np.ndarray], x_partition: np.ndarray) -> List[OraclePlan]:
    nranks = len(x_partition) - 1
    plans = [OraclePlan(rank=r) for r in range(nranks)]
    # step 1: group col_indices by owner, remembering (global_idx, dst_idx)
    needed = [[None] * nranks for _ in range(nranks)]   # needed[r][owner] = (globals, dst)
    for r in range(nranks):
        ci = col_indices_per_rank[r]
        own = owner_of(x_partition, ci)
        for o in range(nranks):
            sel = np.nonzero(own == o)[0]
            needed[r][o] = (ci[sel], sel)
        plans[r].n_gathered = len(ci)
    # steps 2-5: counts alltoall + index exchange; sender converts global -> local
    for r in range(nranks):
        my_start = x_partition[r]
        for o in range(nranks):              # ascending rank order == sort!(recv_rank_ids)
            g, dst = needed[r][o]
            if o != r and len(g) > 0:
                plans[r].recv_rank_ids.append(o)
                plans[r].recv_perm.append(dst.astype(np.int64))
        for q in range(nranks):              # requests received from q, ascending == sort!(send_rank_ids)
            g, _ = needed[q][r]
            if q != r and len(g) > 0:
                plans[r].send_rank_ids.append(q)
                plans[r].send_indices.append((g - my_start).astype(np.int64))
        # step 6: local elements
        g, dst = needed[r][r]
        plans[r].local_src_indices = (g - my_start).astype(np.int64)
        plans[r].local_dst_indices = dst.astype(np.int64)
    return plans


def execute_plans(plans: List[OraclePlan], x_locals: List[np.ndarray]) -> List[np.ndarray]:
    """execute_plan! (src/vectors.jl:394-463) for every rank: gathered = x[col_indices]."""
    nranks = len(plans)
    out = []
    for r in range(nranks):
        pl = plans[r]
        gathered = np.full(pl.n_gathered, np.nan)
        gathered[pl.local_dst_indices] = x_locals[r][pl.local_src_indices]       # :426-428
        for i, src_rank in enumerate(pl.recv_rank_ids):                          # :442-455
            sp = plans[src_rank]
            k = sp.send_rank_ids.index(r)
            buf = x_locals[src_rank][sp.send_indices[k]]                         # :431-439
            gathered[pl.recv_perm[i]] = buf
        out.append(gathered)
    return out


# ---------------------------------------------------------------------------------------------
# arithmetic (ctypes into hpcla_oracle.c)
# ---------------------------------------------------------------------------------------------
def spmv(rowptr: np.ndarray, colval: np.ndarray, nzval: np.ndarray, x: np.ndarray,
         base: int = 0, nthreads: int = 0) -> np.ndarray:
    """_spmv_kernel! (src/sparse.jl:2055-2066)."""
    nrows = len(rowptr) - 1
    assert rowptr.dtype == colval.dtype and rowptr.dtype in (np.int32, np.int64)
    assert nzval.dtype == x.dtype and nzval.dtype in (np.float64, np.float32)      # T of the loop: acc = zero(T)
    y = np.empty(nrows, dtype=nzval.dtype)
    rowptr, colval, nzval, x = map(np.ascontiguousarray, (rowptr, colval, nzval, x))
    sfx = ("f32_" if nzval.dtype == np.float32 else "") + ("i32" if rowptr.dtype == np.int32 else "i64")
    fn = getattr(lib(), "orc_spmv_" + sfx)
    fn(_p(rowptr), _p(colval), _p(nzval), _p(x), _p(y), nrows, base, nthreads)
    return y


def abs_spmv(rowptr, colval, nzval, x, base: int = 0) -> np.ndarray:
    nrows = len(rowptr) - 1
    y = np.empty(nrows, dtype=np.float64)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colval = np.ascontiguousarray(colval, dtype=np.int32)
    lib().orc_abs_spmv_i32(_p(rowptr), _p(colval), _p(np.ascontiguousarray(nzval)),
                           _p(np.ascontiguousarray(x)), _p(y), nrows, base)
    return y


def spmm(rowptr, colval, nzval, B: np.ndarray, base: int = 0) -> np.ndarray:
    """A * B column loop (src/sparse.jl:2391-2413).  B is (ncols_compressed, k), any strides;
    returns C (nrows, k) with the same memory order as B."""
    nrows = len(rowptr) - 1
    k = B.shape[1]
    order = "F" if B.flags.f_contiguous and not B.flags.c_contiguous else "C"
    assert nzval.dtype == B.dtype and B.dtype in (np.float64, np.float32)
    C = np.empty((nrows, k), dtype=B.dtype, order=order)
    rowptr, colval, nzval = map(np.ascontiguousarray, (rowptr, colval, nzval))
    sfx = ("f32_" if B.dtype == np.float32 else "") + ("i32" if rowptr.dtype == np.int32 else "i64")
    fn = getattr(lib(), "orc_spmm_" + sfx)
    es = B.itemsize
    fn(_p(rowptr), _p(colval), _p(nzval), _p(B), B.strides[0] // es, B.strides[1] // es,
       _p(C), C.strides[0] // es, C.strides[1] // es, nrows, k, base)
    return C


def dot(x_locals: List[np.ndarray], y_locals: List[np.ndarray]) -> float:
    """dot (src/vectors.jl:798-812): local dot then allreduce(+) (rank order here)."""
    s = 0.0
    for xv, yv in zip(x_locals, y_locals):
        xv, yv = np.ascontiguousarray(xv), np.ascontiguousarray(yv)
        s += lib().orc_dot_local(_p(xv), _p(yv), len(xv))
    return s


def norm(x_locals: List[np.ndarray], p: float = 2.0) -> float:
    """norm (src/vectors.jl:758-780)."""
    parts = []
    for xv in x_locals:
        xv = np.ascontiguousarray(xv)
        parts.append(lib().orc_norm_local(_p(xv), len(xv), float(p)))
    if p == 2:
        return float(np.sqrt(np.sum(parts)))
    if p == 1:
        return float(np.sum(parts))
    if np.isinf(p):
        return float(np.max(parts))
    return float(np.sum(parts) ** (1.0 / p))


def axpy(alpha: float, x: np.ndarray, y: np.ndarray) -> None:
    lib().orc_axpy(float(alpha), _p(x), _p(y), len(x))


def xpay(x: np.ndarray, beta: float, y: np.ndarray) -> None:
    lib().orc_xpay(_p(x), float(beta), _p(y), len(x))


def cg(rowptr, colval, nzval, b: np.ndarray, iters: int):
    """Textbook CG, exactly `iters` iterations, no convergence exit.
    Not in the reference (section 3.4): composed from A*p, dot, axpy-style broadcasts.
    Single-rank restatement (identity plan).  Returns (x, residual-norm history)."""
    n = len(b)
    x = np.zeros(n)
    r = b.copy()
    p = r.copy()
    rr = dot([r], [r])
    hist = [float(np.sqrt(rr))]
    for _ in range(iters):
        Ap = spmv(rowptr, colval, nzval, p)
        pAp = dot([p], [Ap])
        alpha = rr / pAp
        axpy(alpha, p, x)
        axpy(-alpha, Ap, r)
        rr_new = dot([r], [r])
        beta = rr_new / rr
        xpay(r, beta, p)
        rr = rr_new
        hist.append(float(np.sqrt(rr)))
    return x, hist


def spgemm(a_rowptr, a_col, a_val, g_rowptr, g_col, g_val, ncols: int):
    """Local sparse x sparse product in the reference's order (src/sparse.jl:991-1059 + SparseArrays
    Gustavson): returns (c_rowptr, c_col_global, c_val), columns ascending."""
    L = lib()
    L.orc_spgemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] + \
        [ctypes.c_void_p] * 3
    L.orc_spgemm.restype = ctypes.c_int64
    a_rowptr, a_col, g_rowptr, g_col = (np.ascontiguousarray(v, dtype=np.int64) for v in (a_rowptr, a_col, g_rowptr, g_col))
    a_val, g_val = np.ascontiguousarray(a_val, dtype=np.float64), np.ascontiguousarray(g_val, dtype=np.float64)
    nrows = len(a_rowptr) - 1
    c_rowptr = np.empty(nrows + 1, dtype=np.int64)
    nnz = L.orc_spgemm(_p(a_rowptr), _p(a_col), _p(a_val), nrows, _p(g_rowptr), _p(g_col), _p(g_val), ncols,
                       _p(c_rowptr), None, None)
    c_col = np.empty(nnz, dtype=np.int64)
    c_val = np.empty(nnz, dtype=np.float64)
    L.orc_spgemm(_p(a_rowptr), _p(a_col), _p(a_val), nrows, _p(g_rowptr), _p(g_col), _p(g_val), ncols,
                 _p(c_rowptr), _p(c_col), _p(c_val))
    return c_rowptr, c_col, c_val
