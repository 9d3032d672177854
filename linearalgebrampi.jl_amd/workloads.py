"""Synthetic workloads of BASELINE.json (product-side generators; the oracle has its own, in C,
and tests compare the two).

* ``poisson2d_rows`` / ``poisson3d_rows``: the reference's ``create_2d_laplacian``
  (test/test_factorization.jl:60-102; ``idx = (j-1)*nx + i``, diagonal 4, neighbours -1, columns
  ascending within a row) and its 7-point 3-D analogue (diagonal 6), emitted for a row range with
  GLOBAL 0-based column ids -- the input form of ``HPCSparseMatrix_local``.
* ``u01``: the counter-based generator of SURVEY.md section 8d,
  ``u(seed,i) = (splitmix64(seed + 0x9E3779B97F4A7C15*(i+1)) >> 11) * 2^-53``; the device twin is
  ``hpcla_fill_uniform_f64``.
"""
from __future__ import annotations

import numpy as np

SEED_STRUCT = 0xA11CE
SEED_VALS = 0xB0B
SEED_X = 0xC0FFEE
SEED_RHS = 0xBEEF

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def u01(seed: int, idx: np.ndarray) -> np.ndarray:
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
    return (_splitmix64(z) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def _stencil_rows(idx, cand, mask, diag_col, diag_val):
    counts = mask.sum(axis=1)
    rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    colidx = cand[mask]                         # row-major flatten keeps ascending column order
    vals = np.where(np.arange(cand.shape[1])[None, :] == diag_col, diag_val, -1.0)
    vals = np.broadcast_to(vals, cand.shape)[mask].astype(np.float64)
    return rowptr, colidx.astype(np.int64), vals


def poisson2d_rows(nx: int, ny: int, row_start: int, row_end: int):
    """Rows [row_start,row_end) of the nx*ny 5-point Laplacian -> (rowptr, colidx_global, vals)."""
    idx = np.arange(row_start, row_end, dtype=np.int64)
    i, j = idx % nx, idx // nx
    cand = np.stack([idx - nx, idx - 1, idx, idx + 1, idx + nx], axis=1)
    mask = np.stack([j > 0, i > 0, np.ones_like(i, dtype=bool), i < nx - 1, j < ny - 1], axis=1)
    return _stencil_rows(idx, cand, mask, 2, 4.0)


def poisson3d_rows(nx: int, ny: int, nz: int, row_start: int, row_end: int):
    """Rows of the 7-point Laplacian, idx = (k*ny + j)*nx + i (0-based)."""
    idx = np.arange(row_start, row_end, dtype=np.int64)
    nxy = nx * ny
    i, j, k = idx % nx, (idx // nx) % ny, idx // nxy
    cand = np.stack([idx - nxy, idx - nx, idx - 1, idx, idx + 1, idx + nx, idx + nxy], axis=1)
    mask = np.stack([k > 0, j > 0, i > 0, np.ones_like(i, dtype=bool), i < nx - 1, j < ny - 1,
                     k < nz - 1], axis=1)
    return _stencil_rows(idx, cand, mask, 3, 6.0)


def spmv_algorithmic_bytes(nnz: int, nrows: int, ncols_compressed: int, index_bytes: int = 4) -> int:
    """B_alg of SURVEY.md section 8d / BASELINE.md section 2: every array touched once."""
    return nnz * (8 + index_bytes) + (nrows + 1) * index_bytes + 8 * nrows + 8 * ncols_compressed


def spmm_algorithmic_bytes(nnz: int, nrows: int, ncols_compressed: int, k: int, index_bytes: int = 4) -> int:
    return nnz * (8 + index_bytes) + (nrows + 1) * index_bytes + 8 * k * nrows + 8 * k * ncols_compressed
