#!/usr/bin/env python3
"""Generate tests/golden/pin_large.npz: the oracle's pin at the sizes the REFERENCE ITSELF defines.

`hotpath_golden.json` (make_golden.py) restates the reference's unit-test inputs, n <= 12.  The reference also
defines two larger hot-path inputs, in its benchmark drivers:

* ``laplacian_2d_sparse(n)`` with n = 10 000 -- tools/benchmark_vs_petsc.jl:42-49: ``grid = round(Int, sqrt(n))``,
  ``L1D = spdiagm(-1 => -e[1:end-1], 0 => 2e, 1 => -e[1:end-1])``, ``L2D = kron(I, L1D) + kron(L1D, I)`` -- the matrix
  behind the ONLY number the reference publishes for this path (A*A, tools/benchmark_vs_petsc_results.txt:3-11);
* ``generate_sparse(n, Float64; nnz_per_row = 10)`` with n = 1000 -- tools/benchmark_single_rank.jl:48-71: per row
  ``randperm(n)[1:nnz_per_row]`` columns with ``randn()`` values, then ``A + A'``.  Julia's RNG stream cannot be
  reproduced here; the SHAPE is restated with the counter-based generator of SURVEY 8d (splitmix64): a partial
  Fisher-Yates draw of 10 distinct columns per row, Box-Muller values, and ``A + A'`` with ONE fp64 addition where
  both (i, j) and (j, i) are stored (what SparseArrays' ``+`` does).

Expected outputs: exact rational arithmetic on the exact fp64 inputs, rounded ONCE to fp64 -- independent of the
oracle and of the HIP kernels -- for x = 1..n (integer-valued on the Laplacian: every partial sum is exact, so ANY
summation order must reproduce it bit for bit) and x = u01(SEED_X, i) (the oracle / kernels are then held to
gamma_k * (|A||x|)_i, k = the row's length, and to BASELINE's 1e-12 relative).  Also the exact product A*A of the
Laplacian (small integers: exact in every order).

The file holds DATA only (inputs as COO triplets, expected outputs).  Run:  python tests/golden/make_golden_large.py
"""
import math
import os
from fractions import Fraction

import numpy as np

MASK = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15
SEED_X = 0xC0FFEE            # SURVEY 8d seeds (restated, not imported: this script must not depend on oracle/)
SEED_STRUCT = 0xA11CE
SEED_VALS = 0xB0B


def splitmix64(z):
    """the splitmix64 finaliser (the increment is the caller's: u01 below)"""
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def u01(seed, i):
    """u(seed, i) = (splitmix64(seed + GOLDEN * (i + 1)) >> 11) * 2^-53"""
    return (splitmix64((seed + GOLDEN * (i + 1)) & MASK) >> 11) * 2.0 ** -53


def laplacian_2d_sparse(n):
    """tools/benchmark_vs_petsc.jl:42-49, as 1-based COO triplets in CSR order (row-major, ascending columns --
    the order HPCSparseMatrix stores a row in: A^T held as CSC, src/sparse.jl:319-337)."""
    g = max(1, int(round(math.sqrt(n))))
    rows = []
    for b in range(g):               # kron(I_g, L1D): block b, inner index a; kron(L1D, I_g): couples blocks b +- 1
        for a in range(g):
            i = b * g + a
            ent = {}
            ent[i] = 2.0 + 2.0       # kron(I, L1D)[i, i] + kron(L1D, I)[i, i]: one fp64 addition
            if a > 0:
                ent[i - 1] = -1.0
            if a < g - 1:
                ent[i + 1] = -1.0
            if b > 0:
                ent[i - g] = -1.0
            if b < g - 1:
                ent[i + g] = -1.0
            rows.append(sorted(ent.items()))
    return g * g, rows


def generate_sparse(n, nnz_per_row=10):
    """tools/benchmark_single_rank.jl:48-71 restated with the counter RNG (see the module docstring)."""
    A = [dict() for _ in range(n)]
    for i in range(n):
        rs = splitmix64(SEED_STRUCT ^ ((0xD1B54A32D192ED03 * (i + 1)) & MASK))
        vs = splitmix64(SEED_VALS ^ ((0xD1B54A32D192ED03 * (i + 1)) & MASK))
        perm = list(range(n))
        for t in range(min(nnz_per_row, n)):                  # randperm(n)[1:ncols]: partial Fisher-Yates
            j = t + int(u01(rs, t) * (n - t))
            perm[t], perm[j] = perm[j], perm[t]
            u1, u2 = u01(vs, 2 * t), u01(vs, 2 * t + 1)
            A[i][perm[t]] = math.sqrt(-2.0 * math.log(1.0 - u1)) * math.cos(2.0 * math.pi * u2)     # randn()
    rows = []
    for i in range(n):                                        # A + A'
        ent = dict(A[i])
        for j in range(n):
            if i in A[j]:
                ent[j] = (ent[j] + A[j][i]) if j in ent else A[j][i]       # one fp64 add where both are stored
        rows.append(sorted(ent.items()))
    return n, rows


def exact_matvec(rows, x):
    xf = [Fraction(v) for v in x]
    return np.array([float(sum(Fraction(a) * xf[j] for j, a in r)) for r in rows], dtype=np.float64)


def coo(rows):
    I = np.array([i + 1 for i, r in enumerate(rows) for _ in r], dtype=np.int32)
    J = np.array([j + 1 for r in rows for j, _ in r], dtype=np.int32)
    V = np.array([a for r in rows for _, a in r], dtype=np.float64)
    return I, J, V


def exact_square(rows):
    """C = A*A with every C(i, j) summed exactly (Fractions), rounded once; CSR order."""
    out = []
    for r in rows:
        acc = {}
        for k, a in r:
            fa = Fraction(a)
            for j, b in rows[k]:
                acc[j] = acc.get(j, Fraction(0)) + fa * Fraction(b)
        out.append(sorted((j, float(v)) for j, v in acc.items()))
    return out


def main():
    out = {}
    n, lap = laplacian_2d_sparse(10_000)
    assert n == 10_000
    I, J, V = coo(lap)
    x_int = np.arange(1, n + 1, dtype=np.float64)
    x_u = np.array([u01(SEED_X, i) for i in range(n)], dtype=np.float64)
    out.update(lap_n=np.int64(n), lap_I=I, lap_J=J, lap_V=V,
               lap_y_int=exact_matvec(lap, x_int), lap_x_u01=x_u, lap_y_u01=exact_matvec(lap, x_u))
    sq = exact_square(lap)
    CI, CJ, CV = coo(sq)
    out.update(lap_sq_I=CI, lap_sq_J=CJ, lap_sq_V=CV)

    n2, gs = generate_sparse(1000, 10)
    I, J, V = coo(gs)
    x_int = np.arange(1, n2 + 1, dtype=np.float64)
    x_u = np.array([u01(SEED_X, i) for i in range(n2)], dtype=np.float64)
    out.update(gs_n=np.int64(n2), gs_I=I, gs_J=J, gs_V=V,
               gs_y_int=exact_matvec(gs, x_int), gs_x_u01=x_u, gs_y_u01=exact_matvec(gs, x_u))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pin_large.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()},
          f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
