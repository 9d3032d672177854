#!/usr/bin/env python3
"""Generate tests/golden/hotpath_golden.json.

The reference (Julia) cannot run here and its tests hold no literal expected outputs: every
hot-path test builds a small closed-form input and compares with Julia's own ``A*x`` on the
global matrix to 1e-10 absolute (test/test_utils.jl:154-157).  This script therefore restates
those INPUTS (citing the reference test that defines them) and computes the expected outputs
with exact rational arithmetic (``fractions.Fraction`` on the exact fp64 input values),
rounded once to fp64 -- independent of both the oracle and the HIP kernels.

Run:  python tests/golden/make_golden.py   (writes hotpath_golden.json next to this file)
"""
import json
import os
from decimal import Decimal, getcontext
from fractions import Fraction

getcontext().prec = 60


def F(x):
    return Fraction(float(x))


def dense_from_coo(I, J, V, m, n):
    """Julia sparse(I,J,V,m,n): duplicates are summed."""
    A = [[Fraction(0)] * n for _ in range(m)]
    for i, j, v in zip(I, J, V):
        A[i - 1][j - 1] += F(v)
    return A


def matvec(A, x):
    return [float(sum(a * F(xj) for a, xj in zip(row, x) if a != 0)) for row in A]


def matmat(A, B):
    k = len(B[0])
    return [[float(sum(a * F(B[j][c]) for j, a in enumerate(row) if a != 0)) for c in range(k)]
            for row in A]


cases = {}

# --- test/test_vector_multiplication.jl:42-65 and :70-92 (mul!) ---------------------------------
# tridiagonal_matrix(T, 8): test/test_utils.jl:90-100 ; test_vector: :124-130
n = 8
I = list(range(1, n + 1)) + list(range(1, n)) + list(range(2, n + 1))
J = list(range(1, n + 1)) + list(range(2, n + 1)) + list(range(1, n))
V = [2.0] * n + [-0.5] * (n - 1) + [-0.5] * (n - 1)
x = [float(i) for i in range(1, n + 1)]
cases["spmv_tridiagonal"] = dict(
    ref="test/test_vector_multiplication.jl:42-65,70-92; test/test_utils.jl:90-100,124-130",
    m=n, n=n, I=I, J=J, V=V, x=x, y=matvec(dense_from_coo(I, J, V, n, n), x))

# --- test/test_vector_multiplication.jl:95-118 non-square ---------------------------------------
m, k = 6, 8
I = [1, 2, 3, 4, 5, 6, 1, 2, 3, 4]
J = [1, 2, 3, 4, 5, 6, 7, 8, 1, 2]
V = [float(i) for i in range(1, len(I) + 1)]
x = [float(i) for i in range(1, k + 1)]
cases["spmv_nonsquare"] = dict(
    ref="test/test_vector_multiplication.jl:95-118",
    m=m, n=k, I=I, J=J, V=V, x=x, y=matvec(dense_from_coo(I, J, V, m, k), x))

# --- test/test_local_constructors.jl:214-231 -----------------------------------------------------
I = [1, 2, 3, 4, 5, 6, 1, 3, 5, 7, 9]
J = [1, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2]
V = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 0.5, 0.5, 0.5, 0.5, 0.5]
x = [float(i) for i in range(1, 9)]
cases["spmv_local_ctor"] = dict(
    ref="test/test_local_constructors.jl:214-231",
    m=10, n=8, I=I, J=J, V=V, x=x, y=matvec(dense_from_coo(I, J, V, 10, 8), x))

# --- test/test_new_operations.jl:43-59, 79-82  SpMM ---------------------------------------------
n, mcols = 8, 6
Iv = [1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 3, 4, 5, 6, 7, 8]
Jv = [1, 2, 3, 4, 5, 6, 7, 8, 2, 3, 4, 5, 6, 7, 8, 1]
Vv = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
# A = S + transpose(S) + 2I  (in fp64: each entry is a sum of fp64 values, Julia rounds each +)
S = dense_from_coo(Iv, Jv, Vv, n, n)
A = [[None] * n for _ in range(n)]
for i in range(n):
    for j in range(n):
        t = float(S[i][j]) + float(S[j][i])                 # S + S'
        t = t + (2.0 if i == j else 0.0)                    # + 2I
        A[i][j] = Fraction(t)
I2, J2, V2 = [], [], []
for i in range(n):
    for j in range(n):
        if A[i][j] != 0:
            I2.append(i + 1); J2.append(j + 1); V2.append(float(A[i][j]))
B = [[float(i + j * 0.1) for j in range(1, mcols + 1)] for i in range(1, n + 1)]
C = matmat(A, B)
fro = Decimal(0)
for row in C:
    for v in row:
        fro += Decimal(v) * Decimal(v)
cases["spmm_sym"] = dict(
    ref="test/test_new_operations.jl:43-59,79-82",
    m=n, n=n, I=I2, J=J2, V=V2, B=B, C=C, C_fro=float(fro.sqrt()))

# --- test/test_new_operations.jl:139-147  dot ---------------------------------------------------
xl = [float(i) + 0.1 for i in range(1, 9)]
yl = [float(i) + 0.1 for i in range(8, 0, -1)]
cases["dot"] = dict(
    ref="test/test_new_operations.jl:59-63,139-147",
    x=xl, y=yl,
    dot_xy=float(sum(F(a) * F(b) for a, b in zip(xl, yl))),
    dot_xx=float(sum(F(a) * F(a) for a in xl)))

# --- test/test_vector_multiplication.jl:163-195  norms ------------------------------------------
xs = [float(i) for i in range(1, 11)]
d = [Decimal(v) for v in xs]
cases["norms"] = dict(
    ref="test/test_vector_multiplication.jl:163-195",
    x=xs,
    norm2=float(sum(v * v for v in d).sqrt()),
    norm1=float(sum(abs(v) for v in d)),
    norminf=float(max(abs(v) for v in d)),
    norm3=float((sum(v ** 3 for v in d)).ln().__truediv__(Decimal(3)).exp()),
    norm1p5=float((sum((v.ln() * Decimal("1.5")).exp() for v in d)).ln().__truediv__(Decimal("1.5")).exp()))

# --- test/test_vector_multiplication.jl:228-309  u+v, u-v, -v, a*v, v/a ; test_utils.jl:137-145 --
u = [float(i) for i in range(1, 9)]
v = [float(i) for i in range(8, 0, -1)]
cases["vector_ops"] = dict(
    ref="test/test_vector_multiplication.jl:228-309; test/test_utils.jl:137-145",
    u=u, v=v,
    add=[a + b for a, b in zip(u, v)], sub=[a - b for a, b in zip(u, v)],
    neg=[-a for a in v], scale=3.5, scaled=[3.5 * a for a in v], divided=[a / 2.0 for a in v])

# --- test/test_sparse_api.jl:257-308 broadcast  dest .= v .* 2 .+ w .^ 2 --------------------------
vv = [float(i) for i in range(1, 11)]
ww = [float(i) for i in range(11, 21)]
cases["broadcast"] = dict(
    ref="test/test_sparse_api.jl:257-308",
    v=vv, w=ww, add=[a + b for a, b in zip(vv, ww)],
    fused=[a * 2.0 + b * b for a, b in zip(vv, ww)])

# --- create_2d_laplacian(T, nx, ny): test/test_factorization.jl:60-102 -----------------------------
# Fixture = the (I,J,V) triplets that loop emits (closed form), for nx=4, ny=3 and nx=3, ny=5.
def lap2d(nx, ny):
    I, J, V = [], [], []
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            idx = (j - 1) * nx + i
            I.append(idx); J.append(idx); V.append(4.0)
            if i > 1:
                I.append(idx); J.append(idx - 1); V.append(-1.0)
            if i < nx:
                I.append(idx); J.append(idx + 1); V.append(-1.0)
            if j > 1:
                I.append(idx); J.append(idx - nx); V.append(-1.0)
            if j < ny:
                I.append(idx); J.append(idx + nx); V.append(-1.0)
    return I, J, V


for nx, ny in ((4, 3), (3, 5)):
    I, J, V = lap2d(nx, ny)
    nn = nx * ny
    x = [float(i) * 0.25 for i in range(1, nn + 1)]
    cases[f"laplacian2d_{nx}x{ny}"] = dict(
        ref="test/test_factorization.jl:60-102", nx=nx, ny=ny, m=nn, n=nn, I=I, J=J, V=V,
        x=x, y=matvec(dense_from_coo(I, J, V, nn, nn), x))

# --- test/test_matrix_multiplication.jl:38-88  sparse x sparse ---------------------------------------
def matmat_sparse(Aop, Bop):
    """exact product of two dense-of-Fraction matrices; keep only structurally possible entries"""
    m, k, n2 = len(Aop), len(Bop), len(Bop[0])
    C = [[None] * n2 for _ in range(m)]
    for i in range(m):
        for j in range(n2):
            terms = [Aop[i][t] * Bop[t][j] for t in range(k) if Aop[i][t] != 0 and Bop[t][j] != 0]
            if terms:
                C[i][j] = float(sum(terms))
    return C


n = 8
IA = list(range(1, n + 1)) + list(range(1, n)) + list(range(2, n + 1))
JA = list(range(1, n + 1)) + list(range(2, n + 1)) + list(range(1, n))
VA = [2.0] * n + [-0.5] * (n - 1) + [-0.5] * (n - 1)
VB = [1.5] * n + [0.25] * (n - 1) + [0.25] * (n - 1)
Cd = matmat_sparse(dense_from_coo(IA, JA, VA, n, n), dense_from_coo(IA, JA, VB, n, n))
cases["spgemm_tridiagonal"] = dict(
    ref="test/test_matrix_multiplication.jl:38-61", m=n, k=n, n=n, IA=IA, JA=JA, VA=VA, IB=IA, JB=JA, VB=VB,
    C=[[(j + 1, v) for j, v in enumerate(row) if v is not None] for row in Cd])
IA2 = [1, 2, 3, 4, 5, 6, 1, 2, 3, 4]; JA2 = [1, 2, 3, 4, 5, 6, 7, 8, 1, 2]
VA2 = [float(i) for i in range(1, 11)]
IB2 = [1, 2, 3, 4, 5, 6, 7, 8, 1, 3]; JB2 = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
VB2 = [float(i) for i in range(1, 11)]
Cd2 = matmat_sparse(dense_from_coo(IA2, JA2, VA2, 6, 8), dense_from_coo(IB2, JB2, VB2, 8, 10))
cases["spgemm_nonsquare"] = dict(
    ref="test/test_matrix_multiplication.jl:64-88", m=6, k=8, n=10, IA=IA2, JA=JA2, VA=VA2, IB=IB2, JB=JB2, VB=VB2,
    C=[[(j + 1, v) for j, v in enumerate(row) if v is not None] for row in Cd2])

# --- widened rows: transposes, A +/- B, D' * W * D chains --------------------------------
def transpose_d(A):
    return [list(r) for r in zip(*A)]


def add_d(A, B, sign=1):
    return [[a + sign * b for a, b in zip(ra, rb)] for ra, rb in zip(A, B)]


def mul_d(A, B):
    return [[sum(a * B[j][c] for j, a in enumerate(row) if a != 0) for c in range(len(B[0]))] for row in A]


def coo_of(A):
    I, J, V = [], [], []
    for i, row in enumerate(A):
        for j, v in enumerate(row):
            if v != 0:
                I.append(i + 1); J.append(j + 1); V.append(float(v))
    return I, J, V


def dense_f(A):
    return [[float(v) for v in row] for row in A]


def spdiagm(n, diags):
    A = [[Fraction(0)] * n for _ in range(n)]
    for off, vals in diags.items():
        for t, v in enumerate(vals):
            i, j = (t, t + off) if off >= 0 else (t - off, t)
            A[i][j] = F(v)
    return A


# test/test_new_operations.jl:43-76: transpose(A_sparse) * x with the spmm_sym matrix, x = (1:8) .+ 0.1
xt = [float(i) + 0.1 for i in range(1, 9)]
cases["transpose_spmv"] = dict(
    ref="test/test_new_operations.jl:43-76", m=8, n=8, I=I2, J=J2, V=V2, x=xt,
    y=[float(sum(A[j][i] * F(xt[j]) for j in range(8))) for i in range(8)])

# test/test_addition_different_sparsity.jl:41-62: tridiagonal + (diagonal, second superdiagonal)
n8 = 8
Aa = spdiagm(n8, {-1: [1.0] * 7, 0: [2.0] * 8, 1: [1.0] * 7})
Bb = spdiagm(n8, {0: [3.0] * 8, 2: [0.5] * 6})
IAa, JAa, VAa = coo_of(Aa)
IBb, JBb, VBb = coo_of(Bb)
cases["add_different_sparsity"] = dict(
    ref="test/test_addition_different_sparsity.jl:41-62 (A+B); A-B by the same inputs (src/sparse.jl:1454-1494)",
    n=n8, IA=IAa, JA=JAa, VA=VAa, IB=IBb, JB=JBb, VB=VBb,
    sum=dense_f(add_d(Aa, Bb)), diff=dense_f(add_d(Aa, Bb, -1)))

# test/test_addition_different_sparsity.jl:65-118: D' * W * D products and their sums
dx = spdiagm(n8, {0: [-1.0] * 8, 1: [1.0] * 7})
dx[7][7] = Fraction(0)
idm = spdiagm(n8, {0: [1.0] * 8})
Wm = spdiagm(n8, {0: [0.5] * 8})
M1 = mul_d(mul_d(transpose_d(idm), Wm), dx)
M2 = mul_d(mul_d(transpose_d(dx), Wm), idm)
Hh = add_d(mul_d(mul_d(transpose_d(dx), Wm), dx), mul_d(mul_d(transpose_d(idm), Wm), idm))
Idx, Jdx, Vdx = coo_of(dx)
cases["dtwd_products"] = dict(
    ref="test/test_addition_different_sparsity.jl:65-118",
    n=n8, Idx=Idx, Jdx=Jdx, Vdx=Vdx, w=[0.5] * 8,
    M_sum=dense_f(add_d(M1, M2)), H=dense_f(Hh))

# --- uniform_partition docstring example: src/HPCLinearAlgebra.jl:269-277 ------------------------
cases["uniform_partition"] = dict(
    ref="src/HPCLinearAlgebra.jl:269-289",
    examples=[dict(n=10, nranks=4, partition_1based=[1, 4, 7, 9, 11])])

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_golden.json")
with open(out, "w") as f:
    json.dump(cases, f, indent=1)
print("wrote", out, "cases:", ", ".join(cases))
