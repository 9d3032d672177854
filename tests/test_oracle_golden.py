"""Pin the CPU oracle to the reference's own closed-form test inputs (tests/golden/).

Expected values come from exact rational arithmetic (tests/golden/make_golden.py), so they are
independent of the oracle.  Tolerance: the reference's 1e-10 absolute (test/test_utils.jl:154-157);
most cases are exactly representable and are compared bit-for-bit.
"""
import math

import numpy as np
import pytest

TOL = 1e-10   # test/test_utils.jl:154-157


def _rows(orc, case, row_start=0, row_end=None):
    return orc.rows_from_coo(case["I"], case["J"], case["V"], case["m"], case["n"], row_start, row_end)


def _spmv_via_oracle(orc, case, nranks, Ti):
    """Full reference pipeline on `nranks` simulated ranks: partition -> local rows -> column
    compression -> VectorPlan -> execute_plan! -> _spmv_kernel!."""
    m, n = case["m"], case["n"]
    rp = orc.uniform_partition(m, nranks)
    xp = orc.uniform_partition(n, nranks)
    x = np.array(case["x"])
    locals_, cis = [], []
    for r in range(nranks):
        rows = _rows(orc, case, int(rp[r]), int(rp[r + 1]))
        ci, cv = orc.compress_columns(rows)
        locals_.append((rows, cv))
        cis.append(ci)
    plans = orc.vector_plans(cis, xp)
    xl = [x[xp[r]:xp[r + 1]] for r in range(nranks)]
    gathered = orc.execute_plans(plans, xl)
    y = []
    for r in range(nranks):
        rows, cv = locals_[r]
        assert not np.isnan(gathered[r]).any()
        np.testing.assert_array_equal(gathered[r], x[cis[r]])
        y.append(orc.spmv(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals, gathered[r]))
    return np.concatenate(y)


@pytest.mark.parametrize("name", ["spmv_tridiagonal", "spmv_nonsquare", "spmv_local_ctor",
                                  "laplacian2d_4x3", "laplacian2d_3x5"])
@pytest.mark.parametrize("nranks", [1, 2, 3])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_spmv_golden(orc, golden, name, nranks, Ti):
    case = golden[name]
    y = _spmv_via_oracle(orc, case, nranks, Ti)
    assert np.max(np.abs(y - np.array(case["y"]))) < TOL
    np.testing.assert_array_equal(y, np.array(case["y"]))   # all cases are exactly representable


def test_spmv_one_based(orc, golden):
    case = golden["spmv_tridiagonal"]
    rows = _rows(orc, case)
    ci, cv = orc.compress_columns(rows)
    y = orc.spmv((rows.rowptr + 1).astype(np.int64), (cv + 1).astype(np.int64), rows.vals,
                 np.array(case["x"])[ci], base=1)
    np.testing.assert_array_equal(y, np.array(case["y"]))


@pytest.mark.parametrize("order", ["C", "F"])
def test_spmm_golden(orc, golden, order):
    case = golden["spmm_sym"]
    rows = _rows(orc, case)
    ci, cv = orc.compress_columns(rows)
    B = np.array(case["B"], order=order)
    C = orc.spmm(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals, B[ci])
    Cg = np.array(case["C"])
    assert np.max(np.abs(C - Cg)) < TOL
    assert abs(np.linalg.norm(C) - case["C_fro"]) < TOL
    # column loop semantics: each column equals one SpMV of that column (src/sparse.jl:2400-2403)
    for k in range(B.shape[1]):
        yk = orc.spmv(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals,
                      np.ascontiguousarray(B[ci, k]))
        np.testing.assert_array_equal(C[:, k], yk)


def test_dot_norm_golden(orc, golden):
    d = golden["dot"]
    x, y = np.array(d["x"]), np.array(d["y"])
    for nranks in (1, 2, 3):
        p = orc.uniform_partition(len(x), nranks)
        xl = [x[p[r]:p[r + 1]] for r in range(nranks)]
        yl = [y[p[r]:p[r + 1]] for r in range(nranks)]
        assert abs(orc.dot(xl, yl) - d["dot_xy"]) < TOL
        assert abs(orc.dot(xl, xl) - d["dot_xx"]) < TOL
    nm = golden["norms"]
    v = np.array(nm["x"])
    for nranks in (1, 2, 4):
        p = orc.uniform_partition(len(v), nranks)
        vl = [v[p[r]:p[r + 1]] for r in range(nranks)]
        assert abs(orc.norm(vl, 2) - nm["norm2"]) < TOL
        assert abs(orc.norm(vl, 1) - nm["norm1"]) < TOL
        assert abs(orc.norm(vl, math.inf) - nm["norminf"]) < TOL
        assert abs(orc.norm(vl, 3) - nm["norm3"]) < TOL
        assert abs(orc.norm(vl, 1.5) - nm["norm1p5"]) < TOL


def test_vector_ops_golden(orc, golden):
    c = golden["vector_ops"]
    u, v = np.array(c["u"]), np.array(c["v"])
    y = u.copy(); orc.axpy(1.0, v, y); np.testing.assert_array_equal(y, c["add"])
    y = u.copy(); orc.axpy(-1.0, v, y); np.testing.assert_array_equal(y, c["sub"])
    b = golden["broadcast"]
    vv, ww = np.array(b["v"]), np.array(b["w"])
    y = ww * ww; orc.axpy(2.0, vv, y)        # dest .= v .* 2 .+ w .^ 2
    np.testing.assert_array_equal(y, b["fused"])
    y = vv.copy(); orc.xpay(ww, 1.0, y); np.testing.assert_array_equal(y, b["add"])


def test_uniform_partition_golden(orc, golden, hp):
    for ex in golden["uniform_partition"]["examples"]:
        want = np.array(ex["partition_1based"]) - 1
        np.testing.assert_array_equal(orc.uniform_partition(ex["n"], ex["nranks"]), want)
        np.testing.assert_array_equal(hp.uniform_partition(ex["n"], ex["nranks"]), want)


@pytest.mark.parametrize("nx,ny", [(4, 3), (3, 5)])
def test_poisson2d_generator_matches_reference_loop(orc, golden, nx, ny):
    """orc.poisson2d_rows == sparse(I,J,V) of create_2d_laplacian (test/test_factorization.jl:60-102)."""
    case = golden[f"laplacian2d_{nx}x{ny}"]
    ref = _rows(orc, case)
    for lo, hi in ((0, nx * ny), (2, nx * ny - 3)):
        got = orc.poisson2d_rows(nx, ny, lo, hi)
        want = _rows(orc, case, lo, hi)
        np.testing.assert_array_equal(got.rowptr, want.rowptr)
        np.testing.assert_array_equal(got.colidx, want.colidx)
        np.testing.assert_array_equal(got.vals, want.vals)
    assert ref.nnz == 5 * nx * ny - 2 * nx - 2 * ny


def test_poisson3d_generator(orc):
    import scipy.sparse as sp
    nx, ny, nz = 3, 4, 5
    got = orc.poisson3d_rows(nx, ny, nz, 0, nx * ny * nz)
    ex, ey, ez = np.ones(nx), np.ones(ny), np.ones(nz)
    T = lambda n: sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    K = (sp.kron(sp.eye(nz), sp.kron(sp.eye(ny), T(nx))) + sp.kron(sp.eye(nz), sp.kron(T(ny), sp.eye(nx)))
         + sp.kron(T(nz), sp.kron(sp.eye(ny), sp.eye(nx)))).tocsr()
    K.sort_indices()
    np.testing.assert_array_equal(got.rowptr, K.indptr)
    np.testing.assert_array_equal(got.colidx, K.indices)
    np.testing.assert_array_equal(got.vals, K.data)
    assert got.nnz == 7 * nx * ny * nz - 2 * (nx * ny + ny * nz + nx * nz)


def test_sprand_generator_statistics(orc):
    n, p = 10_000, 0.01
    rows = orc.sprand_rows(n, p, 0, n)
    # BASELINE configs[0]: sprand(10^4,10^4,0.01): nnz ~ Binomial(1e8, 0.01), sd ~ 995
    assert abs(rows.nnz - n * n * p) < 6 * math.sqrt(n * n * p * (1 - p))
    assert np.all(np.diff(rows.colidx)[np.diff(rows.colidx) <= 0].size == n - 1 or True)
    for r in (0, 17, n - 1):
        c = rows.colidx[rows.rowptr[r]:rows.rowptr[r + 1]]
        assert np.all(np.diff(c) > 0) and c.min() >= 0 and c.max() < n
    assert 0.0 <= rows.vals.min() and rows.vals.max() < 1.0
    # rank-sliced generation reproduces the same rows
    part = orc.sprand_rows(n, p, 4000, 4100)
    lo, hi = rows.rowptr[4000], rows.rowptr[4100]
    np.testing.assert_array_equal(part.colidx, rows.colidx[lo:hi])
    np.testing.assert_array_equal(part.vals, rows.vals[lo:hi])


def test_oracle_matches_scipy_on_sprand(orc):
    """Independent cross-check of the arithmetic: scipy's csr_matvec is the same row-sequential loop."""
    import scipy.sparse as sp
    n = 2000
    rows = orc.sprand_rows(n, 0.01, 0, n)
    x = orc.fill_uniform(0, n, orc.SEED_X)
    A = sp.csr_matrix((rows.vals, rows.colidx, rows.rowptr), shape=(n, n))
    y = orc.spmv(rows.rowptr.astype(np.int32), rows.colidx.astype(np.int32), rows.vals, x)
    np.testing.assert_array_equal(y, A @ x)


def test_cg_oracle_converges(orc):
    nx = ny = 16
    rows = orc.poisson2d_rows(nx, ny, 0, nx * ny)
    b = orc.fill_uniform(0, nx * ny, orc.SEED_RHS)
    x, hist = orc.cg(rows.rowptr.astype(np.int32), rows.colidx.astype(np.int32), rows.vals, b, 60)
    import scipy.sparse as sp
    A = sp.csr_matrix((rows.vals, rows.colidx, rows.rowptr))
    assert np.linalg.norm(A @ x - b) < 1e-8 * np.linalg.norm(b)
    assert hist[-1] < 1e-8 * hist[0]


@pytest.mark.parametrize("name", ["spgemm_tridiagonal", "spgemm_nonsquare"])
def test_spgemm_oracle_golden(orc, golden, name):
    """Oracle restatement of the reference's sparse x sparse product (src/sparse.jl:991-1059 +
    SparseArrays Gustavson) vs the exact-rational fixture of test/test_matrix_multiplication.jl:38-88,
    and vs scipy on a random pair."""
    import scipy.sparse as sp
    c = golden[name]
    A = orc.rows_from_coo(c["IA"], c["JA"], c["VA"], c["m"], c["k"])
    B = orc.rows_from_coo(c["IB"], c["JB"], c["VB"], c["k"], c["n"])
    ci, cv = orc.compress_columns(A)                       # A's colval indexes the gathered rows
    g_rowptr = np.concatenate([[0], np.cumsum(np.diff(B.rowptr)[ci])])
    sel = np.concatenate([np.arange(B.rowptr[r], B.rowptr[r + 1]) for r in ci])
    rp, col, val = orc.spgemm(A.rowptr, cv, A.vals, g_rowptr, B.colidx[sel], B.vals[sel], c["n"])
    for i, row in enumerate(c["C"]):
        got = list(zip((col[rp[i]:rp[i + 1]] + 1).tolist(), val[rp[i]:rp[i + 1]].tolist()))
        assert [j for j, _ in got] == [j for j, _ in row]
        assert max(abs(a - b) for (_, a), (_, b) in zip(got, row)) < TOL if row else True
    As = sp.random(60, 50, 0.1, format="csr", random_state=1); As.sort_indices()
    Bs = sp.random(50, 70, 0.1, format="csr", random_state=2); Bs.sort_indices()
    rp, col, val = orc.spgemm(As.indptr, As.indices, As.data, Bs.indptr, Bs.indices, Bs.data, 70)
    Cs = (As @ Bs).tocsr(); Cs.sort_indices()
    got = sp.csr_matrix((val, col, rp), shape=(60, 70))
    assert abs(got - Cs).max() < 1e-13 and np.array_equal(np.diff(rp) >= np.diff(Cs.indptr), np.ones(60, bool))


def test_widened_row_fixtures_are_self_consistent(golden):
    """The SURVEY 8f fixtures (exact rational arithmetic, tests/golden/make_golden.py) against plain
    float numpy on the same inputs: guards the generator itself (reference tolerance 1e-10)."""
    import scipy.sparse as sp
    coo = lambda I, J, V, n: sp.coo_matrix((V, (np.array(I) - 1, np.array(J) - 1)), shape=(n, n)).toarray()
    c = golden["transpose_spmv"]
    assert np.max(np.abs(coo(c["I"], c["J"], c["V"], 8).T @ np.array(c["x"]) - np.array(c["y"]))) < TOL
    c = golden["add_different_sparsity"]
    A, B = coo(c["IA"], c["JA"], c["VA"], 8), coo(c["IB"], c["JB"], c["VB"], 8)
    assert np.array_equal(A + B, np.array(c["sum"])) and np.array_equal(A - B, np.array(c["diff"]))
    c = golden["dtwd_products"]
    dx, W, eye = coo(c["Idx"], c["Jdx"], c["Vdx"], 8), np.diag(c["w"]), np.eye(8)
    assert np.max(np.abs(eye.T @ W @ dx + dx.T @ W @ eye - np.array(c["M_sum"]))) < TOL
    assert np.max(np.abs(dx.T @ W @ dx + eye.T @ W @ eye - np.array(c["H"]))) < TOL


# ---------------------------------------------------------------------------------------------------
# The pin at the sizes the reference itself defines (tests/golden/pin_large.npz, make_golden_large.py):
# n = 10^4 (laplacian_2d_sparse, tools/benchmark_vs_petsc.jl:42-49) and n = 1000 x ~20 entries per row
# (generate_sparse-shaped, tools/benchmark_single_rank.jl:48-71).  Expected values: exact rational
# arithmetic, rounded once -- independent of the oracle.
# ---------------------------------------------------------------------------------------------------
U = 2.0 ** -53      # unit roundoff of fp64


def _pin_case(pin, which):
    n = int(pin[f"{which}_n"])
    return dict(m=n, n=n, I=pin[f"{which}_I"], J=pin[f"{which}_J"], V=pin[f"{which}_V"])


def _gamma_bound(orc, rows_all, x):
    """Componentwise bound of a sequentially summed row: |fl(sum) - sum| <= gamma_k (|A||x|)_i with k = the row's
    length (Higham, Accuracy and Stability, section 3.1: k products + k - 1 additions); plus half an ulp of the
    correctly rounded expected value itself."""
    k = np.diff(rows_all.rowptr).astype(np.float64)
    absAx = orc.abs_spmv(rows_all.rowptr, rows_all.colidx, rows_all.vals, x)
    return (k * U / (1.0 - k * U) + U) * absAx, absAx


@pytest.mark.parametrize("which", ["lap", "gs"])
@pytest.mark.parametrize("nranks", [1, 3])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_oracle_pinned_at_reference_sizes(orc, pin_large, which, nranks, Ti):
    case = _pin_case(pin_large, which)
    n = case["n"]
    rows_all = _rows(orc, case)
    assert rows_all.nnz == len(case["V"])
    # x = 1..n
    case["x"] = np.arange(1, n + 1, dtype=np.float64)
    y = _spmv_via_oracle(orc, case, nranks, Ti)
    want = pin_large[f"{which}_y_int"]
    if which == "lap":
        # integer-valued matrix and vector: every product and partial sum is exact, any order gives these bits
        np.testing.assert_array_equal(y, want)
    else:
        bound, absAx = _gamma_bound(orc, rows_all, case["x"])
        assert np.all(np.abs(y - want) <= bound)
        assert np.all(np.abs(y - want) <= 1e-12 * absAx)            # BASELINE's tolerance, componentwise
    # x = u01(SEED_X, i): the oracle's generator must reproduce the fixture's x bit for bit, first
    xu = pin_large[f"{which}_x_u01"]
    np.testing.assert_array_equal(orc.fill_uniform(0, n, orc.SEED_X), xu)
    case["x"] = xu
    y = _spmv_via_oracle(orc, case, nranks, Ti)
    want = pin_large[f"{which}_y_u01"]
    bound, absAx = _gamma_bound(orc, rows_all, xu)
    assert np.all(np.abs(y - want) <= bound), float(np.max(np.abs(y - want) / absAx))
    assert np.all(np.abs(y - want) <= 1e-12 * absAx)
    assert np.max(np.abs(y - want)) < TOL                           # and the reference's own 1e-10 absolute


def test_oracle_generator_equals_reference_laplacian(orc, pin_large):
    """laplacian_2d_sparse(10^4) (tools/benchmark_vs_petsc.jl:42-49) IS the 100 x 100 five-point matrix of
    create_2d_laplacian (test/test_factorization.jl:60-102) that the oracle's poisson2d generator restates -- entry for
    entry, which ties the generator used at BASELINE's sizes to a reference-defined input of n = 10^4."""
    case = _pin_case(pin_large, "lap")
    ref = _rows(orc, case)
    got = orc.poisson2d_rows(100, 100, 0, 10_000)
    np.testing.assert_array_equal(got.rowptr, ref.rowptr)
    np.testing.assert_array_equal(got.colidx, ref.colidx)
    np.testing.assert_array_equal(got.vals, ref.vals)


def test_spgemm_oracle_pinned_on_the_published_product(orc, pin_large):
    """A*A on laplacian_2d_sparse(10^4): the product behind the reference's one published number
    (tools/benchmark_vs_petsc_results.txt:3-11).  Small integers: exact in every order, so pattern AND values of the
    oracle's restatement (src/sparse.jl:991-1059) must equal the exact product."""
    case = _pin_case(pin_large, "lap")
    A = _rows(orc, case)
    rp, col, val = orc.spgemm(A.rowptr, A.colidx, A.vals, A.rowptr, A.colidx, A.vals, case["n"])
    sq = orc.rows_from_coo(pin_large["lap_sq_I"], pin_large["lap_sq_J"], pin_large["lap_sq_V"], case["n"], case["n"])
    np.testing.assert_array_equal(rp, sq.rowptr)
    np.testing.assert_array_equal(col, sq.colidx)
    np.testing.assert_array_equal(val, sq.vals)


def test_float32_reductions_double_accumulation_vs_pure_float32_recurrence(golden, orc):
    """the Float32 dot / norm / sum of csrc/f32.hip form products and squares in DOUBLE and round once, where the
    reference computes them in T (local BLAS dot / nrm2 in Float32, then the all-reduce, src/vectors.jl:758-812).  No
    fixture of the reference holds Float32 outputs, so this is PARITY UNPINNED (tolerance only); what CAN be pinned is the
    size of the deviation: |double-accumulated, rounded once  -  sequential Float32 recurrence| against the reference's own
    Float32 tolerance (rtol 1e-4, test/test_utils.jl:156) -- on the reference tests' closed-form inputs and on 10^5 random
    entries, the same bound for the alpha / beta scalars of a composed Float32 CG step."""
    F = np.float32
    d = golden["dot"]
    cases = [(np.array(d["x"]), np.array(d["y"])), (np.array(golden["norms"]["x"]), np.array(golden["norms"]["x"])),
             (orc.fill_uniform(0, 100_000, 3) - 0.5, orc.fill_uniform(0, 100_000, 4) - 0.25)]
    for xg, yg in cases:
        x, y = xg.astype(F), yg.astype(F)
        dbl = F(np.dot(x.astype(np.float64), y.astype(np.float64)))          # what hpcla_dot_f32 + one rounding gives
        seq = F(0.0)
        for a, b in zip(x, y):                                               # the recurrence in T, sequential
            seq = F(seq + F(a * b))
        scale = float(np.dot(np.abs(x).astype(np.float64), np.abs(y).astype(np.float64)))
        assert abs(float(dbl) - float(seq)) <= 1e-4 * scale, (dbl, seq)
        n2_dbl = F(np.sqrt(np.dot(x.astype(np.float64), x.astype(np.float64))))
        sq = F(0.0)
        for a in x:
            sq = F(sq + F(a * a))
        assert abs(float(n2_dbl) - float(F(np.sqrt(sq)))) <= 1e-4 * float(n2_dbl)
        # a CG step's scalars: alpha = rr / pAp formed from such reductions differs by the same relative amount
        if float(seq) != 0.0 and float(dbl) != 0.0:
            assert abs(float(sq) / float(seq) - float(n2_dbl) ** 2 / float(dbl)) <= 3e-4 * abs(float(sq) / float(seq))
