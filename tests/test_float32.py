"""Float32 element type on the hot path (csrc/f32.hip; include/hpcla_rocm.h "Float32 element type").

The reference is generic in T and its GPU test configurations are CUDA x {Float32, Float64} and Metal x Float32
(test/test_utils.jl:62-80) with tolerance 1e-4 for Float32 (:156).  Bar here, as for Float64: the row sums run in stored
order in T with separately rounded multiply and add, so SpMV / SpMM results are BIT-IDENTICAL to the oracle's Float32 loop
(the restatement of src/sparse.jl:2055-2066 with T = Float32); reductions are formed in double and must agree with a numpy
double sum to 1e-12 relative (far inside the reference's 1e-4).

The oracle's Float32 loop is pinned like the Float64 one: on the reference tests' closed-form inputs and on
laplacian_2d_sparse(10^4) with x = 1..n every value is a small integer or a half, exact in Float32, so the loop must
reproduce the exact-rational fixtures to the bit.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the oracle's Float32 loop against the fixtures
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["spmv_tridiagonal", "spmv_nonsquare", "spmv_local_ctor", "laplacian2d_4x3",
                                  "laplacian2d_3x5"])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_oracle_f32_reproduces_the_golden_products(orc, golden, name, Ti):
    case = golden[name]
    rows = orc.rows_from_coo(case["I"], case["J"], case["V"], case["m"], case["n"])
    ci, cv = orc.compress_columns(rows)
    want = np.array(case["y"])
    assert np.array_equal(want.astype(F32).astype(np.float64), want), "fixture not exact in Float32"
    y = orc.spmv(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals.astype(F32), np.array(case["x"])[ci].astype(F32))
    assert y.dtype == F32
    np.testing.assert_array_equal(y.astype(np.float64), want)


def test_oracle_f32_pinned_on_the_reference_laplacian(orc, pin_large):
    """laplacian_2d_sparse(10^4) (tools/benchmark_vs_petsc.jl:42-49), x = 1..n: integers below 2^24, exact in Float32."""
    n = int(pin_large["lap_n"])
    rows = orc.rows_from_coo(pin_large["lap_I"], pin_large["lap_J"], pin_large["lap_V"], n, n)
    ci, cv = orc.compress_columns(rows)
    x = np.arange(1, n + 1, dtype=F32)
    y = orc.spmv(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals.astype(F32), x[ci])
    np.testing.assert_array_equal(y.astype(np.float64), pin_large["lap_y_int"])
    # u01 values are not exact in Float32: the loop stays within the Float32 rounding bound of the exact product
    xu = pin_large["lap_x_u01"].astype(F32)
    yu = orc.spmv(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals.astype(F32), xu[ci])
    scale = orc.abs_spmv(rows.rowptr, cv, rows.vals, np.abs(pin_large["lap_x_u01"])[ci])
    assert np.all(np.abs(yu.astype(np.float64) - pin_large["lap_y_u01"]) <= 8 * 2.0 ** -24 * scale)


def test_oracle_f32_spmm_is_the_column_loop(orc):
    rows = orc.sprand_rows(700, 0.02, 0, 700)
    ci, cv = orc.compress_columns(rows)
    rng = np.random.default_rng(3)
    B = rng.random((len(ci), 5)).astype(F32)
    C = orc.spmm(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals.astype(F32), B)
    for c in range(5):
        np.testing.assert_array_equal(C[:, c], orc.spmv(rows.rowptr.astype(np.int32), cv.astype(np.int32),
                                                        rows.vals.astype(F32), np.ascontiguousarray(B[:, c])))


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the C ABI
# ---------------------------------------------------------------------------------------------------------------------
def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def _raw_spmv_f32(hp, rowptr, colval, vals, x, Ti, base=0, misalign=0):
    import torch
    sfx = "i32" if Ti == np.int32 else "i64"
    rp = _t((rowptr + base).astype(Ti))
    cv_h, nz_h = (colval + base).astype(Ti), vals.astype(F32)
    # misalign > 0: colval / nzval start `misalign` elements into their allocations (no 16-byte alignment: the kernel's
    # entry-by-entry path)
    cv = _t(np.concatenate([np.zeros(misalign, Ti), cv_h]))[misalign:]
    nz = _t(np.concatenate([np.zeros(misalign, F32), nz_h]))[misalign:]
    xd = _t(x.astype(F32))
    nrows = len(rowptr) - 1
    y = torch.full((max(nrows, 1),), float("nan"), dtype=torch.float32, device="cuda")
    hp._capi.call(f"hpcla_spmv_csr_f32_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), xd.data_ptr(), y.data_ptr(),
                  nrows, len(vals), base, _stream())
    torch.cuda.synchronize()
    return y[:nrows].cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["spmv_tridiagonal", "spmv_nonsquare", "spmv_local_ctor", "laplacian2d_4x3"])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
@pytest.mark.parametrize("base", [0, 1])
def test_spmv_f32_golden_raw_abi(hp, orc, golden, name, Ti, base):
    case = golden[name]
    rows = orc.rows_from_coo(case["I"], case["J"], case["V"], case["m"], case["n"])
    ci, cv = orc.compress_columns(rows)
    y = _raw_spmv_f32(hp, rows.rowptr, cv, rows.vals, np.array(case["x"])[ci], Ti, base)
    np.testing.assert_array_equal(y.astype(np.float64), np.array(case["y"]))


@pytest.mark.gpu
@pytest.mark.parametrize("n,p", [(1, 1.0), (255, 0.05), (256, 0.05), (257, 0.05), (1000, 0.01), (10_000, 0.01),
                                 (5000, 0.0002)])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
@pytest.mark.parametrize("misalign", [0, 1])
def test_spmv_f32_sprand_bit_exact(hp, orc, n, p, Ti, misalign):
    rows = orc.sprand_rows(n, p, 0, n)
    ci, cv = orc.compress_columns(rows)
    x = orc.fill_uniform(0, n, orc.SEED_X).astype(F32)
    want = orc.spmv(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals.astype(F32), x[ci])
    got = _raw_spmv_f32(hp, rows.rowptr, cv, rows.vals, x[ci], Ti, misalign=misalign)
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
def test_spmv_f32_long_rows_and_pass_boundaries(hp, orc):
    """Rows longer than one 464-entry wave pass (running sum carried through the pass loop), next to empty rows and to
    waves whose entry count sits at 463 / 464 / 465."""
    rng = np.random.default_rng(5)
    ncols = 20_000
    lens = np.zeros(600, dtype=np.int64)
    lens[[0, 7, 130, 131, 299]] = [9000, 464, 465, 1, 15000]
    lens[200:260] = 37
    lens[320:384] = 7                      # a whole wave of 7-entry rows: 448 entries
    lens[384:447] = 7
    lens[447] = 22                         # wave of 463
    lens[448:511] = 7
    lens[511] = 23                         # wave of 464
    lens[512:575] = 7
    lens[575] = 24                         # wave of 465
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    colval = np.concatenate([np.sort(rng.choice(ncols, int(l), replace=False)) for l in lens]).astype(np.int64)
    vals = (rng.random(len(colval)) - 0.5).astype(F32)
    x = (rng.random(ncols) - 0.5).astype(F32)
    for Ti in (np.int32, np.int64):
        want = orc.spmv(rowptr.astype(Ti), colval.astype(Ti), vals, x)
        for base in (0, 1):
            got = _raw_spmv_f32(hp, rowptr, colval, vals, x, Ti, base)
            np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(300, 41), (129, 257), (64, 64, 9)])
def test_spmv_f32_stencils_bit_exact(hp, orc, shape):
    if len(shape) == 2:
        rows = orc.poisson2d_rows(shape[0], shape[1], 0, shape[0] * shape[1])
    else:
        rows = orc.poisson3d_rows(*shape, 0, shape[0] * shape[1] * shape[2])
    n = rows.nrows
    ci, cv = orc.compress_columns(rows)
    x = orc.fill_uniform(0, n, orc.SEED_X).astype(F32)
    want = orc.spmv(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals.astype(F32), x[ci])
    got = _raw_spmv_f32(hp, rows.rowptr, cv, rows.vals, x[ci], np.int32)
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_spmv_split_f32_ghosts_are_doubles_and_block_lists_apply(hp, orc, Ti):
    """Split column space: columns < n_own index x_own (float), the others the WIDENED ghost segment (double); a block list
    restricts the launch to those row blocks and leaves the other rows untouched."""
    import torch
    sfx = "i32" if Ti == np.int32 else "i64"
    n, n_own = 3000, 2200
    rows = orc.sprand_rows(n, 0.004, 0, n)
    x = orc.fill_uniform(0, n, orc.SEED_X).astype(F32)
    want = orc.spmv(rows.rowptr.astype(Ti), rows.colidx.astype(Ti), rows.vals.astype(F32), x)
    rp, cv, nz = _t(rows.rowptr.astype(Ti)), _t(rows.colidx.astype(Ti)), _t(rows.vals.astype(F32))
    x_own, ghost = _t(x[:n_own]), _t(x[n_own:].astype(np.float64))
    rpb = hp._capi.load().hpcla_spmv_rows_per_block()
    nblk = (n + rpb - 1) // rpb
    y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    hp._capi.call(f"hpcla_spmv_split_f32_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), x_own.data_ptr(),
                  ghost.data_ptr(), n_own, y.data_ptr(), n, len(rows.vals), 0, 0, 0, _stream())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y.cpu().numpy(), want)
    some = np.array([b for b in range(nblk) if b % 3 != 1], dtype=np.int32)
    lst = _t(some)
    y2 = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    hp._capi.call(f"hpcla_spmv_split_f32_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), x_own.data_ptr(),
                  ghost.data_ptr(), n_own, y2.data_ptr(), n, len(rows.vals), 0, lst.data_ptr(), len(some), _stream())
    torch.cuda.synchronize()
    got = y2.cpu().numpy()
    for b in range(nblk):
        sl = slice(b * rpb, min(n, (b + 1) * rpb))
        if b % 3 != 1:
            np.testing.assert_array_equal(got[sl], want[sl])
        else:
            assert np.all(np.isnan(got[sl]))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3, 8, 16, 17])
@pytest.mark.parametrize("layout", ["row", "col"])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_spmm_f32_bit_exact_raw_abi(hp, orc, k, layout, Ti):
    """A * HPCMatrix (src/sparse.jl:2391-2413) in Float32: every column is one Float32 SpMV; both layouts."""
    import torch
    sfx = "i32" if Ti == np.int32 else "i64"
    n = 1500
    rows = orc.sprand_rows(n, 0.01, 0, n)
    ci, cv = orc.compress_columns(rows)
    rng = np.random.default_rng(k)
    B = rng.random((len(ci), k)).astype(F32)
    want = orc.spmm(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals.astype(F32), B)
    rp, cvd, nz = _t(rows.rowptr.astype(Ti)), _t(cv.astype(Ti)), _t(rows.vals.astype(F32))
    lay = hp._capi.LAYOUT_ROW if layout == "row" else hp._capi.LAYOUT_COL
    Bd = _t(B if layout == "row" else np.ascontiguousarray(B.T))
    ldb = k if layout == "row" else len(ci)
    ldc = k if layout == "row" else n
    C = torch.full((n * k,), float("nan"), dtype=torch.float32, device="cuda")
    hp._capi.call(f"hpcla_spmm_csr_f32_{sfx}", rp.data_ptr(), cvd.data_ptr(), nz.data_ptr(), Bd.data_ptr(), ldb, lay,
                  C.data_ptr(), ldc, lay, n, len(rows.vals), k, 0, _stream())
    torch.cuda.synchronize()
    got = C.cpu().numpy().reshape((n, k) if layout == "row" else (k, n))
    np.testing.assert_array_equal(got if layout == "row" else got.T, want)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["row", "col"])
@pytest.mark.parametrize("k", [2, 16, 40, 70])
def test_spmm_f32_long_and_empty_rows(hp, orc, layout, k):
    """Rows spanning many wave passes (the row-major kernel carries their running sums in C), empty rows at the start, in the
    middle and at the end of a wave, more than 64 columns (two column groups of the row-major kernel)."""
    import torch
    rng = np.random.default_rng(k)
    ncols = 12_000
    lens = np.zeros(333, dtype=np.int64)
    lens[[1, 7, 64, 130, 131, 255, 256, 300]] = [5000, 464, 1, 465, 930, 3, 2000, 11]
    lens[200:250] = 9
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    colval = np.concatenate([np.sort(rng.choice(ncols, int(l), replace=False)) for l in lens]).astype(np.int32)
    vals = (rng.random(len(colval)) - 0.5).astype(F32)
    B = (rng.random((ncols, k)) - 0.5).astype(F32)
    n = len(lens)
    want = orc.spmm(rowptr.astype(np.int32), colval, vals, B)
    lay = hp._capi.LAYOUT_ROW if layout == "row" else hp._capi.LAYOUT_COL
    Bd = _t(B if layout == "row" else np.ascontiguousarray(B.T))
    C = torch.full((n * k,), float("nan"), dtype=torch.float32, device="cuda")
    rp, cv, nz = _t(rowptr.astype(np.int32)), _t(colval), _t(vals)
    hp._capi.call("hpcla_spmm_csr_f32_i32", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(),
                  Bd.data_ptr(), k if layout == "row" else ncols, lay, C.data_ptr(), k if layout == "row" else n, lay, n,
                  len(vals), k, 0, _stream())
    torch.cuda.synchronize()
    got = C.cpu().numpy().reshape((n, k) if layout == "row" else (k, n))
    np.testing.assert_array_equal(got if layout == "row" else got.T, want)


@pytest.mark.gpu
@pytest.mark.parametrize("k,ldb,ldg,ldc", [(16, 16, 16, 16), (16, 20, 18, 24), (16, 17, 16, 16), (12, 12, 12, 12), (6, 6, 7, 6),
                                          (1, 1, 1, 1), (1, 1, 1, 3), (1, 2, 1, 1), (1, 1, 2, 1)])
def test_spmm_split_f32_with_widened_ghost_rows(hp, orc, k, ldb, ldg, ldc):
    """Row-major operands with their own leading dimensions: multiples of 4 (2 for the double ghosts) take the kernel's
    four-columns-per-lane form, anything else the one-column-per-lane form; padding columns stay untouched."""
    import torch
    n, n_own = 2000, 1300
    rows = orc.sprand_rows(n, 0.005, 0, n)
    rng = np.random.default_rng(11)
    B = rng.random((n, k)).astype(F32)
    want = orc.spmm(rows.rowptr.astype(np.int32), rows.colidx.astype(np.int32), rows.vals.astype(F32), B)
    rp, cv, nz = _t(rows.rowptr.astype(np.int32)), _t(rows.colidx.astype(np.int32)), _t(rows.vals.astype(F32))
    Bo = np.full((n_own, ldb), np.nan, F32)
    Bo[:, :k] = B[:n_own]
    Bg = np.full((n - n_own, ldg), np.nan, np.float64)
    Bg[:, :k] = B[n_own:]
    B_own, B_ghost = _t(Bo), _t(Bg)
    C = torch.full((n, ldc), float("nan"), dtype=torch.float32, device="cuda")
    hp._capi.call("hpcla_spmm_split_f32_i32", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), B_own.data_ptr(), ldb,
                  B_ghost.data_ptr(), ldg, n_own, C.data_ptr(), ldc, n, len(rows.vals), k, 0, 0, 0, _stream())
    torch.cuda.synchronize()
    got = C.cpu().numpy()
    np.testing.assert_array_equal(got[:, :k], want)
    assert np.all(np.isnan(got[:, k:]))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 3, 4, 1023, 1024, 1025, 1_000_003])
def test_reductions_and_updates_f32(hp, n):
    import torch
    rng = np.random.default_rng(n)
    x = (rng.random(n) - 0.5).astype(F32)
    y = (rng.random(n) - 0.5).astype(F32)
    xd, yd = _t(np.concatenate([x, np.zeros(4, F32)]))[:n], _t(np.concatenate([y, np.zeros(4, F32)]))[:n]
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    work = torch.empty(hp._capi.load().hpcla_reduce_work_bytes() // 8, dtype=torch.float64, device="cuda")
    x64, y64 = x.astype(np.float64), y.astype(np.float64)

    def red(fn, *args):
        hp._capi.call(fn, None, *args, out.data_ptr(), work.data_ptr(), _stream())
        return float(out.item())

    def close(a, b):
        assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), (a, b)

    close(red("hpcla_dot_f32", xd.data_ptr(), yd.data_ptr(), n), float(np.dot(x64, y64)))
    close(red("hpcla_nrm2sq_f32", xd.data_ptr(), n), float(np.dot(x64, x64)))
    close(red("hpcla_asum_f32", xd.data_ptr(), n), float(np.abs(x64).sum()))
    close(red("hpcla_sum_f32", xd.data_ptr(), n), float(x64.sum()))
    assert red("hpcla_amax_f32", xd.data_ptr(), n) == (float(np.abs(x64).max()) if n else 0.0)
    assert red("hpcla_maxval_f32", xd.data_ptr(), n, 0) == (float(x64.max()) if n else -np.inf)
    assert red("hpcla_maxval_f32", xd.data_ptr(), n, 1) == (float((-x64).max()) if n else -np.inf)
    if n:
        # one NaN element: the max reductions must return NaN like Julia's maximum / norm(., Inf) (`v > s ? v : s`
        # never selected it, so rows poisoned by an expired halo wait passed as a finite maximum)
        xn = x.copy()
        xn[n // 2] = np.nan
        xnd = _t(np.concatenate([xn, np.zeros(4, F32)]))[:n]
        assert np.isnan(red("hpcla_amax_f32", xnd.data_ptr(), n))
        assert np.isnan(red("hpcla_maxval_f32", xnd.data_ptr(), n, 0)) and np.isnan(red("hpcla_maxval_f32", xnd.data_ptr(), n, 1))
    # updates: separately rounded multiply and add in float -- numpy's float32 arithmetic, bit for bit
    z = torch.full((max(n, 1),), float("nan"), dtype=torch.float32, device="cuda")
    a, b = F32(1.7), F32(-0.3)
    hp._capi.call("hpcla_axpby_f32", float(a), xd.data_ptr(), float(b), yd.data_ptr(), z.data_ptr(), n, _stream())
    np.testing.assert_array_equal(z[:n].cpu().numpy(), a * x + b * y)
    hp._capi.call("hpcla_scale_f32", float(a), xd.data_ptr(), z.data_ptr(), n, _stream())
    np.testing.assert_array_equal(z[:n].cpu().numpy(), a * x)
    hp._capi.call("hpcla_divide_f32", xd.data_ptr(), float(a), z.data_ptr(), n, _stream())
    np.testing.assert_array_equal(z[:n].cpu().numpy(), x / a)


@pytest.mark.gpu
def test_f32_entries_validate_their_arguments(hp):
    import torch
    d = torch.zeros(16, dtype=torch.float32, device="cuda")
    i = torch.zeros(16, dtype=torch.int32, device="cuda")
    s = _stream()
    with pytest.raises(hp._capi.HPCLAError, match="index_base"):
        hp._capi.call("hpcla_spmv_csr_f32_i32", i.data_ptr(), i.data_ptr(), d.data_ptr(), d.data_ptr(), d.data_ptr(), 4, 4, 2, s)
    with pytest.raises(hp._capi.HPCLAError, match="negative"):
        hp._capi.call("hpcla_spmv_csr_f32_i32", i.data_ptr(), i.data_ptr(), d.data_ptr(), d.data_ptr(), d.data_ptr(), -1, 4, 0, s)
    with pytest.raises(hp._capi.HPCLAError, match="null"):
        hp._capi.call("hpcla_spmv_csr_f32_i32", None, i.data_ptr(), d.data_ptr(), d.data_ptr(), d.data_ptr(), 4, 4, 0, s)
    with pytest.raises(hp._capi.HPCLAError, match="layout"):
        hp._capi.call("hpcla_spmm_csr_f32_i32", i.data_ptr(), i.data_ptr(), d.data_ptr(), d.data_ptr(), 4, 7, d.data_ptr(), 4, 0,
                      4, 4, 4, 0, s)
    with pytest.raises(hp._capi.HPCLAError, match="aligned"):
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        hp._capi.call("hpcla_nrm2sq_f32", None, d.data_ptr() + 4, 3, out.data_ptr(), out.data_ptr(), s)
    # empty problems are no-ops
    hp._capi.call("hpcla_spmv_csr_f32_i32", None, None, None, None, None, 0, 0, 0, s)
    hp._capi.call("hpcla_axpby_f32", 1.0, None, 1.0, None, None, 0, s)


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the host layer (reads like the reference's tests run over its CUDA x Float32 configuration)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def f32_backends(hp):
    return {np.int32: hp.backend_rocm_serial(F32, np.int32), np.int64: hp.backend_rocm_serial(F32, np.int64)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["spmv_tridiagonal", "spmv_nonsquare", "spmv_local_ctor"])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_spmv_f32_golden_host_layer(hp, golden, f32_backends, name, Ti):
    """test/test_vector_multiplication.jl:42-118 for T = Float32: HPCSparseMatrix(A, backend) * HPCVector, mul!, exact types."""
    import scipy.sparse as sp
    import torch
    backend = f32_backends[Ti]
    case = golden[name]
    A = sp.coo_matrix((case["V"], (np.array(case["I"]) - 1, np.array(case["J"]) - 1)), shape=(case["m"], case["n"])).tocsr()
    Adist = hp.HPCSparseMatrix_from_global(A, backend)
    xdist = hp.HPCVector.from_global(np.array(case["x"]), backend)
    assert Adist.nzval.dtype == torch.float32 and xdist.v.dtype == torch.float32
    ydist = Adist @ xdist
    assert isinstance(ydist, hp.HPCVector) and ydist.backend is backend and ydist.v.dtype == torch.float32
    np.testing.assert_array_equal(ydist.partition, Adist.row_partition)
    np.testing.assert_array_equal(ydist.local_values().astype(np.float64), np.array(case["y"]))   # exact in Float32
    y2 = hp.HPCVector.zeros(Adist.row_partition, backend)
    hp.mul_(y2, Adist, xdist)
    np.testing.assert_array_equal(y2.local_values(), ydist.local_values())
    # to_backend moves, it does not convert: CPU copy keeps Float32; a Float64 backend is refused
    y_cpu = hp.to_backend(ydist, hp.cpu_version(backend))
    assert y_cpu.v.dtype == torch.float32 and y_cpu.v.device.type == "cpu"
    with pytest.raises(TypeError, match="element type"):
        hp.to_backend(ydist, hp.backend_rocm_serial(np.float64, Ti))


@pytest.mark.gpu
def test_f32_host_layer_products_reductions_and_refusals(hp, orc, f32_backends):
    import torch
    backend = f32_backends[np.int64]             # the reference's default Ti: narrowed plan, Int32 kernels
    nx, ny = 300, 41
    n = nx * ny
    rows = orc.poisson2d_rows(nx, ny, 0, n)
    A = hp.HPCSparseMatrix_local(rows.rowptr, rows.colidx, rows.vals, n, backend)
    xg = orc.fill_uniform(0, n, orc.SEED_X).astype(F32)
    x = hp.HPCVector.from_global(xg, backend)
    ci, cv = orc.compress_columns(rows)
    want = orc.spmv(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals.astype(F32), xg[ci])
    y = A @ x
    np.testing.assert_array_equal(y.local_values(), want)
    plan = hp.get_vector_plan(A, x)
    assert plan.is_f32 and plan.narrowed and plan.colval_split.dtype == torch.int32
    # SpMM through the host layer: the row-major kernel, k = 16 and a ragged k
    for k in (16, 5):
        Bg = (orc.fill_uniform(0, n * k, 99).reshape(n, k) - 0.5).astype(F32)
        C = A @ hp.HPCMatrix.from_global(Bg, backend)
        assert C.A.dtype == torch.float32
        np.testing.assert_array_equal(C.A.cpu().numpy(), orc.spmm(rows.rowptr.astype(np.int32), cv.astype(np.int32),
                                                                  rows.vals.astype(F32), np.ascontiguousarray(Bg[ci])))
    # dot / norm return T (rounded once from the double the kernels form)
    d = hp.dot(x, y)
    assert d == float(F32(d)) and abs(d - float(xg.astype(np.float64) @ want.astype(np.float64))) <= 1e-6 * abs(d)
    assert hp.norm(x, 1) == float(F32(np.abs(xg.astype(np.float64)).sum()))
    # a*A shares the structure and scales in float
    np.testing.assert_array_equal((2.5 * A).nzval.cpu().numpy(), F32(2.5) * rows.vals.astype(F32))
    # the widened rows and the CG pieces stay Float64 entries
    for call in (lambda: hp.mul_dot_(y, A, x, None), lambda: hp.transpose(A) @ x, lambda: A @ A, lambda: A + A,
                 lambda: hp.norm(x, 3), lambda: hp.prod(x), lambda: x.axpy_(1.0, x)):
        with pytest.raises(TypeError, match="Float64"):
            call()
    with pytest.raises(TypeError, match="float64 or float32"):
        hp.backend_rocm_serial(np.float16, np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 3])
def test_float32_backend_across_ranks(nranks):
    """Real processes, real exchanges (peer-window push on a shared GPU): tests/_multirank_f32_worker.py.  2 ranks: Int32
    indices; 3 ranks: Int64 (the reference's default Ti, narrowed plans)."""
    from hpcla_amd.launch import spawn_ranks
    os.environ.pop("HPCLA_HALO_MODE", None)
    rc = spawn_ranks([os.path.join(ROOT, "tests", "_multirank_f32_worker.py")], nranks,
                     env_extra={"HPCLA_PUSH_TIMEOUT_S": "30", "HPCLA_MR_TYPES": "i32" if nranks == 2 else "i64"},
                     timeout=600, forward_rank0_stdout=False)
    assert rc == 0


@pytest.mark.gpu
def test_f32_golden_spmm_dot_norm_at_the_reference_tolerance(hp, golden, f32_backends):
    """test/test_new_operations.jl:43-82, :139-147 and test/test_vector_multiplication.jl:163-195 as the reference runs them
    over its CUDA x Float32 configuration: tolerance(Float32) = 1e-4 (test/test_utils.jl:156); exact where Float32 is."""
    import scipy.sparse as sp
    backend = f32_backends[np.int32]
    TOL = 1e-4
    case = golden["spmm_sym"]
    A = sp.coo_matrix((case["V"], (np.array(case["I"]) - 1, np.array(case["J"]) - 1)), shape=(case["m"], case["n"])).tocsr()
    C = hp.HPCSparseMatrix_from_global(A, backend) @ hp.HPCMatrix.from_global(np.array(case["B"]), backend)
    got = C.A.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - np.array(case["C"]))) < TOL * max(1.0, np.max(np.abs(case["C"])))
    assert abs(C.norm() - case["C_fro"]) < TOL * case["C_fro"]
    d = golden["dot"]
    x, y = hp.HPCVector.from_global(np.array(d["x"]), backend), hp.HPCVector.from_global(np.array(d["y"]), backend)
    assert abs(hp.dot(x, y) - d["dot_xy"]) < TOL * abs(d["dot_xy"]) and abs(hp.dot(x, x) - d["dot_xx"]) < TOL * d["dot_xx"]
    nr = golden["norms"]
    v = hp.HPCVector.from_global(np.array(nr["x"]), backend)
    assert abs(hp.norm(v) - nr["norm2"]) < TOL * nr["norm2"]
    assert hp.norm(v, 1) == nr["norm1"] and hp.norm(v, np.inf) == nr["norminf"]      # 55 and 10: exact in Float32


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(1, 1), (33, 16), (1000, 3), (70, 100)])
def test_transpose_f32_both_directions(hp, rows, cols):
    """hpcla_transpose_f32: column-major (Julia Matrix) <-> row-major rows, with padded leading dimensions."""
    import torch
    rng = np.random.default_rng(rows * 1000 + cols)
    M = rng.random((rows, cols)).astype(F32)
    ld_c, ld_r = rows + 3, cols + 2
    colmaj = np.full((cols, ld_c), np.nan, F32)
    colmaj[:, :rows] = M.T                                  # element (i, c) at c * ld_c + i
    src = _t(colmaj)
    dst = torch.full((rows, ld_r), float("nan"), dtype=torch.float32, device="cuda")
    hp._capi.call("hpcla_transpose_f32", src.data_ptr(), ld_c, hp._capi.LAYOUT_COL, dst.data_ptr(), ld_r, hp._capi.LAYOUT_ROW,
                  rows, cols, _stream())
    got = dst.cpu().numpy()
    np.testing.assert_array_equal(got[:, :cols], M)
    assert np.all(np.isnan(got[:, cols:]))
    back = torch.full((cols, ld_c), float("nan"), dtype=torch.float32, device="cuda")
    hp._capi.call("hpcla_transpose_f32", dst.data_ptr(), ld_r, hp._capi.LAYOUT_ROW, back.data_ptr(), ld_c, hp._capi.LAYOUT_COL,
                  rows, cols, _stream())
    np.testing.assert_array_equal(back.cpu().numpy()[:, :rows], M.T)
    with pytest.raises(hp._capi.HPCLAError, match="layout"):
        hp._capi.call("hpcla_transpose_f32", src.data_ptr(), ld_c, 5, dst.data_ptr(), ld_r, 0, rows, cols, _stream())


@pytest.mark.gpu
@pytest.mark.parametrize("b_lay,c_lay", [("row", "col"), ("col", "row")])
@pytest.mark.parametrize("k", [1, 5, 16])
def test_spmm_f32_mixed_layouts(hp, orc, b_lay, c_lay, k):
    """hpcla_spmm_csr_f32_* takes B and C in different layouts (the strided lanes = rows kernel)."""
    import torch
    n = 900
    rows = orc.sprand_rows(n, 0.02, 0, n)
    ci, cv = orc.compress_columns(rows)
    rng = np.random.default_rng(7 * k)
    B = rng.random((len(ci), k)).astype(F32)
    want = orc.spmm(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals.astype(F32), B)
    rp, cvd, nz = _t(rows.rowptr.astype(np.int32)), _t(cv.astype(np.int32)), _t(rows.vals.astype(F32))
    lay = {"row": hp._capi.LAYOUT_ROW, "col": hp._capi.LAYOUT_COL}
    Bd = _t(B if b_lay == "row" else np.ascontiguousarray(B.T))
    C = torch.full((n * k,), float("nan"), dtype=torch.float32, device="cuda")
    hp._capi.call("hpcla_spmm_csr_f32_i32", rp.data_ptr(), cvd.data_ptr(), nz.data_ptr(), Bd.data_ptr(),
                  k if b_lay == "row" else len(ci), lay[b_lay], C.data_ptr(), k if c_lay == "row" else n, lay[c_lay],
                  n, len(rows.vals), k, 0, _stream())
    torch.cuda.synchronize()
    got = C.cpu().numpy().reshape((n, k) if c_lay == "row" else (k, n))
    np.testing.assert_array_equal(got if c_lay == "row" else got.T, want)


@pytest.mark.gpu
def test_cg_on_a_float32_backend_is_the_composed_iteration(hp, orc, f32_backends):
    """CG on a Float32 backend: composed from the Float32 operators (mul!, dot, u + a*v) like a caller of the reference
    composes it; against the same recurrence in numpy Float32 (its dot sums in Float32, the kernels' in double: compared at
    the reference's Float32 tolerance, test/test_utils.jl:156), and it converges on the SPD 5-point matrix."""
    backend = f32_backends[np.int32]
    nx, ny = 48, 40
    n = nx * ny
    rows = orc.poisson2d_rows(nx, ny, 0, n)
    A = hp.HPCSparseMatrix_local(rows.rowptr, rows.colidx, rows.vals, n, backend)
    bg = orc.fill_uniform(0, n, orc.SEED_RHS).astype(F32)
    b = hp.HPCVector.from_global(bg, backend)
    iters = 30
    x, hist = hp.cg_fixed_iterations(A, b, iters)
    assert x.v.dtype.is_floating_point and x.local_values().dtype == F32 and len(hist) == iters + 1
    import scipy.sparse as sp
    M = sp.csr_matrix((rows.vals.astype(F32), rows.colidx, rows.rowptr), shape=(n, n))
    xr, r, p = np.zeros(n, F32), bg.copy(), bg.copy()
    rr = F32(r @ r)
    ref = [float(np.sqrt(rr))]
    for _ in range(iters):
        Ap = M @ p
        alpha = rr / F32(p @ Ap)
        xr = xr + alpha * p
        r = r - alpha * Ap
        rr_new = F32(r @ r)
        p = r + (rr_new / rr) * p
        rr = rr_new
        ref.append(float(np.sqrt(rr)))
    assert np.allclose(hist, ref, rtol=2e-3, atol=1e-4 * ref[0]), (hist[-3:], ref[-3:])
    assert hist[-1] < 0.5 * hist[0]
    assert np.max(np.abs(x.local_values() - xr)) <= 1e-3 * np.max(np.abs(xr))
