"""The HIP path against the pin at the sizes the REFERENCE defines (tests/golden/pin_large.npz,
tests/golden/make_golden_large.py): laplacian_2d_sparse(10^4) -- tools/benchmark_vs_petsc.jl:42-49, the matrix
behind the reference's one published number -- and a generate_sparse(1000)-shaped matrix
(tools/benchmark_single_rank.jl:48-71).  Expected values are exact-rational products rounded once, independent
of the oracle and of the kernels.

* integer-valued case (Laplacian, x = 1..n): bit-equal, whatever the summation order;
* x = u01: within gamma_k (|A||x|)_i of the exact product (k = the row's length: the bound of ANY sequential
  sum), within BASELINE's 1e-12 (|A||x|)_i and the reference's own 1e-10 absolute (test/test_utils.jl:154-157) --
  and bit-equal to the oracle, as everywhere else;
* through the raw C ABI (`hpcla_spmv_csr_f64_*`, both index types and bases), through the host layer, and (3 real
  ranks: tests/_multirank_gpu_worker.py, cases "pin_lap" / "pin_gs") through the distributed path;
* A*A of the Laplacian (the published product): pattern and values equal the exact product.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U = 2.0 ** -53


def _raw_spmv(hp, rowptr, colval, vals, x, Ti, base=0):
    """hpcla_spmv_csr_f64_{i32,i64} directly (the entry point a Julia @ccall binds); operands stay bound to names
    until after the synchronising read-back (profiles/MEASUREMENTS_r04.md section C, r01a)."""
    import torch
    sfx = "i32" if Ti == np.int32 else "i64"
    rp = torch.from_numpy((rowptr + base).astype(Ti)).cuda()
    cv = torch.from_numpy((colval + base).astype(Ti)).cuda()
    nz, xd = torch.from_numpy(np.ascontiguousarray(vals)).cuda(), torch.from_numpy(np.ascontiguousarray(x)).cuda()
    nrows = len(rowptr) - 1
    y = torch.full((max(nrows, 1),), float("nan"), dtype=torch.float64, device="cuda")
    hp._capi.call(f"hpcla_spmv_csr_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), xd.data_ptr(), y.data_ptr(),
                  nrows, len(vals), base, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return y[:nrows].cpu().numpy()


def _case(orc, pin, which):
    n = int(pin[f"{which}_n"])
    return n, orc.rows_from_coo(pin[f"{which}_I"], pin[f"{which}_J"], pin[f"{which}_V"], n, n)


def check_against_pin(orc, rows, x, y, want, exact):
    """y (computed) against the exact-rational `want`: bit-equal when `exact`, else the sequential-sum bound."""
    if exact:
        np.testing.assert_array_equal(y, want)
        return
    k = np.diff(rows.rowptr).astype(np.float64)
    absAx = orc.abs_spmv(rows.rowptr, rows.colidx, rows.vals, x)
    err = np.abs(y - want)
    assert np.all(err <= (k * U / (1.0 - k * U) + U) * absAx), float(np.max(err / absAx))
    assert np.all(err <= 1e-12 * absAx) and np.max(err) < 1e-10


@pytest.mark.parametrize("which", ["lap", "gs"])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
@pytest.mark.parametrize("base", [0, 1])
def test_pin_raw_abi(hp, orc, pin_large, which, Ti, base):
    n, rows = _case(orc, pin_large, which)
    for xname, x in (("int", np.arange(1, n + 1, dtype=np.float64)), ("u01", pin_large[f"{which}_x_u01"])):
        y = _raw_spmv(hp, rows.rowptr, rows.colidx, rows.vals, x, Ti, base)
        check_against_pin(orc, rows, x, y, pin_large[f"{which}_y_{xname}"], exact=(which == "lap" and xname == "int"))
        np.testing.assert_array_equal(y, orc.spmv(rows.rowptr.astype(Ti), rows.colidx.astype(Ti), rows.vals, x))


@pytest.mark.parametrize("which", ["lap", "gs"])
@pytest.mark.parametrize("idx", ["i32", "i64"])
def test_pin_host_layer(hp, orc, pin_large, gpu_backend_i32, gpu_backend_i64, which, idx):
    import scipy.sparse as sp
    backend = gpu_backend_i32 if idx == "i32" else gpu_backend_i64
    n, rows = _case(orc, pin_large, which)
    A = hp.HPCSparseMatrix_from_global(sp.csr_matrix((rows.vals, rows.colidx, rows.rowptr), shape=(n, n)), backend)
    for xname, x in (("int", np.arange(1, n + 1, dtype=np.float64)), ("u01", pin_large[f"{which}_x_u01"])):
        y = (A @ hp.HPCVector.from_global(x, backend)).local_values()
        check_against_pin(orc, rows, x, y, pin_large[f"{which}_y_{xname}"], exact=(which == "lap" and xname == "int"))


def test_pin_published_product(hp, orc, pin_large, gpu_backend_i32):
    """A*A on laplacian_2d_sparse(10^4) (tools/benchmark_vs_petsc_results.txt:3-11): first product (numeric
    kernels), later ones (cached structure, then the mapped product lists) -- all equal the exact product."""
    import scipy.sparse as sp
    n, rows = _case(orc, pin_large, "lap")
    sq = orc.rows_from_coo(pin_large["lap_sq_I"], pin_large["lap_sq_J"], pin_large["lap_sq_V"], n, n)
    A = hp.HPCSparseMatrix_from_global(sp.csr_matrix((rows.vals, rows.colidx, rows.rowptr), shape=(n, n)), gpu_backend_i32)
    for _ in range(5):
        C = A @ A
        np.testing.assert_array_equal(C.rowptr.astype(np.int64), sq.rowptr)
        np.testing.assert_array_equal(C.col_indices[C.colval.astype(np.int64)], sq.colidx)
        np.testing.assert_array_equal(C.nzval.cpu().numpy(), sq.vals)
