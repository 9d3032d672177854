"""More than 2^31 stored entries: the size at which a plan can NOT be narrowed to Int32 (sparse.can_narrow_indices) and the
Int64 kernels are the only ones that may run -- every entry offset, every pass start, every quad address is 64-bit arithmetic
or wrong.  No BASELINE configuration is this large per GPU (config 4's share has 1.2e8 entries), but the reference's default
index type is Int (src/backends.jl:348, 369) precisely so that such matrices exist, and 288 GB of HBM holds them.

A banded matrix with exactly 8 entries per row, n = 2^28 + 3 rows, nnz = 2^31 + 24, generated on the device; values and x are
small integers, so every row sum is exact in Float64 AND Float32 whatever the order: the expected y comes from the closed form
in int64 arithmetic (chunked torch ops on the device -- no oracle can hold 34 GB in seconds) and must match bit for bit.
Covered: the row-gather and the quad SpMV kernel, the Float32 SpMV, the row-major SpMM (k = 2) -- all through the Int64
entry points of the C ABI.  ~60 GB of device memory; skipped when the card does not have it free.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 2 ** 28 + 3
W = 8
NNZ = N * W
CHUNK = 2 ** 24


def _v(i, j):          # entry (row i, j-th stored entry), column i + j
    return ((i + 3 * j) % 7) - 3


def _x(c):
    return (c % 5) - 2


def test_spmv_and_spmm_with_more_than_2_31_entries(hp):
    import torch
    assert NNZ > 2 ** 31
    free, _total = torch.cuda.mem_get_info()
    if free < 80 * 2 ** 30:
        pytest.skip(f"needs ~60 GB of free device memory, {free / 2**30:.0f} GiB available")
    dev = "cuda"
    s = torch.cuda.current_stream().cuda_stream
    rowptr = torch.arange(0, N + 1, dtype=torch.int64, device=dev) * W
    colval = torch.empty(NNZ, dtype=torch.int64, device=dev)
    nz64 = torch.empty(NNZ, dtype=torch.float64, device=dev)
    j = torch.arange(W, dtype=torch.int64, device=dev)[None, :]
    for r0 in range(0, N, CHUNK):
        r1 = min(N, r0 + CHUNK)
        i = torch.arange(r0, r1, dtype=torch.int64, device=dev)[:, None]
        colval[r0 * W:r1 * W] = (i + j).reshape(-1)
        nz64[r0 * W:r1 * W] = _v(i, j).reshape(-1).to(torch.float64)
    ncols = N + W
    c = torch.arange(ncols, dtype=torch.int64, device=dev)
    xi = _x(c)
    x64 = xi.to(torch.float64)
    del c

    def expected(r0, r1, col_scale=None):
        i = torch.arange(r0, r1, dtype=torch.int64, device=dev)[:, None]
        xv = xi[(i + j).reshape(-1)].reshape(-1, W)
        if col_scale is not None:
            xv = xv * col_scale
        return (_v(i, j) * xv).sum(dim=1)

    def check(y, what, col_scale=None):
        for r0 in range(0, N, CHUNK):
            r1 = min(N, r0 + CHUNK)
            want = expected(r0, r1, col_scale).to(y.dtype)
            if not torch.equal(y[r0:r1], want):
                bad = torch.nonzero(y[r0:r1] != want).flatten()
                raise AssertionError(f"{what}: {bad.numel()} rows differ in [{r0}, {r1}); first row {r0 + int(bad[0])}: "
                                     f"got {float(y[r0 + int(bad[0])])}, want {float(want[int(bad[0])])}")

    y = torch.full((N,), float("nan"), dtype=torch.float64, device=dev)
    try:
        for kind, name in ((0, "row-gather"),):
            y.fill_(float("nan"))
            hp._capi.call("hpcla_spmv_csr_f64_i64", rowptr.data_ptr(), colval.data_ptr(), nz64.data_ptr(), x64.data_ptr(), y.data_ptr(),
                          N, NNZ, 0, s)
            check(y, f"Float64 SpMV, {name} kernel")
    finally:
        pass
    # 1-based arrays (Julia's): the same product
    rowptr += 1
    colval += 1
    y.fill_(float("nan"))
    hp._capi.call("hpcla_spmv_csr_f64_i64", rowptr.data_ptr(), colval.data_ptr(), nz64.data_ptr(), x64.data_ptr(), y.data_ptr(), N, NNZ, 1, s)
    check(y, "Float64 SpMV, index_base = 1")
    rowptr -= 1
    colval -= 1
    del y
    # row-major SpMM, k = 2: column 1 of B is 2 * x
    B = torch.stack([x64, 2.0 * x64], dim=1).contiguous()
    C = torch.full((N, 2), float("nan"), dtype=torch.float64, device=dev)
    hp._capi.call("hpcla_spmm_csr_f64_i64", rowptr.data_ptr(), colval.data_ptr(), nz64.data_ptr(), B.data_ptr(), 2, hp._capi.LAYOUT_ROW,
                  C.data_ptr(), 2, hp._capi.LAYOUT_ROW, N, NNZ, 2, 0, s)
    check(C[:, 0].contiguous(), "SpMM column 0")
    check(C[:, 1].contiguous(), "SpMM column 1", col_scale=2)
    # ... and on column-major blocks (csrc/colmajor.hip)
    Bc = B.t().contiguous()
    Cc = torch.full((2, N), float("nan"), dtype=torch.float64, device=dev)
    hp._capi.call("hpcla_spmm_csr_f64_i64", rowptr.data_ptr(), colval.data_ptr(), nz64.data_ptr(), Bc.data_ptr(), ncols, hp._capi.LAYOUT_COL,
                  Cc.data_ptr(), N, hp._capi.LAYOUT_COL, N, NNZ, 2, 0, s)
    check(Cc[0], "column-major SpMM column 0")
    check(Cc[1], "column-major SpMM column 1", col_scale=2)
    del B, C, Bc, Cc
    # Float32 values (csrc/f32.hip), same structure
    nz32 = nz64.to(torch.float32)
    del nz64
    x32 = x64.to(torch.float32)
    y32 = torch.full((N,), float("nan"), dtype=torch.float32, device=dev)
    hp._capi.call("hpcla_spmv_csr_f32_i64", rowptr.data_ptr(), colval.data_ptr(), nz32.data_ptr(), x32.data_ptr(), y32.data_ptr(), N, NNZ, 0, s)
    check(y32, "Float32 SpMV")
    torch.cuda.synchronize()
    del nz32, x32, y32, colval, rowptr
    torch.cuda.empty_cache()


def test_more_than_2_31_rows(hp):
    """n = 2^31 + 5 rows with one stored entry each: row-block indices times 256, row offsets and vector lengths beyond Int32
    in the SpMV, the reductions and the updates.  Small integers again: exact, order-independent, compared bit for bit with
    chunked int64 arithmetic on the device.  ~90 GB."""
    import torch
    n = 2 ** 31 + 5
    free, _total = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip(f"needs ~90 GB of free device memory, {free / 2**30:.0f} GiB available")
    dev = "cuda"
    s = torch.cuda.current_stream().cuda_stream
    ncols = 1_000_003
    rowptr = torch.arange(0, n + 1, dtype=torch.int64, device=dev)
    colval = torch.empty(n, dtype=torch.int64, device=dev)
    nz = torch.empty(n, dtype=torch.float64, device=dev)
    big = 2 ** 26
    for r0 in range(0, n, big):
        r1 = min(n, r0 + big)
        i = torch.arange(r0, r1, dtype=torch.int64, device=dev)
        colval[r0:r1] = i % ncols
        nz[r0:r1] = ((i % 7) - 3).to(torch.float64)
    xi = (torch.arange(ncols, dtype=torch.int64, device=dev) % 5) - 2
    x = xi.to(torch.float64)
    y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    hp._capi.call("hpcla_spmv_csr_f64_i64", rowptr.data_ptr(), colval.data_ptr(), nz.data_ptr(), x.data_ptr(), y.data_ptr(), n, n, 0, s)
    del rowptr
    tot_dot = tot_sq = 0
    for r0 in range(0, n, big):
        r1 = min(n, r0 + big)
        i = torch.arange(r0, r1, dtype=torch.int64, device=dev)
        want = ((i % 7) - 3) * xi[i % ncols]
        assert torch.equal(y[r0:r1], want.to(torch.float64)), f"rows [{r0}, {r1}) differ"
        tot_dot += int((want * ((i % 7) - 3)).sum())
        tot_sq += int((want * want).sum())
    # reductions and an update over n > 2^31 elements: dot(y, nz), sum y^2, z = 2y - nz
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    work = torch.empty(hp._capi.load().hpcla_reduce_work_bytes() // 8, dtype=torch.float64, device=dev)
    hp._capi.call("hpcla_dot_f64", None, y.data_ptr(), nz.data_ptr(), n, out.data_ptr(), work.data_ptr(), s)
    assert float(out.item()) == float(tot_dot)
    hp._capi.call("hpcla_nrm2sq_f64", None, y.data_ptr(), n, out.data_ptr(), work.data_ptr(), s)
    assert float(out.item()) == float(tot_sq)
    del colval
    z = torch.empty(n, dtype=torch.float64, device=dev)
    hp._capi.call("hpcla_axpby_f64", 2.0, y.data_ptr(), -1.0, nz.data_ptr(), z.data_ptr(), n, s)
    for r0 in range(0, n, big):
        r1 = min(n, r0 + big)
        assert torch.equal(z[r0:r1], 2.0 * y[r0:r1] - nz[r0:r1])
    del x, y, z, nz
    torch.cuda.empty_cache()
