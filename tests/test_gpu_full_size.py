"""Full-size parity cases of BASELINE.json's configs that tests/test_gpu_parity.py covers only at reduced
size or through their pieces ("parity caveats"):

* config 3 at its own size: the whole 8192^2 Poisson problem (n = 67 108 864, nnz = 335 511 552, 5.4 GB)
  on ONE GPU, built on the device, bit-exact against the oracle;
* config 4's per-GPU share (512 x 512 x 64 planes of the 7-point Laplacian), the contract item of
  SURVEY 8d C4: EXACTLY 100 CG iterations, final ||r||_2 and x against the oracle's restatement, with the
  tolerance that is actually met written here;
* config 5's per-GPU share at its own size: 2 097 152 rows x ~29.8 stored entries with columns uniform over
  2^24, times B = 2^24 x 16 (2.1 GB, the gather set one GPU of the 8-GPU job sees): two columns of A*B bit-equal
  to the SpMV of that column (the reference's definition, src/sparse.jl:2391-2413) and 4096 sampled rows
  bit-equal to the sequential row sum;
* the packed copy with nnz % 8 != 0, arrays followed by NaN / wrong-column guard entries, so an octet
  load that strays past the end changes the result instead of faulting (the r01q fault's tail case,
  profiles/MEASUREMENTS_r04.md section C).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# CG: two CG runs that differ only in the summation order of their dot products (tree on the GPU,
# sequential in the oracle: ~1e-16 relative per dot) drift apart through the recurrence -- alpha and beta
# feed back into every later iterate.  MEASURED on this problem (the test prints it; gpurun_out/r02e_pytest_new.log):
# the residual histories agree to 2.4e-15 relative at every one of the 100 iterations and x to 2.2e-15 of
# max|x| (fused and unfused alike) -- the matrix is far from converged after 100 iterations (||r|| 2366 ->
# 502), so the recurrence has not amplified the rounding differences.  The bound asserted is BASELINE's own
# 1e-12 relative, per iteration, with a factor 400 of margin over what was observed.
CG_HIST_RTOL = 1e-12
CG_X_RTOL = 1e-12


def test_config3_poisson8192_whole_problem_one_gpu(hp, orc, gpu_backend_i32):
    import torch
    N = 8192
    n = N * N
    s0 = torch.cuda.current_stream().cuda_stream
    nnz = hp._capi.load().hpcla_poisson2d_nnz(N, N, 0, n)
    assert nnz == 5 * n - 4 * N == 335_511_552
    rp_d = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ci_d = torch.empty(nnz, dtype=torch.int64, device="cuda")
    va_d = torch.empty(nnz, dtype=torch.float64, device="cuda")
    hp._capi.call("hpcla_gen_poisson2d", N, N, 0, n, rp_d.data_ptr(), ci_d.data_ptr(), va_d.data_ptr(), s0)
    A = hp.HPCSparseMatrix_local_device(rp_d, ci_d, va_d, n, gpu_backend_i32, col_window=(0, n - 1))
    del ci_d, rp_d
    x = hp.HPCVector.zeros(A.row_partition, gpu_backend_i32)
    hp._capi.call("hpcla_fill_uniform_f64", x.v.data_ptr(), 0, n, orc.SEED_X, s0)
    y = A @ x
    torch.cuda.synchronize()
    rows = orc.poisson2d_rows(N, N, 0, n)                   # the oracle's own generator (C, host)
    assert rows.nnz == nnz
    xg = orc.fill_uniform(0, n, orc.SEED_X)
    np.testing.assert_array_equal(x.local_values(), xg)     # device fill == oracle generator
    want = orc.spmv(rows.rowptr.astype(np.int32), rows.colidx.astype(np.int32), rows.vals, xg)
    got = y.local_values()
    assert np.array_equal(got, want), f"max abs err {np.abs(got - want).max()}"
    # the device-built structure is the oracle's, too (sampled: first / last rows and a stride)
    idx = np.unique(np.concatenate([np.arange(0, 3 * N), np.arange(n - 3 * N, n), np.arange(0, n, 4099)]))
    rp_h = A.rowptr_target.cpu().numpy()
    np.testing.assert_array_equal(rp_h[idx], rows.rowptr[idx].astype(np.int32))
    del A, x, y, va_d, rows, want, got
    hp.clear_plan_cache()
    torch.cuda.empty_cache()


def test_config4_share_100_cg_iterations_vs_oracle(hp, orc, gpu_backend_i32):
    import torch
    from hpcla_amd import workloads as wl
    nx, ny, nz = 512, 512, 64
    n = nx * ny * nz
    rp, ci, va = wl.poisson3d_rows(nx, ny, nz, 0, n)
    A = hp.HPCSparseMatrix_local(rp, ci, va, n, gpu_backend_i32)
    bg = orc.fill_uniform(0, n, orc.SEED_RHS)
    b = hp.HPCVector.from_global(bg, gpu_backend_i32)
    iters = 100
    rp32, ci32 = rp.astype(np.int32), ci.astype(np.int32)
    x_ref, hist_ref = orc.cg(rp32, ci32, va, bg, iters)
    hist_ref = np.asarray(hist_ref)
    for fused in (True, False):
        x, hist = hp.cg_fixed_iterations(A, b, iters, fused=fused)
        hist = np.asarray(hist)
        assert len(hist) == iters + 1
        dev_hist = float(np.max(np.abs(hist - hist_ref) / hist_ref))
        xv = x.local_values()
        dev_x = float(np.max(np.abs(xv - x_ref)) / np.max(np.abs(x_ref)))
        print(f"CG 100 iterations (fused={fused}): final ||r|| {hist[-1]:.6e} (oracle {hist_ref[-1]:.6e}), "
              f"max rel history deviation {dev_hist:.2e}, max |x - x_ref| / max|x_ref| {dev_x:.2e}")
        assert dev_hist <= CG_HIST_RTOL, dev_hist
        assert dev_x <= CG_X_RTOL, dev_x
        assert abs(hist[-1] - hist_ref[-1]) <= CG_HIST_RTOL * hist_ref[-1]
        # the recurrence residual is the true residual
        res = b - A @ x
        assert abs(hp.norm(res) - hist[-1]) <= 1e-9 * hist[0]
    del A, b, x, res
    hp.clear_plan_cache()
    torch.cuda.empty_cache()


def test_config5_share_spmm_full_size(hp, orc, gpu_backend_i32):
    """BASELINE configs[4] (SpMM, k = 16), one GPU's share at full size, B as large as the 8-GPU job's gather
    set.  The matrix is generated on the device like the bench's (counts ~ Poisson(29.8), columns uniform)."""
    import torch
    dev = "cuda"
    rows_loc, ncols, k = 2_097_152, 1 << 24, 16
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xA11CE)
    counts = torch.poisson(torch.full((rows_loc,), 29.8, dtype=torch.float64, device=dev), generator=gen).to(torch.int64)
    rowptr = torch.zeros(rows_loc + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    nnz = int(rowptr[-1].item())
    assert 6.2e7 < nnz < 6.3e7
    cols = torch.randint(0, ncols, (nnz,), generator=gen, device=dev, dtype=torch.int64)
    rowid = torch.repeat_interleave(torch.arange(rows_loc, device=dev, dtype=torch.int64), counts)
    key = torch.sort(rowid * ncols + cols).values
    cols = key - rowid * ncols                                   # ascending within each row
    del key, rowid, counts
    vals = torch.rand(nnz, generator=gen, device=dev, dtype=torch.float64)
    # sampled rows for the sequential check: their entries, kept before the library compresses the columns
    rng = np.random.default_rng(11)
    samp = np.unique(np.concatenate([np.arange(64), np.arange(rows_loc - 64, rows_loc), rng.integers(0, rows_loc, 4096)]))
    rp_h = rowptr.cpu().numpy()
    ent = np.concatenate([np.arange(rp_h[r], rp_h[r + 1]) for r in samp])
    ent_d = torch.from_numpy(ent).to(dev)
    s_cols, s_vals = cols[ent_d].cpu().numpy(), vals[ent_d].cpu().numpy()
    A = hp.HPCSparseMatrix_local_device(rowptr, cols, vals, ncols, gpu_backend_i32, col_window=(0, ncols - 1))
    del cols
    Bl = torch.empty((ncols, k), dtype=torch.float64, device=dev)
    hp._capi.call("hpcla_fill_uniform_f64", Bl.data_ptr(), 0, ncols * k, orc.SEED_X, torch.cuda.current_stream().cuda_stream)
    B = hp.HPCMatrix_local(Bl, gpu_backend_i32)
    C = hp.spmm(A, B)
    torch.cuda.synchronize()
    assert tuple(C.A.shape) == (rows_loc, k)
    # (a) the reference's definition: column j of A*B is the SpMV of column j, bit for bit
    for j in (3, 12):
        yj = A @ B[:, j]
        assert torch.equal(C.A[:, j].contiguous(), yj.v), f"column {j} differs from A*B[:, {j}]"
    # (b) sequential row sums in stored order (multiply, then add: src/sparse.jl:2060-2064) on the sampled rows
    b_rows = Bl[torch.from_numpy(s_cols).to(dev)].cpu().numpy()             # B row of every sampled entry
    got = C.A[torch.from_numpy(samp).to(dev)].cpu().numpy()
    pos = 0
    for i, r in enumerate(samp):
        acc = np.zeros(k)
        for e in range(int(rp_h[r + 1] - rp_h[r])):
            acc = acc + s_vals[pos] * b_rows[pos]
            pos += 1
        assert np.array_equal(got[i], acc), f"row {r}"
    del A, B, C, Bl, vals
    hp.clear_plan_cache(); hp.clear_spmm_cache()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("nx,ny", [(701, 301), (1000, 53), (257, 129), (255, 131), (640, 77), (93, 1001), (129, 65)])
def test_packed_copy_tail_not_a_multiple_of_eight(hp, orc, gpu_backend_i32, nx, ny):
    """nnz % 8 covers several remainders; colval / nzval are views that END inside guarded buffers (NaN values,
    column 0), so entries read past the end would poison the last rows."""
    import torch
    b = gpu_backend_i32
    n = nx * ny
    rows = orc.poisson2d_rows(nx, ny, 0, n)
    assert rows.nnz % 8 != 0
    A = hp.HPCSparseMatrix_local(rows.rowptr, rows.colidx, rows.vals, n, b)
    xg = orc.fill_uniform(0, n, orc.SEED_X) - 0.25
    x = hp.HPCVector.from_global(xg, b)
    plan = hp.get_vector_plan(A, x)
    nnz = A.nnz
    guard = 64
    vbuf = torch.full((nnz + guard,), float("nan"), dtype=torch.float64, device="cuda")
    vbuf[:nnz] = A.nzval
    A.nzval = vbuf[:nnz]
    cbuf = torch.zeros(nnz + guard, dtype=torch.int32, device="cuda")
    cbuf[:nnz] = plan.colval_split
    plan.colval_split = cbuf[:nnz]
    want = orc.spmv(rows.rowptr.astype(np.int32), rows.colidx.astype(np.int32), rows.vals, xg)
    np.testing.assert_array_equal((A @ x).local_values(), want)          # CSR path with the guarded views
    assert A.enable_packed(x) is True
    for _ in range(2):
        np.testing.assert_array_equal((A @ x).local_values(), want)      # packed path
    A.disable_packed()
    hp.clear_plan_cache()
