"""Randomised sweep through the raw C ABI: ragged CSR shapes (empty matrices, empty rows, single rows, rows around the
464-entry wave pass and the 1984- / 2048-entry workgroup passes, duplicate-free ascending columns), both index types and bases, every
kernel family -- Float64 SpMV, Float32 SpMV, SpMM in Float64 and Float32 on row-major and column-major
blocks with ragged k, the row-major-B / column-major-C product, the run tiles on column-major blocks and the opt-in long-row entry (its short rows) -- each bit for bit against the oracle's loops (src/sparse.jl:2055-2066, 2391-2413).  Fixed seeds: a
failure names its case.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def _case(seed):
    rng = np.random.default_rng(seed)
    nrows = int(rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, int(rng.integers(1, 700))]))
    ncols = int(rng.integers(1, 3000))
    kind = seed % 4
    if kind == 0:
        lens = rng.integers(0, min(ncols, 12) + 1, nrows)
    elif kind == 1:
        lens = np.where(rng.random(nrows) < 0.5, 0, rng.integers(0, min(ncols, 40) + 1, nrows))
    elif kind == 2:
        lens = rng.integers(0, min(ncols, 9) + 1, nrows)
        for r in rng.integers(0, max(nrows, 1), min(nrows, 3)):
            lens[r] = min(ncols, int(rng.choice([463, 464, 465, 929, 1983, 1984, 1985, 2047, 2048, 2049])))
    else:
        lens = np.full(nrows, min(ncols, 7))
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    colval = (np.concatenate([np.sort(rng.choice(ncols, int(l), replace=False)) for l in lens]).astype(np.int64)
              if nrows and lens.sum() else np.empty(0, np.int64))
    vals = rng.random(len(colval)) - 0.5
    k = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 17, 20]))
    B = rng.random((ncols, k)) - 0.5
    return nrows, ncols, rowptr, colval, vals, k, B


@pytest.mark.parametrize("seed", range(48))
def test_random_shapes_every_kernel_family(hp, orc, seed):
    import torch
    nrows, ncols, rowptr, colval, vals, k, B = _case(seed)
    Ti = np.int32 if seed % 3 else np.int64
    sfx = "i32" if Ti == np.int32 else "i64"
    base = seed % 2
    s = torch.cuda.current_stream().cuda_stream
    capi = hp._capi
    rp, cv = _t((rowptr + base).astype(Ti)), _t((colval + base).astype(Ti))
    nnz = len(vals)
    ROW, COL = capi.LAYOUT_ROW, capi.LAYOUT_COL
    for T, dt, tT in ((np.float64, "f64", torch.float64), (F32, "f32", torch.float32)):
        v, Bt = vals.astype(T), B.astype(T)
        nz = _t(v)
        want = orc.spmm(rowptr.astype(Ti), colval.astype(Ti), v, Bt) if nrows else np.empty((0, k), T)
        # SpMV on column 0
        x = _t(np.ascontiguousarray(Bt[:, 0]))
        kinds = (0,)
        try:
            for kind in kinds:
                y = torch.full((max(nrows, 1),), float("nan"), dtype=tT, device="cuda")
                capi.call(f"hpcla_spmv_csr_{dt}_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), x.data_ptr(), y.data_ptr(), nrows, nnz,
                          base, s)
                np.testing.assert_array_equal(y[:nrows].cpu().numpy(), want[:, 0], err_msg=f"seed {seed} {dt} spmv kernel {kind}")
        finally:
            pass
        # SpMM, both layouts
        for lay, name in ((ROW, "row"), (COL, "col")):
            Bd = _t(Bt if lay == ROW else np.ascontiguousarray(Bt.T))
            C = torch.full((max(nrows, 1) * k,), float("nan"), dtype=tT, device="cuda")
            capi.call(f"hpcla_spmm_csr_{dt}_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bd.data_ptr(), k if lay == ROW else ncols, lay,
                      C.data_ptr(), k if lay == ROW else max(nrows, 1), lay, nrows, nnz, k, base, s)
            got = C.cpu().numpy()
            got = got[:nrows * k].reshape(nrows, k) if lay == ROW else got.reshape(k, max(nrows, 1))[:, :nrows].T
            np.testing.assert_array_equal(got, want, err_msg=f"seed {seed} {dt} spmm {name}-major k={k}")
        if dt == "f64":
            # round 5: row-major B, COLUMN-major C (the CCOL store for even k <= 16, the strided kernel otherwise)
            Bd = _t(Bt)
            ldc = max(nrows, 1) + seed % 3
            C = torch.full((k * ldc,), float("nan"), dtype=tT, device="cuda")
            capi.call(f"hpcla_spmm_csr_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bd.data_ptr(), k, ROW, C.data_ptr(), ldc, COL,
                      nrows, nnz, k, base, s)
            got = C.cpu().numpy().reshape(k, ldc)
            np.testing.assert_array_equal(got[:, :nrows].T, want, err_msg=f"seed {seed} spmm row-major B, column-major C, k={k}")
            assert np.all(np.isnan(got[:, nrows:])), f"seed {seed}: the column-major store wrote into the padding"
            # round 6: B (and a row-major C) on the padded pitch k + (k & 1): odd k on the vector kernel; NaN padding in B
            # must not reach any real column, the padding of C stays untouched
            kp = k + (k & 1) + 2 * (seed % 2)
            Bp = np.full((ncols, kp), np.nan)
            Bp[:, :k] = Bt
            Bd = _t(Bp)
            C = torch.full((max(nrows, 1), kp), float("nan"), dtype=tT, device="cuda")
            capi.call(f"hpcla_spmm_csr_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bd.data_ptr(), kp, ROW, C.data_ptr(), kp, ROW,
                      nrows, nnz, k, base, s)
            got = C.cpu().numpy()
            np.testing.assert_array_equal(got[:nrows, :k], want, err_msg=f"seed {seed} spmm padded pitch {kp}, k={k}")
            assert np.all(np.isnan(got[:nrows, k:]) | (got[:nrows, k:] == 0.0)) and np.all(np.isnan(got[:nrows, k + 1:])), \
                f"seed {seed}: the padded-pitch product wrote something else than 0.0 into the padding of C"
            C = torch.full((k * ldc,), float("nan"), dtype=tT, device="cuda")
            capi.call(f"hpcla_spmm_csr_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bd.data_ptr(), kp, ROW, C.data_ptr(), ldc, COL,
                      nrows, nnz, k, base, s)
            got = C.cpu().numpy().reshape(k, ldc)
            np.testing.assert_array_equal(got[:, :nrows].T, want, err_msg=f"seed {seed} spmm padded-pitch B, column-major C, k={k}")
            assert np.all(np.isnan(got[:, nrows:]))
            # round 5: the run tiles on COLUMN-major blocks (k = 16; most of these ragged blocks do not fit and take the per-entry
            # path, a few do): own block column-major with an even leading dimension, ghost rows row-major, odd / even n_own
            if nrows:
                import ctypes
                rng16 = np.random.default_rng(1000 + seed)
                B16 = rng16.random((ncols, 16)) - 0.5
                want16 = orc.spmm(rowptr.astype(Ti), colval.astype(Ti), v, B16)
                n_own = int(rng16.integers(1, ncols + 1))
                ldb16 = n_own + (n_own & 1) + 2 * (seed % 2)
                Bo = np.full((16, ldb16), np.nan)
                Bo[:, :n_own] = B16[:n_own].T
                Bgh = np.full((max(ncols - n_own, 1), 18), np.nan)
                Bgh[:ncols - n_own, :16] = B16[n_own:]
                dBo, dBg = _t(Bo), _t(Bgh)
                desc = torch.empty(capi.load().hpcla_spmm_runs_desc_bytes(nrows), dtype=torch.uint8, device="cuda")
                n_fit = ctypes.c_int64(-1)
                capi.call(f"hpcla_spmm_runs_build_{sfx}", rp.data_ptr(), cv.data_ptr(), nrows, nnz, base, n_own, desc.data_ptr(), ctypes.byref(n_fit), s)
                ldc16 = nrows + seed % 3
                C16 = torch.full((16, ldc16), float("nan"), dtype=tT, device="cuda")
                capi.call(f"hpcla_spmm_runs_colmajor_k16_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), dBo.data_ptr(), ldb16,
                          dBg.data_ptr() if n_own < ncols else None, 18, n_own, C16.data_ptr(), ldc16, nrows, nnz, base, desc.data_ptr(), None, 0, s)
                got = C16.cpu().numpy()
                np.testing.assert_array_equal(got[:, :nrows].T, want16, err_msg=f"seed {seed} run tiles on column-major blocks (fit {n_fit.value})")
                assert np.all(np.isnan(got[:, nrows:]))
            # round 5: the OPT-IN long-row entry -- rows of >= 928 entries in tree order (1e-12 |A||x|), every other row bit-exact
            if nrows:
                lens = np.diff(rowptr)
                long_rows = np.flatnonzero(lens >= 928).astype(np.int64)
                lr = _t(long_rows if len(long_rows) else np.zeros(1, np.int64))
                work = torch.empty(capi.load().hpcla_spmv_longrows_work_bytes(len(long_rows)) // 8, dtype=torch.float64, device="cuda")
                y = torch.full((nrows,), float("nan"), dtype=tT, device="cuda")
                capi.call(f"hpcla_spmv_longrows_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), x.data_ptr(), None, ncols, y.data_ptr(),
                          nrows, nnz, base, lr.data_ptr(), len(long_rows), 928, work.data_ptr(), s)
                got = y.cpu().numpy()
                short = np.ones(nrows, bool)
                short[long_rows] = False
                np.testing.assert_array_equal(got[short], want[short, 0], err_msg=f"seed {seed} long-row entry, short rows")
                if len(long_rows):
                    bound = orc.spmv(rowptr.astype(Ti), colval.astype(Ti), np.abs(v), np.abs(np.ascontiguousarray(Bt[:, 0])))
                    assert np.all(np.abs(got[long_rows] - want[long_rows, 0]) <= 1e-12 * bound[long_rows]), f"seed {seed} long rows"
