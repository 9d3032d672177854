"""A * B on the caller's COLUMN-major blocks (csrc/colmajor.hip, rowgather_t.h): Julia's Matrix is column-major
(src/dense.jl:63), and with lanes = rows the product needs no layout conversion around it.  Bar: the same bits as the
reference's column loop (src/sparse.jl:2391-2413) = the oracle = the row-major kernels.

hpcla_spmm_csr_f64_* routes here by itself when both layouts are HPCLA_LAYOUT_COL (tests/test_gpu_parity.py's layout cases run
through it); this file adds what those do not reach: long and empty rows, padded leading dimensions, the split form with the
halo plan's row-major ghost segment and block lists, both element types, and -- with real ranks -- the exchange posted from a
column-major block (hpcla_halo_begin_strided_*).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("k", [2, 8, 16, 17, 40])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_spmm_colmajor_f64_long_and_empty_rows(hp, orc, k, Ti):
    import torch
    sfx = "i32" if Ti == np.int32 else "i64"
    rng = np.random.default_rng(k)
    ncols = 12_000
    lens = np.zeros(333, dtype=np.int64)
    lens[[1, 7, 64, 130, 131, 255, 256, 300]] = [5000, 464, 1, 465, 930, 3, 2000, 11]
    lens[200:250] = 9
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    colval = np.concatenate([np.sort(rng.choice(ncols, int(l), replace=False)) for l in lens]).astype(Ti)
    vals = rng.random(len(colval)) - 0.5
    B = rng.random((ncols, k)) - 0.5
    n = len(lens)
    want = orc.spmm(rowptr.astype(Ti), colval, vals, B)
    ldb, ldc = ncols + 5, n + 3                              # padded leading dimensions
    Bc = np.full((k, ldb), np.nan)
    Bc[:, :ncols] = B.T
    rp, cv, nz, Bd = _t(rowptr.astype(Ti)), _t(colval), _t(vals), _t(Bc)
    C = torch.full((k, ldc), float("nan"), dtype=torch.float64, device="cuda")
    for base in (0, 1):
        rpb, cvb = _t(rowptr.astype(Ti) + base), _t(colval + base)
        C.fill_(float("nan"))
        hp._capi.call(f"hpcla_spmm_csr_f64_{sfx}", rpb.data_ptr(), cvb.data_ptr(), nz.data_ptr(), Bd.data_ptr(), ldb,
                      hp._capi.LAYOUT_COL, C.data_ptr(), ldc, hp._capi.LAYOUT_COL, n, len(vals), k, base, _stream())
        got = C.cpu().numpy()
        np.testing.assert_array_equal(got[:, :n].T, want)
        assert np.all(np.isnan(got[:, n:]))
    del rp, cv


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("k", [1, 3, 16])
def test_spmm_split_colmajor_ghost_segment_and_block_lists(hp, orc, dt, k):
    """Own block and result column-major, ghost rows row-major doubles (the halo plan's segment); a block list restricts the
    launch to those 256-row blocks and leaves the other rows untouched."""
    import torch
    T = np.float64 if dt == "f64" else np.float32
    tT = torch.float64 if dt == "f64" else torch.float32
    n, n_own = 3000, 2100
    rows = orc.sprand_rows(n, 0.004, 0, n)
    rng = np.random.default_rng(5 + k)
    B = (rng.random((n, k)) - 0.5).astype(T)
    vals = rows.vals.astype(T)
    want = orc.spmm(rows.rowptr.astype(np.int32), rows.colidx.astype(np.int32), vals, B)
    rp, cv, nz = _t(rows.rowptr.astype(np.int32)), _t(rows.colidx.astype(np.int32)), _t(vals)
    ldb, ldg, ldc = n_own + 4, k + 2, n + 8
    Bo = np.full((k, ldb), np.nan, T)
    Bo[:, :n_own] = B[:n_own].T
    Bg = np.full((n - n_own, ldg), np.nan)
    Bg[:, :k] = B[n_own:]
    B_own, B_ghost = _t(Bo), _t(Bg)
    rpb = hp._capi.load().hpcla_spmv_rows_per_block()
    nblk = (n + rpb - 1) // rpb
    fn = f"hpcla_spmm_split_colmajor_{dt}_i32"
    C = torch.full((k, ldc), float("nan"), dtype=tT, device="cuda")
    hp._capi.call(fn, rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), B_own.data_ptr(), ldb, B_ghost.data_ptr(), ldg, n_own,
                  C.data_ptr(), ldc, n, len(vals), k, 0, 0, 0, _stream())
    got = C.cpu().numpy()
    np.testing.assert_array_equal(got[:, :n].T, want)
    assert np.all(np.isnan(got[:, n:]))
    some = np.array([b for b in range(nblk) if b % 3 != 1], dtype=np.int32)
    lst = _t(some)
    C.fill_(float("nan"))
    hp._capi.call(fn, rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), B_own.data_ptr(), ldb, B_ghost.data_ptr(), ldg, n_own,
                  C.data_ptr(), ldc, n, len(vals), k, 0, lst.data_ptr(), len(some), _stream())
    got = C.cpu().numpy()[:, :n].T
    for b in range(nblk):
        sl = slice(b * rpb, min(n, (b + 1) * rpb))
        if b % 3 != 1:
            np.testing.assert_array_equal(got[sl], want[sl])
        else:
            assert np.all(np.isnan(got[sl]))
    with pytest.raises(hp._capi.HPCLAError, match="leading dimension"):
        hp._capi.call(fn, rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), B_own.data_ptr(), n_own - 1, None, k, n_own,
                      C.data_ptr(), ldc, n, len(vals), 2, 0, 0, 0, _stream())


def test_colmajor_stencil_equals_rowmajor_product(hp, orc):
    """The 5-point matrix x 16: the column-major product equals the row-major (run-tile / gather) product bit for bit."""
    import torch
    nx, ny, k = 256, 130, 16
    n = nx * ny
    rows = orc.poisson2d_rows(nx, ny, 0, n)
    ci, cv = orc.compress_columns(rows)
    B = orc.fill_uniform(0, n * k, 77).reshape(n, k)
    rp, cvd, nz = _t(rows.rowptr.astype(np.int32)), _t(cv.astype(np.int32)), _t(rows.vals)
    Br, Bc = _t(B), _t(B.T)
    Cr = torch.empty((n, k), dtype=torch.float64, device="cuda")
    Cc = torch.empty((k, n), dtype=torch.float64, device="cuda")
    ROW, COL = hp._capi.LAYOUT_ROW, hp._capi.LAYOUT_COL
    hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cvd.data_ptr(), nz.data_ptr(), Br.data_ptr(), k, ROW, Cr.data_ptr(), k, ROW,
                  n, len(rows.vals), k, 0, _stream())
    hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cvd.data_ptr(), nz.data_ptr(), Bc.data_ptr(), n, COL, Cc.data_ptr(), n, COL,
                  n, len(rows.vals), k, 0, _stream())
    assert torch.equal(Cr, Cc.t())
    np.testing.assert_array_equal(Cr.cpu().numpy(), orc.spmm(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals, B[ci]))


@pytest.mark.parametrize("nranks", [2, 3])
def test_colmajor_product_across_ranks(nranks):
    """Real processes (peer-window push on a shared GPU): the exchange posted from a column-major block
    (hpcla_halo_begin_strided_f64 / _f32), interior blocks overlapping it, boundary blocks behind it; tests/_multirank_colmajor_worker.py."""
    from hpcla_amd.launch import spawn_ranks
    os.environ.pop("HPCLA_HALO_MODE", None)
    rc = spawn_ranks([os.path.join(ROOT, "tests", "_multirank_colmajor_worker.py")], nranks,
                     env_extra={"HPCLA_PUSH_TIMEOUT_S": "30"}, timeout=600, forward_rank0_stdout=False)
    assert rc == 0


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_special_values_propagate_like_the_reference_loop(hp, orc, dt):
    """Inf, NaN, signed zeros and denormals in x / B and in the stored values: the kernels perform the reference loop's
    operations one by one (acc = 0; acc += a * b, separately rounded), so Inf - Inf, 0 * Inf and the sign of an exact zero come
    out as on the CPU -- SpMV, row-major and column-major SpMM.  (NaN payloads are not compared: IEEE leaves them open.)"""
    import torch
    T = np.float64 if dt == "f64" else np.float32
    tT = torch.float64 if dt == "f64" else torch.float32
    n, k = 2000, 4
    rows = orc.sprand_rows(n, 0.01, 0, n)
    ci, cv = orc.compress_columns(rows)
    rng = np.random.default_rng(12)
    vals = (rows.vals - 0.5).astype(T)
    vals[rng.integers(0, len(vals), 40)] = 0.0
    vals[rng.integers(0, len(vals), 40)] = -0.0
    vals[rng.integers(0, len(vals), 10)] = np.inf
    vals[rng.integers(0, len(vals), 5)] = np.finfo(T).tiny / 8          # denormal
    for r, z in ((10, 0.0), (11, -0.0), (700, -0.0)):                     # whole rows of signed zeros: exact zero sums
        vals[rows.rowptr[r]:rows.rowptr[r + 1]] = z
    B = (rng.random((len(ci), k)) - 0.5).astype(T)
    for col, special in enumerate((np.inf, -np.inf, np.nan, -0.0)):
        B[rng.integers(0, len(ci), 25), col] = special
    B[rng.integers(0, len(ci), 25), 0] = np.finfo(T).max                # overflow to Inf in a product or a sum
    rp, cvd, nz = _t(rows.rowptr.astype(np.int32)), _t(cv.astype(np.int32)), _t(vals)
    with np.errstate(all="ignore"):
        want = orc.spmm(rows.rowptr.astype(np.int32), cv.astype(np.int32), vals, B)

    def same(got, what):
        np.testing.assert_array_equal(got, want, err_msg=what)             # NaN == NaN, values and infinities exact
        assert np.array_equal(np.signbit(got[want == 0]), np.signbit(want[want == 0])), f"{what}: sign of zero"
    assert np.isnan(want).any() and np.isinf(want).any() and (want == 0).any()
    # SpMV per column
    for c in range(k):
        y = torch.full((n,), 7.0, dtype=tT, device="cuda")
        hp._capi.call(f"hpcla_spmv_csr_{dt}_i32", rp.data_ptr(), cvd.data_ptr(), nz.data_ptr(), _t(np.ascontiguousarray(B[:, c])).data_ptr(),
                      y.data_ptr(), n, len(vals), 0, _stream())
        np.testing.assert_array_equal(y.cpu().numpy(), want[:, c], err_msg=f"SpMV column {c}")
    ROW, COL = hp._capi.LAYOUT_ROW, hp._capi.LAYOUT_COL
    Br, Bc = _t(B), _t(np.ascontiguousarray(B.T))
    Cr = torch.full((n, k), 7.0, dtype=tT, device="cuda")
    hp._capi.call(f"hpcla_spmm_csr_{dt}_i32", rp.data_ptr(), cvd.data_ptr(), nz.data_ptr(), Br.data_ptr(), k, ROW, Cr.data_ptr(), k, ROW,
                  n, len(vals), k, 0, _stream())
    same(Cr.cpu().numpy(), "row-major SpMM")
    Cc = torch.full((k, n), 7.0, dtype=tT, device="cuda")
    hp._capi.call(f"hpcla_spmm_csr_{dt}_i32", rp.data_ptr(), cvd.data_ptr(), nz.data_ptr(), Bc.data_ptr(), len(ci), COL, Cc.data_ptr(), n, COL,
                  n, len(vals), k, 0, _stream())
    same(Cc.cpu().numpy().T, "column-major SpMM")


def test_banded_block_count(hp, orc):
    """hpcla_spmm_banded_blocks_*: the structure test a column-major caller applies (stencils: every block; random: none)."""
    import ctypes

    def count(rows, max_runs, Ti=np.int32):
        ci, cv = orc.compress_columns(rows)
        rp, cvd = _t(rows.rowptr.astype(Ti)), _t(cv.astype(Ti))
        nb = ctypes.c_int64(-1)
        hp._capi.call("hpcla_spmm_banded_blocks_" + ("i32" if Ti == np.int32 else "i64"), rp.data_ptr(), cvd.data_ptr(), rows.nrows,
                      rows.nnz, 0, len(ci), max_runs, ctypes.byref(nb), _stream())
        return int(nb.value), (rows.nrows + 63) // 64
    p2 = orc.poisson2d_rows(256, 70, 0, 256 * 70)
    p3 = orc.poisson3d_rows(200, 20, 6, 0, 200 * 20 * 6)        # lines longer than a block: i, i +- nx, i +- nx*ny are 5 separate runs
    sr = orc.sprand_rows(20000, 0.001, 0, 20000)
    for Ti in (np.int32, np.int64):
        assert count(p2, 16, Ti) == ((256 * 70 + 63) // 64,) * 2
        got, blocks = count(p3, 16, Ti)
        assert got == blocks
    got3, blocks3 = count(p3, 4)
    assert got3 < blocks3                              # 5 runs per interior block: the run-tile limit does not fit, the banded one does
    got, blocks = count(sr, 16)
    assert got < 0.05 * blocks
    with pytest.raises(hp._capi.HPCLAError, match="max_runs"):
        count(p2, 0)


@pytest.mark.parametrize("k", [1, 2, 5, 6, 12, 15, 16, 17, 40])
@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_spmm_f64_rowmajor_B_colmajor_C(hp, orc, k, Ti):
    """Round 5: row-major B rows, COLUMN-major C -- the unstructured product of a column-major caller without the conversion
    of C (csrc/spmm.hip CCOL: the results leave through LDS as runs of 64 doubles per column, one launch per 16-column tile;
    odd k on B's own pitch k -- as here -- takes the strided kernel, on a padded pitch the same store:
    tests/test_gpu_parity.py::test_spmm_bit_exact_padded_pitch).
    hpcla_spmm_csr_f64_* (ROW, COL) and hpcla_spmm_split_ccol_f64_* with a ghost segment and block lists; long rows (several
    LDS passes), empty rows, a last block of fewer than 64 rows, even and odd leading dimensions of C; untouched padding.
    Bar: the oracle's bits (= the reference's column loop, src/sparse.jl:2391-2413)."""
    import torch
    sfx = "i32" if Ti == np.int32 else "i64"
    rng = np.random.default_rng(100 + k)
    ncols = 9_000
    lens = rng.integers(0, 12, 333)
    lens[[1, 7, 64, 130, 131, 255, 256, 300]] = [3000, 513, 1, 0, 1537, 3, 2000, 11]
    lens[200:230] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    colval = np.concatenate([np.sort(rng.choice(ncols, int(l), replace=False)) for l in lens]).astype(Ti)
    vals = rng.random(len(colval)) - 0.5
    B = rng.random((ncols, k)) - 0.5
    n = len(lens)
    want = orc.spmm(rowptr.astype(Ti), colval, vals, B)
    rp, cv, nz, Bd = _t(rowptr.astype(Ti)), _t(colval), _t(vals), _t(B)
    ROW, COL = hp._capi.LAYOUT_ROW, hp._capi.LAYOUT_COL
    for ldc in (n, n + 3, n + 4):
        C = torch.full((k, ldc), float("nan"), dtype=torch.float64, device="cuda")
        hp._capi.call(f"hpcla_spmm_csr_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bd.data_ptr(), k, ROW,
                      C.data_ptr(), ldc, COL, n, len(vals), k, 0, _stream())
        got = C.cpu().numpy()
        np.testing.assert_array_equal(got[:, :n].T, want)
        assert np.all(np.isnan(got[:, n:]))
    # split form: columns >= n_own live in a ghost segment with its own leading dimension; interior / boundary block lists
    n_own = 6_000
    ldg = k + (k % 2) + 2
    Bo = _t(B[:n_own])
    Bg = np.full((ncols - n_own, ldg), np.nan)
    Bg[:, :k] = B[n_own:]
    Bgd = _t(Bg)
    rpb = hp._capi.load().hpcla_spmm_rows_per_block()
    nblk = (n + rpb - 1) // rpb
    touches = np.array([np.any(colval[rowptr[b * rpb]:rowptr[min((b + 1) * rpb, n)]] >= n_own) for b in range(nblk)])
    interior, boundary = _t(np.flatnonzero(~touches).astype(np.int32)), _t(np.flatnonzero(touches).astype(np.int32))
    ldc = n + 2
    C = torch.full((k, ldc), float("nan"), dtype=torch.float64, device="cuda")
    for blocks in (interior, boundary):
        if blocks.numel():
            hp._capi.call(f"hpcla_spmm_split_ccol_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bo.data_ptr(), k,
                          Bgd.data_ptr(), ldg, n_own, C.data_ptr(), ldc, n, len(vals), k, 0, blocks.data_ptr(), blocks.numel(),
                          _stream())
    got = C.cpu().numpy()
    np.testing.assert_array_equal(got[:, :n].T, want)
    assert np.all(np.isnan(got[:, n:]))
    with pytest.raises(hp._capi.HPCLAError):
        hp._capi.call(f"hpcla_spmm_split_ccol_f64_{sfx}", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bo.data_ptr(), k,
                      Bgd.data_ptr(), ldg, n_own, C.data_ptr(), n - 1, n, len(vals), k, 0, None, 0, _stream())


def test_spmm_ccol_full_size_unstructured_same_bits_as_rowmajor(hp):
    """The CCOL store at config 5's per-GPU shape in small (2^18 rows x 2^21 columns, ~30 entries per row, k = 16): the
    column-major result equals the row-major kernel's, element for element (the row-major kernel is held to the oracle by
    tests/test_gpu_parity.py; this run is about the store path under a full grid, block order hint included)."""
    import torch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    n, ncols, k = 1 << 18, 1 << 21, 16
    counts = torch.poisson(torch.full((n,), 29.8, dtype=torch.float64, device="cuda"), generator=gen).to(torch.int64)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(counts, 0, out=rowptr[1:])
    nnz = int(rowptr[-1].item())
    rowid = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int64), counts)
    key = torch.sort(rowid * ncols + torch.randint(0, ncols, (nnz,), generator=gen, device="cuda", dtype=torch.int64)).values
    cols = (key - rowid * ncols).to(torch.int32)
    rp = rowptr.to(torch.int32)
    vals = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float64) - 0.5
    B = torch.rand((ncols, k), generator=gen, device="cuda", dtype=torch.float64) - 0.5
    Cr = torch.empty((n, k), dtype=torch.float64, device="cuda")
    Cc = torch.full((k, n), float("nan"), dtype=torch.float64, device="cuda")
    ROW, COL = hp._capi.LAYOUT_ROW, hp._capi.LAYOUT_COL
    for group in (0, 64):
        hp._capi.call("hpcla_spmm_block_order_hint", rp.data_ptr(), group)
        hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cols.data_ptr(), vals.data_ptr(), B.data_ptr(), k, ROW,
                      Cr.data_ptr(), k, ROW, n, nnz, k, 0, _stream())
        Cc.fill_(float("nan"))
        hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cols.data_ptr(), vals.data_ptr(), B.data_ptr(), k, ROW,
                      Cc.data_ptr(), n, COL, n, nnz, k, 0, _stream())
        assert torch.equal(Cc.t(), Cr)
    hp._capi.call("hpcla_spmm_block_order_hint", rp.data_ptr(), 0)


@pytest.mark.parametrize("Ti", [np.int32, np.int64])
def test_spmm_run_tiles_on_colmajor_blocks(hp, orc, Ti):
    """hpcla_spmm_runs_colmajor_k16_f64_* (round 5): the run tiles for a column-major caller -- own runs widened to even
    rows and staged by the 16-byte LDS-DMA, lanes = rows, ghost entries read from the row-major ghost segment.  Bar: the
    oracle's bits (= the reference's column loop, src/sparse.jl:2391-2413).  Cases: every column owned with even and odd row
    counts (the odd one sends its last block down the per-entry path: a widened run would read past the column), padded
    leading dimensions filled with NaN (a widened run READS one row of padding or a neighbour's row; nothing of it may reach
    a sum), a rank's slab with the own / ghost cut inside a run (even and odd n_own), both index bases, permuted block lists,
    structures whose blocks do not fit (7-point, unstructured, > 512 entries), and the alignment refusals."""
    import ctypes
    import torch
    k = 16
    s = _stream()
    sfx = "i32" if Ti == np.int32 else "i64"
    lib = hp._capi.load()

    def run_case(rows, n_own, base, lists=True, pad_b=2):
        n = rows.nrows
        ci, cv = orc.compress_columns(rows)
        ncomp = len(ci)
        Bg = orc.fill_uniform(0, ncomp * k, 21).reshape(ncomp, k)
        want = orc.spmm(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals, Bg)
        rp, dcv, nz = _t((rows.rowptr + base).astype(Ti)), _t((cv + base).astype(Ti)), _t(rows.vals)
        ldb = n_own + (n_own & 1) + pad_b                    # even, >= n_own
        ldg, ldc = k + 2, n + 3
        Bo = np.full((k, max(ldb, 2)), np.nan)
        Bo[:, :n_own] = Bg[:n_own].T
        dB_own = _t(Bo)
        dB_gh = None
        if n_own < ncomp:
            G = np.full((ncomp - n_own, ldg), np.nan)
            G[:, :k] = Bg[n_own:]
            dB_gh = _t(G)
        desc = torch.empty(lib.hpcla_spmm_runs_desc_bytes(n), dtype=torch.uint8, device="cuda")
        n_fit = ctypes.c_int64(-1)
        hp._capi.call(f"hpcla_spmm_runs_build_{sfx}", rp.data_ptr(), dcv.data_ptr(), n, rows.nnz, base, n_own, desc.data_ptr(),
                      ctypes.byref(n_fit), s)
        nblk = (n + 63) // 64
        perm = np.random.default_rng(4).permutation(nblk).astype(np.int32)
        for lst in ([None, _t(perm)] if lists else [None]):
            C = torch.full((k, ldc), float("nan"), dtype=torch.float64, device="cuda")
            hp._capi.call(f"hpcla_spmm_runs_colmajor_k16_f64_{sfx}", rp.data_ptr(), dcv.data_ptr(), nz.data_ptr(), dB_own.data_ptr(), Bo.shape[1],
                          dB_gh.data_ptr() if dB_gh is not None else None, ldg, n_own, C.data_ptr(), ldc, n, rows.nnz, base,
                          desc.data_ptr(), lst.data_ptr() if lst is not None else None, nblk if lst is not None else 0, s)
            got = C.cpu().numpy()
            np.testing.assert_array_equal(got[:, :n].T, want)
            assert np.all(np.isnan(got[:, n:]))
        return int(n_fit.value), nblk, (rp, dcv, nz, dB_own, desc)

    # 5-point matrix, every column owned: even row count, then odd (200 x 37 = 7400; 201 x 37 = 7437), no padding at all
    for nx, ny in ((200, 37), (201, 37)):
        rows = orc.poisson2d_rows(nx, ny, 0, nx * ny)
        fit, nblk, _ = run_case(rows, nx * ny, 0, pad_b=0)
        assert fit == nblk
        run_case(rows, nx * ny, 1, lists=False)
    # a rank's slab: ghost lines above and below, the own / ghost cut INSIDE a run of consecutive columns; even and odd n_own
    nx, ny = 200, 37
    for lo, hi in ((5 * nx + 13, 21 * nx + 150), (5 * nx + 13, 21 * nx + 151)):
        slab = orc.poisson2d_rows(nx, ny, lo, hi)
        ci, _ = orc.compress_columns(slab)
        for n_own in (int(np.searchsorted(ci, hi)), int(np.searchsorted(ci, hi)) - 1):
            run_case(slab, n_own, 0)
    # four runs of 50 rows with ODD first rows: the descriptor fits (200 tile rows) but the widened runs need 208 -> that block takes
    # the per-entry path; with 49-row runs (widened: 50 each = 200) it is staged.
    for run_len in (50, 49):
        rowsx = 70
        colsx = np.concatenate([[1 + 100 * r + (i % run_len) for r in range(4)] for i in range(rowsx)] + [np.arange(400)]).astype(np.int64)
        rpx = np.concatenate([np.arange(rowsx + 1) * 4, 4 * rowsx + 1 + np.arange(400)]).astype(np.int64)   # + 400 one-entry rows: every column is touched
        fit, nblk, _ = run_case(orc.LocalRows(rpx, colsx, np.random.default_rng(run_len).standard_normal(len(colsx)), 400), 400, 0, lists=False)
        assert fit == nblk == (rowsx + 400 + 63) // 64
    # structures whose blocks do not fit: every block takes the per-entry path, still the right product
    r3 = orc.poisson3d_rows(24, 24, 6, 0, 24 * 24 * 6)
    fit, nblk, _ = run_case(r3, 24 * 24 * 6, 0, lists=False)
    assert fit < nblk
    run_case(orc.sprand_rows(3000, 0.004, 0, 700), 2000, 0, lists=False)
    rng = np.random.default_rng(3)
    lens = np.full(130, 3, dtype=np.int64)
    lens[70] = 600
    rpx = np.concatenate([[0], np.cumsum(lens)])
    colsx = np.concatenate([np.sort(rng.choice(5000, size=l, replace=False)) for l in lens])
    _, _, (rp, dcv, nz, dB, desc) = run_case(orc.LocalRows(rpx, colsx.astype(np.int64), rng.standard_normal(int(rpx[-1])), 5000), 700, 0,
                                             lists=False)
    # refusals: an odd leading dimension, a B_own off the 16-byte grid, a leading dimension below the row count
    C = torch.empty((k, 130), dtype=torch.float64, device="cuda")
    fn = f"hpcla_spmm_runs_colmajor_k16_f64_{sfx}"
    with pytest.raises(hp._capi.HPCLAError, match="even leading dimension"):
        hp._capi.call(fn, rp.data_ptr(), dcv.data_ptr(), nz.data_ptr(), dB.data_ptr(), 701, None, 0, 700, C.data_ptr(), 130, 130, int(rpx[-1]), 0,
                      desc.data_ptr(), None, 0, s)
    with pytest.raises(hp._capi.HPCLAError, match="16-byte aligned"):
        hp._capi.call(fn, rp.data_ptr(), dcv.data_ptr(), nz.data_ptr(), dB.data_ptr() + 8, 702, None, 0, 700, C.data_ptr(), 130, 130, int(rpx[-1]), 0,
                      desc.data_ptr(), None, 0, s)
    with pytest.raises(hp._capi.HPCLAError, match="leading dimension smaller"):
        hp._capi.call(fn, rp.data_ptr(), dcv.data_ptr(), nz.data_ptr(), dB.data_ptr(), 702, None, 0, 700, C.data_ptr(), 129, 130, int(rpx[-1]), 0,
                      desc.data_ptr(), None, 0, s)
    assert lib.hpcla_spmm_runs_colmajor_k16_f64_i32(None, None, None, None, 0, None, 0, 0, None, 5, 5, 5, 0, None, None, 0, None) != 0


def test_spmm_run_tiles_colmajor_full_size_same_bits_as_direct(hp):
    """Config 4 in small (5-point matrix 2048 x 1024 rows x 16): the column-major run tiles under a full
    grid against the lanes = rows kernel (held to the oracle above and in tests/test_gpu_parity.py), element for element,
    with the block-order hint off and on."""
    import ctypes
    import torch
    nx, ny, k = 2048, 1024, 16
    n = nx * ny
    s = _stream()
    lib = hp._capi.load()
    nnz = lib.hpcla_poisson2d_nnz(nx, ny, 0, n)
    rp64 = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ci = torch.empty(nnz, dtype=torch.int64, device="cuda")
    va = torch.empty(nnz, dtype=torch.float64, device="cuda")
    hp._capi.call("hpcla_gen_poisson2d", nx, ny, 0, n, rp64.data_ptr(), ci.data_ptr(), va.data_ptr(), s)
    rp, cv = rp64.to(torch.int32), ci.to(torch.int32)
    B = torch.empty((k, n), dtype=torch.float64, device="cuda")
    hp._capi.call("hpcla_fill_uniform_f64", B.data_ptr(), 0, n * k, 99, s)
    desc = torch.empty(lib.hpcla_spmm_runs_desc_bytes(n), dtype=torch.uint8, device="cuda")
    n_fit = ctypes.c_int64(-1)
    hp._capi.call("hpcla_spmm_runs_build_i32", rp.data_ptr(), cv.data_ptr(), n, nnz, 0, n, desc.data_ptr(), ctypes.byref(n_fit), s)
    assert n_fit.value == n // 64
    C0 = torch.full((k, n), float("nan"), dtype=torch.float64, device="cuda")
    hp._capi.call("hpcla_spmm_split_colmajor_f64_i32", rp.data_ptr(), cv.data_ptr(), va.data_ptr(), B.data_ptr(), n, None, k, n, C0.data_ptr(), n,
                  n, nnz, k, 0, None, 0, s)
    for group in (0, 32):
        hp._capi.call("hpcla_spmm_block_order_hint", rp.data_ptr(), group)
        C1 = torch.full((k, n), float("nan"), dtype=torch.float64, device="cuda")
        hp._capi.call("hpcla_spmm_runs_colmajor_k16_f64_i32", rp.data_ptr(), cv.data_ptr(), va.data_ptr(), B.data_ptr(), n, None, 0, n,
                      C1.data_ptr(), n, n, nnz, 0, desc.data_ptr(), None, 0, s)
        assert torch.equal(C0, C1)
    hp._capi.call("hpcla_spmm_block_order_hint", rp.data_ptr(), 0)
    # the plan-time tuner: the same launch under five block orders; C holds the product afterwards, the chosen group stays
    # set for this rowptr (a bijection of the row blocks whichever it is), and a following product runs under it
    chosen = ctypes.c_int(-1)
    C2 = torch.full((k, n), float("nan"), dtype=torch.float64, device="cuda")
    hp._capi.call("hpcla_spmm_runs_colmajor_tune_block_order_f64_i32", rp.data_ptr(), cv.data_ptr(), va.data_ptr(), B.data_ptr(), n, None, 0, n,
                  C2.data_ptr(), n, n, nnz, 0, desc.data_ptr(), None, 0, s, ctypes.byref(chosen))
    assert chosen.value in (1, 8, 32, 128, 512), chosen.value
    assert torch.equal(C0, C2)
    C2.fill_(float("nan"))
    hp._capi.call("hpcla_spmm_runs_colmajor_k16_f64_i32", rp.data_ptr(), cv.data_ptr(), va.data_ptr(), B.data_ptr(), n, None, 0, n,
                  C2.data_ptr(), n, n, nnz, 0, desc.data_ptr(), None, 0, s)
    assert torch.equal(C0, C2)
    hp._capi.call("hpcla_spmm_block_order_hint", rp.data_ptr(), 0)
    # a small launch (< 4096 blocks) is run once under the natural order; bad arguments are refused like the product's
    few = torch.arange(100, dtype=torch.int32, device="cuda")
    C2.fill_(float("nan"))
    hp._capi.call("hpcla_spmm_runs_colmajor_tune_block_order_f64_i32", rp.data_ptr(), cv.data_ptr(), va.data_ptr(), B.data_ptr(), n, None, 0, n,
                  C2.data_ptr(), n, n, nnz, 0, desc.data_ptr(), few.data_ptr(), 100, s, ctypes.byref(chosen))
    assert chosen.value == 1
    assert torch.equal(C0[:, :6400], C2[:, :6400]) and bool(torch.isnan(C2[:, 6400:]).all())
    with pytest.raises(hp._capi.HPCLAError, match="even leading dimension"):
        hp._capi.call("hpcla_spmm_runs_colmajor_tune_block_order_f64_i32", rp.data_ptr(), cv.data_ptr(), va.data_ptr(), B.data_ptr(), n + 1, None, 0,
                      n, C2.data_ptr(), n, n, nnz, 0, desc.data_ptr(), None, 0, s, ctypes.byref(chosen))
