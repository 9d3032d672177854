"""OPT-IN tile stream for the SpMV of unstructured matrices (csrc/tiled.hip): bit-equality with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


@pytest.mark.parametrize("Ti", [np.int32, np.int64])
@pytest.mark.parametrize("tile_cols", [0, 64, 4096])
def test_spmv_tile_stream_same_bits(hp, orc, Ti, tile_cols):
    """OPT-IN tile stream (csrc/tiled.hip, HPCSparseMatrix.enable_tiled): the entries of each row group re-ordered by
    (column tile, row, column), every row still one running sum in stored order -> the oracle's bits, whatever the tile
    width (64 columns: rows spread over hundreds of tiles; default: everything in one tile).  Unstructured rows, empty rows,
    rows longer than one 64-entry step (runs that continue across steps), a long row of 5000 entries, a row count that is
    not a multiple of the group size, 1-based arrays and a split column space through the raw ABI; the copy is rebuilt when
    the values are written in place."""
    import ctypes
    import torch
    rng = np.random.default_rng(23 + tile_cols)
    n = 50_000
    lens = rng.integers(0, 40, 3001)
    lens[[0, 5, 64, 1000, 3000]] = [5000, 0, 130, 65, 777]
    lens[2000:2050] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cols = np.concatenate([np.sort(rng.choice(n, size=int(l), replace=False)) for l in lens if l]).astype(np.int64)
    vals = rng.standard_normal(len(cols))
    x = rng.standard_normal(n)
    want = orc.spmv(rowptr.astype(Ti), cols.astype(Ti), vals, x)
    backend = hp.backend_rocm_serial(np.float64, Ti)
    A = hp.HPCSparseMatrix_local(rowptr, cols, vals, n, backend)
    xv = hp.HPCVector.from_global(x, backend)
    np.testing.assert_array_equal((A @ xv).local_values(), want)
    assert A.enable_tiled(xv, tile_cols) is True
    np.testing.assert_array_equal((A @ xv).local_values(), want)
    y = xv.similar() if False else hp.HPCVector.zeros(A.row_partition, backend)
    for _ in range(3):
        hp.mul_(y, A, xv)
    np.testing.assert_array_equal(y.local_values(), want)
    # values written in place: the copy follows (torch's version counter)
    A.nzval.mul_(2.0)
    np.testing.assert_array_equal((A @ xv).local_values(), orc.spmv(rowptr.astype(Ti), cols.astype(Ti), 2.0 * vals, x))
    A.nzval.mul_(0.5)
    A.disable_tiled()
    np.testing.assert_array_equal((A @ xv).local_values(), want)
    # raw ABI: 1-based arrays, columns >= n_own from a ghost segment
    sfx = "i32" if Ti == np.int32 else "i64"
    s = torch.cuda.current_stream().cuda_stream
    n_own = 30_000
    rp, cv, nz = _t((rowptr + 1).astype(Ti)), _t((cols + 1).astype(Ti)), _t(vals)
    xo, xg = _t(x[:n_own]), _t(x[n_own:])
    h = ctypes.c_void_p()
    hp._capi.call(f"hpcla_tiled_create_{sfx}", ctypes.byref(h), rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), len(lens), len(vals), n, 1,
                  tile_cols, s)
    nbytes, ntiles, rpg = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
    hp._capi.call("hpcla_tiled_info", h, ctypes.byref(nbytes), ctypes.byref(ntiles), ctypes.byref(rpg))
    assert nbytes.value >= 14 * len(vals) and rpg.value % 64 == 0 and ntiles.value == -(-n // (tile_cols or (1 << 17)))
    yd = torch.full((len(lens),), float("nan"), dtype=torch.float64, device="cuda")
    hp._capi.call("hpcla_spmv_tiled_f64", h, xo.data_ptr(), xg.data_ptr(), n_own, yd.data_ptr(), s)
    np.testing.assert_array_equal(yd.cpu().numpy(), want)
    with pytest.raises(hp._capi.HPCLAError):                    # a column space larger than x_own needs the ghost segment
        hp._capi.call("hpcla_spmv_tiled_f64", h, xo.data_ptr(), None, n_own, yd.data_ptr(), s)
    hp._capi.call("hpcla_tiled_destroy", h)


