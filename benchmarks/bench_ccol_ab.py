#!/usr/bin/env python3
"""A/B of the unstructured SpMM's two C layouts in ONE process (config 5's per-GPU share, k = 16): hpcla_spmm_csr_f64_i32 with
row-major B and C row-major (the host layer's product) against C column-major (CCOL store: the Julia extension's product),
alternating, device time per launch.  Prints both and the ratio."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import hpcla_amd as hp
    dev = "cuda"
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xA11CE)
    n, ncols, k = 2_097_152, 16_777_216, 16
    counts = torch.poisson(torch.full((n,), 29.8, dtype=torch.float64, device=dev), generator=gen).to(torch.int64)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    nnz = int(rowptr[-1].item())
    rowid = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), counts)
    key = torch.sort(rowid * ncols + torch.randint(0, ncols, (nnz,), generator=gen, device=dev, dtype=torch.int64)).values
    cols = (key - rowid * ncols).to(torch.int32)
    del key, rowid, counts
    rp = rowptr.to(torch.int32)
    vals = torch.rand(nnz, generator=gen, device=dev, dtype=torch.float64)
    B = torch.rand((ncols, k), generator=gen, device=dev, dtype=torch.float64)
    Cr = torch.empty((n, k), dtype=torch.float64, device=dev)
    Cc = torch.empty((k, n), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    ROW, COL = hp._capi.LAYOUT_ROW, hp._capi.LAYOUT_COL

    def row():
        hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cols.data_ptr(), vals.data_ptr(), B.data_ptr(), k, ROW, Cr.data_ptr(), k, ROW,
                      n, nnz, k, 0, s)

    def col():
        hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cols.data_ptr(), vals.data_ptr(), B.data_ptr(), k, ROW, Cc.data_ptr(), n, COL,
                      n, nnz, k, 0, s)

    def timed(fn, reps=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    for _ in range(5):
        row(); col()
    torch.cuda.synchronize()
    assert torch.equal(Cc.t(), Cr)
    rs, cs = [], []
    for _ in range(6):
        rs.append(timed(row)); cs.append(timed(col))
    print(f"row-major C: {np.median(rs):.4f} ms (min {min(rs):.4f})   column-major C (CCOL): {np.median(cs):.4f} ms (min {min(cs):.4f})   "
          f"ratio {np.median(cs) / np.median(rs):.4f}")


if __name__ == "__main__":
    main()
