#!/usr/bin/env python3
"""Price of the long-row cliff: an ARROW matrix -- the 5-point Poisson matrix N x N plus ONE dense row
and ONE dense column -- through the default SpMV (every row one lane's sequential sum in stored order: the reference's bits,
and the cliff of its one-work-item-per-row kernel, src/sparse.jl:2055-2066) and through the OPT-IN long-row path
(HPCSparseMatrix.enable_long_rows: the dense row in tree order, csrc/spmv.hip LONGR).

    python benchmarks/bench_arrow.py [--size 4096] [--reps 10]

Prints one JSON line: ms per product of both paths, the byte bound (SURVEY 8d algorithmic bytes at 8 TB/s and at the
6.3 TB/s copy ceiling), parity of the default path against the CPU oracle (bits) and of the opt-in path (bits on short rows,
1e-12 (|A||x|)_r on the dense row).  One GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--min-len", type=int, default=4096)
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    import torch
    import hpcla_amd as hp
    from hpcla_amd import workloads as wl
    N = args.size
    n = N * N
    dev = "cuda"
    backend = hp.backend_rocm_serial(np.float64, np.int32)
    s0 = torch.cuda.current_stream().cuda_stream
    nnz_p = hp._capi.load().hpcla_poisson2d_nnz(N, N, 0, n)
    rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ci = torch.empty(nnz_p, dtype=torch.int64, device=dev)
    va = torch.empty(nnz_p, dtype=torch.float64, device=dev)
    hp._capi.call("hpcla_gen_poisson2d", N, N, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), s0)
    r0, c0 = n // 2 + 3, n // 2 - 5                      # the dense row and the dense column
    rows_p = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), rp[1:] - rp[:-1])
    allr = torch.arange(n, device=dev, dtype=torch.int64)
    # keys (row * n + col): the stencil, then the dense row, then the dense column; duplicates keep the FIRST (the stencil's value)
    key = torch.cat([rows_p * n + ci, r0 * n + allr, allr * n + c0])
    val = torch.cat([va, 1.0e-3 + 1.0e-3 * torch.rand(n, device=dev, dtype=torch.float64), torch.full((n,), 0.5, device=dev, dtype=torch.float64)])
    del rows_p, ci, va
    order = torch.argsort(key, stable=True)
    key, val = key[order], val[order]
    del order
    keep = torch.ones_like(key, dtype=torch.bool)
    keep[1:] = key[1:] != key[:-1]
    key, val = key[keep], val[keep].contiguous()
    del keep
    rows = key // n
    cols = (key - rows * n).contiguous()
    del key
    counts = torch.bincount(rows, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    nnz = int(val.numel())
    del rows, counts
    A = hp.HPCSparseMatrix_local_device(rowptr, cols, val, n, backend, col_window=(0, n - 1))
    x = hp.HPCVector.zeros(A.row_partition, backend)
    hp._capi.call("hpcla_fill_uniform_f64", x.v.data_ptr(), 0, n, wl.SEED_X, s0)
    y = hp.HPCVector.zeros(A.row_partition, backend)

    def timed(reps):
        hp.mul_(y, A, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hp.mul_(y, A, x)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t0 = time.perf_counter()
    ms_default = timed(max(args.reps // 3, 2))           # each product may take tens of ms here
    y_default = y.v.clone()
    n_long = A.enable_long_rows(args.min_len)
    ms_long = timed(args.reps)
    y_long = y.v.clone()
    A.disable_long_rows()
    b_alg = wl.spmv_algorithmic_bytes(nnz, n, A.ncols_compressed, 4)
    out = {"workload": f"arrow: poisson2d 5-pt {N}x{N} + dense row {r0} + dense column {c0}, n={n}, nnz={nnz}, index=i32",
           "default_ms": round(ms_default, 4), "opt_in_long_rows_ms": round(ms_long, 4), "long_rows": n_long, "min_len": args.min_len,
           "algorithmic_bytes": b_alg, "byte_bound_ms_at_8TBs": round(b_alg / 8.0e12 * 1e3, 4),
           "byte_bound_ms_at_copy_ceiling_6p3TBs": round(b_alg / 6.3e12 * 1e3, 4),
           "default_over_byte_bound": round(ms_default / (b_alg / 8.0e12 * 1e3), 1),
           "opt_in_over_byte_bound": round(ms_long / (b_alg / 8.0e12 * 1e3), 2),
           "opt_in_frac_of_peak": round(b_alg / (ms_long * 1e-3) / 8.0e12, 4)}
    same_short = torch.ones(n, dtype=torch.bool, device=dev)
    same_short[r0] = False
    out["short_rows_same_bits_both_paths"] = bool(torch.equal(y_default[same_short], y_long[same_short]))
    if not args.no_oracle:
        from oracle import oracle as orc
        rp_h, cv_h, nz_h = A.rowptr.astype(np.int32), A.colval.astype(np.int32), A.nzval.cpu().numpy()
        xg = x.v.cpu().numpy()[A.col_indices]
        want = orc.spmv(rp_h, cv_h, nz_h, xg)
        bound = orc.spmv(rp_h, cv_h, np.abs(nz_h), np.abs(xg))
        out["default_bits_equal_oracle"] = bool(np.array_equal(y_default.cpu().numpy(), want))
        yl = y_long.cpu().numpy()
        out["opt_in_dense_row_err_over_absAx"] = float(abs(yl[r0] - want[r0]) / bound[r0])
        out["opt_in_within_1e-12_absAx"] = bool(abs(yl[r0] - want[r0]) <= 1e-12 * bound[r0])
    out["wall_s"] = round(time.perf_counter() - t0, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
