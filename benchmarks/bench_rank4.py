#!/usr/bin/env python3
"""Rates of the SURVEY 8f rank-4 operators on one MI355X (HIP-event timing, plans cached):
dense A*x (hpcla_gemv_rowmajor_f64), transpose(A)*x (hpcla_gemv_t_rowmajor_f64 + all-reduce),
sparse A+B (hpcla_merge_combine_f64_* / axpby fast path).  Bytes = 8 per dense entry (+ vectors);
for A+B the bytes the value pass has to move (values in/out + the index lists it reads)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(torch, fn, reps=20, rounds=5):
    ts = []
    for _ in range(rounds):
        fn()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / reps)
    return float(np.median(ts[1:]))


def main():
    import torch
    import hpcla_amd as hp
    from hpcla_amd import workloads as wl
    b = hp.backend_rocm_serial(np.float64, np.int32)
    print("# dense: rows x cols, ms, GB/s of 8*rows*cols bytes, fraction of 8 TB/s")
    for m, n in ((16384, 16384), (2_097_152, 64), (4096, 262144), (1_000_000, 16)):
        Al = torch.empty((m, n), dtype=torch.float64, device="cuda")
        hp._capi.call("hpcla_fill_uniform_f64", Al.data_ptr(), 0, m * n, 7, torch.cuda.current_stream().cuda_stream)
        A = hp.HPCMatrix_local(Al, b)
        x = hp.HPCVector.from_global(np.random.default_rng(1).random(n), b)
        xt = hp.HPCVector.from_global(np.random.default_rng(2).random(m), b)
        y = A @ x
        t = timeit(torch, lambda: hp.dense_matvec(A, x, y))
        print(f"A*x            {m:>9d} x {n:<7d} {t:8.4f} ms  {8*m*n/t/1e6:8.1f} GB/s  {8*m*n/t/1e6/8000:5.3f}")
        At = hp.transpose(A)
        t = timeit(torch, lambda: At @ xt)
        print(f"transpose(A)*x {m:>9d} x {n:<7d} {t:8.4f} ms  {8*m*n/t/1e6:8.1f} GB/s  {8*m*n/t/1e6/8000:5.3f}")
        del A, Al, At, x, xt, y
        torch.cuda.empty_cache()
    hp.clear_dense_plan_cache()

    print("# sparse A + B on the 2-D 5-point pattern, N = 4096 (83.9 M entries each)")
    N = 4096
    rp, ci, va = wl.poisson2d_rows(N, N, 0, N * N)
    A = hp.HPCSparseMatrix_local(rp, ci, va, N * N, b)
    B = hp.HPCSparseMatrix_local(rp, ci, va * 0.5, N * N, b)
    C = A + B
    t = timeit(torch, lambda: A + B, reps=10)
    nnz = A.nnz
    print(f"A+B same pattern        {t:8.4f} ms  {24*nnz/t/1e6:8.1f} GB/s of 24 B/entry (two value reads, one write)")
    # B with a different pattern: the 5-point matrix of the transposed grid ordering shares only part of it
    keep = (np.arange(len(ci)) % 3) != 0
    counts = np.add.reduceat(keep.astype(np.int64), rp[:-1])
    rp2 = np.concatenate([[0], np.cumsum(counts)])
    B2 = hp.HPCSparseMatrix_local(rp2, ci[keep], va[keep], N * N, b)
    C2 = A + B2
    t = timeit(torch, lambda: A + B2, reps=10)
    moved = 8 * (A.nnz + B2.nnz + C2.nnz)
    print(f"A+B subset pattern      {t:8.4f} ms  {moved/t/1e6:8.1f} GB/s of value bytes (8 B per entry of A, B and C; index lists extra)")


if __name__ == "__main__":
    main()
