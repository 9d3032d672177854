#!/usr/bin/env python3
"""The reference's own micro-benchmark (tools/benchmark_single_rank.jl: sizes 100 / 1000 / 10000, 10
stored entries per row, one rank; it compares HPCLinearAlgebra types with native Julia and commits no
output) restated for DeviceROCm: per-call time a caller sees (call + wait for the result) and the
back-to-back device time, against numpy / scipy on the host cores as the "native" column.
Larger sizes are appended to show where the GPU path crosses over."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def best(fn, reps=200, rounds=5):
    out = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        out.append((time.perf_counter() - t0) / reps)
    return min(out)


def main():
    import scipy.sparse as sp
    import torch
    import hpcla_amd as hp
    b = hp.backend_rocm_serial(np.float64, np.int64)      # the reference's default index type
    rng = np.random.default_rng(0)
    sync = torch.cuda.synchronize
    print(f"{'n':>9} {'operation':14} {'call+wait us':>13} {'device us':>10} {'host native us':>15}")
    for n in (100, 1000, 10_000, 1_000_000, 10_000_000):
        reps = 200 if n <= 1_000_000 else 30
        vg, wg = rng.standard_normal(n), rng.standard_normal(n)
        v, w = hp.HPCVector.from_global(vg, b), hp.HPCVector.from_global(wg, b)
        k = min(10, n)
        cols = np.sort(np.argsort(rng.random((n, k if n <= 10_000 else 1)), axis=1)[:, :k], axis=1) if n <= 10_000 else \
            np.sort(rng.integers(0, n, size=(n, k)), axis=1)
        if n > 10_000:                                      # drop duplicate columns inside a row
            keep = np.ones_like(cols, dtype=bool); keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
        else:
            keep = np.ones_like(cols, dtype=bool)
        counts = keep.sum(axis=1)
        rp = np.concatenate([[0], np.cumsum(counts)])
        As = sp.csr_matrix((rng.standard_normal(int(rp[-1])), cols[keep], rp), shape=(n, n))
        A = hp.HPCSparseMatrix_from_global(As, b)
        y = A @ v
        ops = [
            ("v + w", lambda: v + w, lambda: vg + wg),
            ("alpha * v", lambda: 2.0 * v, lambda: 2.0 * vg),
            ("dot(v, w)", lambda: hp.dot(v, w), lambda: float(vg @ wg)),
            ("norm(v)", lambda: hp.norm(v), lambda: float(np.linalg.norm(vg))),
            ("sum(v)", lambda: hp.vsum(v), lambda: float(vg.sum())),
            ("sparse A*x", lambda: hp.mul_(y, A, v), lambda: As @ vg),
        ]
        if n <= 10_000:
            Md = rng.standard_normal((n, n))
            M = hp.HPCMatrix.from_global(Md, b)
            yd = M @ v
            ops.append(("dense A*x", lambda: hp.dense_matvec(M, v, yd), lambda: Md @ vg))
        for name, dev_fn, host_fn in ops:
            dev_fn(); sync()
            t_wait = best(lambda: (dev_fn(), sync()), reps)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                dev_fn()
            e.record(); sync()
            t_dev = a.elapsed_time(e) / reps * 1e-3
            t_host = best(host_fn, max(reps // 10, 3), 3)
            print(f"{n:>9} {name:14} {t_wait*1e6:13.1f} {t_dev*1e6:10.1f} {t_host*1e6:15.1f}")
        hp.clear_plan_cache(); hp.clear_dense_plan_cache()


if __name__ == "__main__":
    main()
