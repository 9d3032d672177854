#!/usr/bin/env python3
"""SpGEMM A*A on the 2-D Laplacian -- the reference's only published number for a sparse product:
n = 10 000 (nnz 49 600), 4 MPI ranks x 3 threads, median 1.216 ms (SafePETSc 0.817 ms),
tools/benchmark_vs_petsc_results.txt:3-11.  Same protocol (warm-up builds/caches the plan, then timed
repetitions, median): first product (symbolic + numeric), repeated product (cached structure)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import hpcla_amd as hp
    from hpcla_amd import workloads as wl
    from hpcla_amd.matmat import clear_matrix_plan_cache
    b = hp.backend_rocm_serial(np.float64, np.int32)
    for N, lists in [(N, m) for N in (100, 1000, 2048) for m in ("1", "0")]:
        # lists = "1": repeated products run on the per-entry product lists (hpcla_spgemm_numeric_mapped_f64, built by
        # the third product); "0": on the expand-sort-combine / hash kernels every time
        os.environ["HPCLA_SPGEMM_MAP"] = lists
        n = N * N
        rowptr, colidx, vals = wl.poisson2d_rows(N, N, 0, n)
        A = hp.HPCSparseMatrix_local(rowptr, colidx, vals, n, b)
        t0 = time.perf_counter()
        C = A @ A
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        C = A @ A
        torch.cuda.synchronize()
        second = time.perf_counter() - t0
        ts = []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            C = A @ A
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            C = A @ A
        ev1.record()
        torch.cuda.synchronize()
        print(f"laplacian2d n={n:9d} nnz(A)={A.nnz:10d} nnz(A*A)={C.nnz:11d}  {'product lists' if lists == '1' else 'numeric kernels'}: "
              f"first {first*1e3:9.3f} ms   second {second*1e3:9.3f} ms   "
              f"repeat median {np.median(ts)*1e3:8.3f} ms (min {np.min(ts)*1e3:.3f})   device/stream {ev0.elapsed_time(ev1)/20:8.3f} ms")
        clear_matrix_plan_cache()
        hp.clear_plan_cache()


if __name__ == "__main__":
    main()
