#!/usr/bin/env python3
"""SpMM rate against the number of dense columns k (5-point matrix 4096 x 2048 rows; row-major B / C on the device):
ms, GFLOP/s and the fraction of the 8 TB/s HBM peak by the algorithmic bytes 12 nnz + 4 rows + 8k rows (C) + 8k rows (B)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import hpcla_amd as hp
    from hpcla_amd import workloads as wl
    from benchmarks.extra_workloads import device_stencil
    backend = hp.backend_rocm_serial(np.float64, np.int32)
    nx, ny = 4096, 2048
    A = device_stencil(hp, torch, backend, (nx, ny), 0, nx * ny)
    n = nx * ny
    s = torch.cuda.current_stream().cuda_stream
    print(f"# 5-point matrix {nx}x{ny}: rows={n} nnz={A.nnz}")
    print(f"{'k':>4} {'ms':>9} {'GFLOP/s':>10} {'GB/s alg':>10} {'frac':>7}")
    for k in [int(a) for a in (sys.argv[1:] or "1 2 3 4 6 8 10 12 14 16 24 32".split())]:
        Bl = torch.empty((n, k), dtype=torch.float64, device="cuda")
        hp._capi.call("hpcla_fill_uniform_f64", Bl.data_ptr(), 0, n * k, wl.SEED_X, s)
        B = hp.HPCMatrix_local(Bl, backend)
        C = A @ B
        for _ in range(5):
            C = A @ B
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            C = A @ B
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        b_alg = wl.spmm_algorithmic_bytes(A.nnz, n, n, k, 4)
        print(f"{k:>4} {ms:>9.4f} {2.0 * k * A.nnz / ms / 1e6:>10.1f} {b_alg / ms / 1e6:>10.1f} {b_alg / ms / 1e6 / 8000:>7.3f}")
        del Bl, B, C
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
