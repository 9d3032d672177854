"""Float32 SpMV / SpMM through the raw C ABI on the headline matrices (csrc/f32.hip), next to the Float64 kernels.

    python benchmarks/bench_f32.py [--nx 4096] [--reps 200]

Prints one line per case: ms per launch (HIP events on the launch stream over `reps` launches after a settled warm-up)
and the fraction of 8 TB/s by the algorithmic bytes of SURVEY 8d restated for 4-byte values:
SpMV  nnz*(4+4) + (nrows+1)*4 + 4*nrows + 4*ncols;   SpMM(k)  nnz*8 + (nrows+1)*4 + 4k*nrows + 4k*ncols.
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only", default="", help="spmv2d | spmv3d | spmm: run one case (profiler passes)")
    ap.add_argument("--settle-ms", type=float, default=250.0, help="warm-up time before each timed loop (0 under a profiler)")
    ap.add_argument("--no-f64", action="store_true", help="skip the Float64 twins")
    args = ap.parse_args()
    import torch
    import hpcla_amd as hp
    L = hp._capi.load()
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream

    def gen2d(nx, ny):
        n = nx * ny
        nnz = L.hpcla_poisson2d_nnz(nx, ny, 0, n)
        rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
        cv = torch.empty(nnz, dtype=torch.int64, device=dev)
        nz = torch.empty(nnz, dtype=torch.float64, device=dev)
        hp._capi.call("hpcla_gen_poisson2d", nx, ny, 0, n, rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), s)
        return n, nnz, rp.int(), cv.int(), nz

    def gen3d(nx, ny, nz_):
        n = nx * ny * nz_
        nnz = L.hpcla_poisson3d_nnz(nx, ny, nz_, 0, n)
        rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
        cv = torch.empty(nnz, dtype=torch.int64, device=dev)
        nz = torch.empty(nnz, dtype=torch.float64, device=dev)
        hp._capi.call("hpcla_gen_poisson3d", nx, ny, nz_, 0, n, rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), s)
        return n, nnz, rp.int(), cv.int(), nz

    def timed(fn, reps):
        for _ in range(3):
            fn()
        t_end = time.time() + args.settle_ms * 1e-3    # settled clocks (the launch itself decides how many warm-ups that is)
        while time.time() < t_end:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def report(tag, ms, nbytes, flops):
        print(f"{tag:44s} {ms:8.4f} ms  {nbytes / ms / 1e6:8.1f} GB/s  {nbytes / ms / 1e6 / 8000:6.3f} of 8 TB/s  "
              f"{flops / ms / 1e6:8.1f} GFLOP/s", flush=True)

    cases = []
    if args.only in ("", "spmv2d"):
        cases.append(("poisson2d %d^2" % args.nx, gen2d(args.nx, args.nx)))
    if args.only in ("", "spmv3d"):
        cases.append(("poisson3d 512x512x64", gen3d(512, 512, 64)))
    for name, (n, nnz, rp, cv, nz64) in cases:
        nz32 = nz64.float()
        x64 = torch.rand(n, dtype=torch.float64, device=dev)
        x32 = x64.float()
        y64 = torch.empty(n, dtype=torch.float64, device=dev)
        y32 = torch.empty(n, dtype=torch.float32, device=dev)
        hp._capi.call("hpcla_spmv_csr_f64_i32", rp.data_ptr(), cv.data_ptr(), nz64.data_ptr(), x64.data_ptr(), y64.data_ptr(), n, nnz, 0, s)
        if not args.no_f64:
            ms = timed(lambda: hp._capi.call("hpcla_spmv_csr_f64_i32", rp.data_ptr(), cv.data_ptr(), nz64.data_ptr(), x64.data_ptr(),
                                             y64.data_ptr(), n, nnz, 0, s), args.reps)
            report(f"spmv f64 {name}", ms, nnz * 12 + (n + 1) * 4 + 16 * n, 2 * nnz)
        ms = timed(lambda: hp._capi.call("hpcla_spmv_csr_f32_i32", rp.data_ptr(), cv.data_ptr(), nz32.data_ptr(), x32.data_ptr(),
                                         y32.data_ptr(), n, nnz, 0, s), args.reps)
        report(f"spmv f32 {name}", ms, nnz * 8 + (n + 1) * 4 + 8 * n, 2 * nnz)
        assert torch.allclose(y32.double(), y64, rtol=0, atol=1e-4 * 16)
        del x64, x32, y64, y32, nz32

    if args.only not in ("", "spmm"):
        return
    # SpMM: 5-point matrix x 16 columns (other_configs.poisson2d_spmm's share: nx x nx/2 rows)
    k = 16
    n, nnz, rp, cv, nz64 = gen2d(args.nx, args.nx // 2)
    nz32 = nz64.float()
    B64 = torch.rand(n, k, dtype=torch.float64, device=dev)
    B32 = B64.float()
    C64 = torch.empty(n, k, dtype=torch.float64, device=dev)
    C32 = torch.empty(n, k, dtype=torch.float32, device=dev)
    row = hp._capi.LAYOUT_ROW
    hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cv.data_ptr(), nz64.data_ptr(), B64.data_ptr(), k, row, C64.data_ptr(), k, row,
                  n, nnz, k, 0, s)
    if not args.no_f64:
        ms = timed(lambda: hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cv.data_ptr(), nz64.data_ptr(), B64.data_ptr(), k,
                                         row, C64.data_ptr(), k, row, n, nnz, k, 0, s), max(args.reps // 4, 10))
        report(f"spmm f64 k=16 row-major {args.nx}x{args.nx // 2}", ms, nnz * 12 + (n + 1) * 4 + 16 * k * n, 2 * k * nnz)
    ms = timed(lambda: hp._capi.call("hpcla_spmm_csr_f32_i32", rp.data_ptr(), cv.data_ptr(), nz32.data_ptr(), B32.data_ptr(), k,
                                     row, C32.data_ptr(), k, row, n, nnz, k, 0, s), max(args.reps // 4, 10))
    report(f"spmm f32 k=16 row-major {args.nx}x{args.nx // 2}", ms, nnz * 8 + (n + 1) * 4 + 8 * k * n, 2 * k * nnz)
    assert torch.allclose(C32.double(), C64, rtol=0, atol=1e-4 * 16)
    Bc = B32.t().contiguous()
    Cc = torch.empty(k, n, dtype=torch.float32, device=dev)
    col = hp._capi.LAYOUT_COL
    ms = timed(lambda: hp._capi.call("hpcla_spmm_csr_f32_i32", rp.data_ptr(), cv.data_ptr(), nz32.data_ptr(), Bc.data_ptr(), n,
                                     col, Cc.data_ptr(), n, col, n, nnz, k, 0, s), max(args.reps // 4, 10))
    report(f"spmm f32 k=16 column-major {args.nx}x{args.nx // 2}", ms, nnz * 8 + (n + 1) * 4 + 8 * k * n, 2 * k * nnz)
    assert torch.equal(Cc.t().contiguous(), C32)


if __name__ == "__main__":
    main()
