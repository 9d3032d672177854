"""What a column-major caller (Julia's Matrix) pays for A*B today: the Float64 product on row-major rows plus the two
layout conversions around it, against the entry called with column-major operands directly.

    python benchmarks/bench_colmajor.py [--nx 4096]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--dim3", default="", help="nx,ny,nz: the 7-point matrix instead of the 5-point one")
    ap.add_argument("--ny", type=int, default=0, help="grid lines (default nx / 2); a non-power-of-two count takes the column stride off the powers of two")
    ap.add_argument("--no-check", action="store_true", help="skip the equality checks (ablation builds)")
    ap.add_argument("--groups", default="0,8,32,128")
    ap.add_argument("--runs-only", action="store_true", help="only the run-tile kernels, no settle loop (for rocprofv3 --pmc passes)")
    args = ap.parse_args()
    import torch
    import hpcla_amd as hp
    L = hp._capi.load()
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    nx, ny, k = args.nx, (args.ny or args.nx // 2), 16
    if args.dim3:
        mx, my, mz = (int(v) for v in args.dim3.split(","))
        n = mx * my * mz
        nnz = L.hpcla_poisson3d_nnz(mx, my, mz, 0, n)
    else:
        n = nx * ny
        nnz = L.hpcla_poisson2d_nnz(nx, ny, 0, n)
    rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
    cv = torch.empty(nnz, dtype=torch.int64, device=dev)
    nz = torch.empty(nnz, dtype=torch.float64, device=dev)
    if args.dim3:
        hp._capi.call("hpcla_gen_poisson3d", mx, my, mz, 0, n, rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), s)
    else:
        hp._capi.call("hpcla_gen_poisson2d", nx, ny, 0, n, rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), s)
    rp, cv = rp.int(), cv.int()
    Bc = torch.rand(k, n, dtype=torch.float64, device=dev)          # column-major n x k
    Cc = torch.empty(k, n, dtype=torch.float64, device=dev)
    Br = torch.empty(n, k, dtype=torch.float64, device=dev)
    Cr = torch.empty(n, k, dtype=torch.float64, device=dev)
    ROW, COL = hp._capi.LAYOUT_ROW, hp._capi.LAYOUT_COL

    def timed(fn, reps):
        for _ in range(3):
            fn()
        t_end = time.time() + (0.0 if args.runs_only else 0.25)
        while time.time() < t_end:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def to_row():
        hp._capi.call("hpcla_transpose_f64", Bc.data_ptr(), n, COL, Br.data_ptr(), k, ROW, n, k, s)

    def to_col():
        hp._capi.call("hpcla_transpose_f64", Cr.data_ptr(), k, ROW, Cc.data_ptr(), n, COL, n, k, s)

    def prod_row():
        hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Br.data_ptr(), k, ROW, Cr.data_ptr(), k, ROW,
                      n, nnz, k, 0, s)

    def prod_col():
        hp._capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bc.data_ptr(), n, COL, Cc.data_ptr(), n, COL,
                      n, nnz, k, 0, s)

    alg = nnz * 12 + (n + 1) * 4 + 16 * k * n
    for name, fn in () if args.runs_only else (("B column-major -> row-major (hpcla_transpose_f64)", to_row), ("product on row-major rows (gather kernel)", prod_row),
                     ("C row-major -> column-major", to_col),
                     ("all three (what the Julia extension does today)", lambda: (to_row(), prod_row(), to_col())),
                     ("product on column-major operands directly", prod_col)):
        ms = timed(fn, args.reps)
        print(f"{name:58s} {ms:8.4f} ms   {alg / ms / 1e6 / 8000:6.3f} of 8 TB/s by the product's algorithmic bytes", flush=True)
    to_row()
    prod_row()
    to_col()
    ref = Cc.clone()
    if not args.runs_only:
        prod_col()
        assert args.no_check or torch.equal(ref, Cc)
    # run tiles (k = 16, descriptors once per structure): row-major (hpcla_spmm_runs_k16_f64_*) and on the column-major blocks
    # (hpcla_spmm_runs_colmajor_k16_f64_*, round 5), by the XCD group of the block order (hpcla_spmm_block_order_hint)
    import ctypes
    desc = torch.empty(L.hpcla_spmm_runs_desc_bytes(n), dtype=torch.uint8, device=dev)
    n_fit = ctypes.c_int64(-1)
    hp._capi.call("hpcla_spmm_runs_build_i32", rp.data_ptr(), cv.data_ptr(), n, nnz, 0, n, desc.data_ptr(), ctypes.byref(n_fit), s)
    print(f"# run descriptors: {n_fit.value} of {(n + 63) // 64} blocks fit", flush=True)

    def runs_row():
        hp._capi.call("hpcla_spmm_runs_k16_f64_i32", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Br.data_ptr(), None, n, Cr.data_ptr(), n, nnz, 0,
                      desc.data_ptr(), None, 0, s)

    def runs_col():
        hp._capi.call("hpcla_spmm_runs_colmajor_k16_f64_i32", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bc.data_ptr(), n, None, 0, n,
                      Cc.data_ptr(), n, n, nnz, 0, desc.data_ptr(), None, 0, s)

    to_row()
    for group in (int(g) for g in args.groups.split(",")):
        hp._capi.call("hpcla_spmm_block_order_hint", rp.data_ptr(), group)
        for name, fn in ((f"run tiles, row-major, group {group}", runs_row), (f"run tiles, COLUMN-major, group {group}", runs_col)):
            ms = timed(fn, args.reps)
            print(f"{name:58s} {ms:8.4f} ms   {alg / ms / 1e6 / 8000:6.3f} of 8 TB/s by the product's algorithmic bytes", flush=True)
    hp._capi.call("hpcla_spmm_block_order_hint", rp.data_ptr(), 0)
    Cc.fill_(float("nan"))
    runs_col()
    assert args.no_check or torch.equal(ref, Cc)


if __name__ == "__main__":
    main()
