#!/usr/bin/env python3
"""Odd k: the generic-stride SpMM kernel (csrc/spmm.hip spmm_rowblock_kernel: odd pitches, strided operands) against the vector
kernel on the padded pitch k + 1 (round 6) and against the next even k, on the 5-point
matrix 4096 x 2048 rows and on a random pattern (2^20 rows x 2^22 columns, ~30 entries per row): ms per product by k.

    python benchmarks/bench_spmm_generic.py [--ks 15,13,7,3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="15,13,7,3")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import hpcla_amd as hp
    L = hp._capi.load()
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    ROW = hp._capi.LAYOUT_ROW

    def timed(fn, reps):
        for _ in range(3):
            fn()
        t_end = time.time() + 0.25
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

    nx, ny = 4096, 2048
    n = nx * ny
    nnz = L.hpcla_poisson2d_nnz(nx, ny, 0, n)
    rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
    cv = torch.empty(nnz, dtype=torch.int64, device=dev)
    nz = torch.empty(nnz, dtype=torch.float64, device=dev)
    hp._capi.call("hpcla_gen_poisson2d", nx, ny, 0, n, rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), s)
    cases = [("5-point 4096 x 2048", rp.int(), cv.int(), nz, n, n, nnz)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    m, nc = 1 << 20, 1 << 22
    counts = torch.poisson(torch.full((m,), 29.8, dtype=torch.float64, device=dev), generator=gen).to(torch.int64)
    rpr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rpr[1:])
    nnzr = int(rpr[-1].item())
    rowid = torch.repeat_interleave(torch.arange(m, device=dev, dtype=torch.int64), counts)
    key = torch.sort(rowid * nc + torch.randint(0, nc, (nnzr,), generator=gen, device=dev, dtype=torch.int64)).values
    cases.append(("random 2^20 x 2^22, ~30 per row", rpr.int(), (key - rowid * nc).int(), torch.rand(nnzr, generator=gen, device=dev, dtype=torch.float64),
                  m, nc, nnzr))
    COL = hp._capi.LAYOUT_COL
    for name, rp_, cv_, nz_, nr, ncol, nnz_ in cases:
        for k in (int(v) for v in args.ks.split(",")):
            kp = k + (k & 1)
            # (label, k of the product, row pitch of B and of a row-major C, C column-major?)
            variants = [(f"k={k:3d} pitch {k:3d} (round 5: generic kernel for odd k)", k, k, False)]
            if k & 1:
                variants += [(f"k={k:3d} pitch {kp:3d} (round 6: vector kernel, last pair masked)", k, kp, False),
                             (f"k={k:3d} pitch {kp:3d}, column-major C (CCOL store)", k, kp, True),
                             (f"k={kp:3d} pitch {kp:3d} (the next even k)", kp, kp, False)]
            base = {}
            for label, kk, pitch, ccol in variants:
                B = torch.rand(ncol, pitch, dtype=torch.float64, device=dev)
                C = torch.empty((kk, nr) if ccol else (nr, pitch), dtype=torch.float64, device=dev)

                def prod():
                    hp._capi.call("hpcla_spmm_csr_f64_i32", rp_.data_ptr(), cv_.data_ptr(), nz_.data_ptr(), B.data_ptr(), pitch, ROW,
                                  C.data_ptr(), nr if ccol else pitch, COL if ccol else ROW, nr, nnz_, kk, 0, s)
                ms = timed(prod, args.reps)
                base[(kk, ccol)] = ms
                alg = nnz_ * 12 + (nr + 1) * 4 + 8 * kk * (nr + min(ncol, nnz_))
                print(f"{name:34s} {label:58s} {ms:8.4f} ms   {alg / ms / 1e6 / 8000:6.3f} of 8 TB/s (each B row once)", flush=True)
                del B, C
            if k & 1:
                print(f"{name:34s} k={k}: padded-pitch / next-even = {base[(k, False)] / base[(kp, False)]:.3f}  (bar: <= 1.15)", flush=True)


if __name__ == "__main__":
    main()
