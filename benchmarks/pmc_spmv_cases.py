#!/usr/bin/env python3
"""One process, several SpMV CONTEXTS, a fixed number of launches each -- the workload of the round-4 counter study
(why does the 3-D 7-point SpMV(+p.Ap) inside the CG loop run at 0.636 of peak against 0.727 for the
2-D 5-point one, at the same ~1.05x traffic?).

Cases, in this order (every launch of `spmv_rowblock_quad_kernel` in the process belongs to exactly one of them; the
plan-time block-order measurement is switched off by fixing the order):

  2d      4096 x 4096 5-point, y = A x                       (the headline kernel)
  2ddot   the same with the x.y epilogue (hpcla_spmv_dist_dot)
  3d      512 x 512 x 64 7-point slab (config 4's per-GPU share), y = A x
  3ddot   the same with the epilogue
  cg      the 3-D SpMV + epilogue INSIDE the fused CG loop (hpcla_cg_iterations: x just rewritten by cg_direction_kernel)

Run under `rocprofv3 --pmc ... -- python3 benchmarks/pmc_spmv_cases.py --manifest FILE`; benchmarks/pmc_spmv_table.py
splits the kernel's dispatches by the manifest and prints the counters side by side.  Without a profiler it prints
HIP-event times per case."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--cases", default="2d,2ddot,3d,3ddot,cg")
    ap.add_argument("--order2d", default="32")
    ap.add_argument("--order3d", default="64")
    ap.add_argument("--manifest", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import hpcla_amd as hp
    from hpcla_amd import workloads as wl
    from benchmarks.extra_workloads import device_stencil

    backend = hp.backend_rocm_serial(np.float64, np.int32)
    s = torch.cuda.current_stream().cuda_stream
    cases = [c for c in args.cases.split(",") if c]
    manifest, times = [], {}

    def build(dims, order):
        os.environ["HPCLA_BLOCK_ORDER"] = order          # fixed: no measurement launches of the same kernel
        n = int(np.prod(dims))
        A = device_stencil(hp, torch, backend, dims, 0, n)
        x = hp.HPCVector.zeros(A.row_partition, backend)
        hp._capi.call("hpcla_fill_uniform_f64", x.v.data_ptr(), 0, n, wl.SEED_X, s)
        y = hp.HPCVector.zeros(A.row_partition, backend)
        return A, x, y

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    for dims, order, names in (((4096, 4096), args.order2d, ("2d", "2ddot")), ((512, 512, 64), args.order3d, ("3d", "3ddot", "cg"))):
        if not any(c in cases for c in names):
            continue
        A, x, y = build(dims, order)
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        plain, dot = names[0], names[1]
        if plain in cases:
            times[plain] = timed(lambda: hp.mul_(y, A, x), args.reps)
            manifest.append({"case": plain, "launches": args.reps, "nnz": A.nnz, "rows": A.nrows_local,
                             "algorithmic_bytes": wl.spmv_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, 4)})
        if dot in cases:
            times[dot] = timed(lambda: hp.mul_dot_(y, A, x, out), args.reps)
            manifest.append({"case": dot, "launches": args.reps, "nnz": A.nnz, "rows": A.nrows_local,
                             "algorithmic_bytes": wl.spmv_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, 4)})
        if "cg" in names and "cg" in cases:
            ws = hp.CGWorkspace(x, args.reps + 2)
            plan, fused = hp.cg_setup(A, x, ws)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            hp.cg_iterate(A, ws, plan, fused, args.reps)
            e1.record()
            torch.cuda.synchronize()
            times["cg"] = e0.elapsed_time(e1) / args.reps       # the whole iteration
            manifest.append({"case": "cg", "launches": args.reps, "nnz": A.nnz, "rows": A.nrows_local,
                             "algorithmic_bytes": wl.spmv_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, 4)})
            del ws
        del A, x, y
        hp.clear_plan_cache()
        torch.cuda.empty_cache()
    if args.manifest:
        json.dump({"cases": manifest, "event_ms": times, "order2d": args.order2d, "order3d": args.order3d,
                   "kernel": "spmv_rowgather_kernel" if hp._capi.load().hpcla_get_spmv_kernel() == 0 else "spmv_rowblock_quad_kernel"},
                  open(args.manifest, "w"))
    print(json.dumps({"event_ms_per_launch": {k: round(v, 5) for k, v in times.items()}}))


if __name__ == "__main__":
    main()
