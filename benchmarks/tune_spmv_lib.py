#!/usr/bin/env python3
"""Times whole-library variants (benchmarks/tune/libhpcla_exp<N>.so = the production library with spmv.hip rebuilt under
-DHPCLA_EXP=N, see build_spmv_lib_variants.sh) against the shipped library, interleaved in ONE process, on config 2's
matrix (4096^2, 5-point) and config 4's per-GPU slab (512 x 512 x 64, 7-point): plain SpMV entry and the fused
SpMV + x.y entry CG uses.  Every variant's y must be bit-identical to the shipped library's."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,1,2,4,5,7")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dims", default="2,3")
    ap.add_argument("--orders", default="", help="e.g. 1,4,16,64: the shipped library with hpcla_spmv_block_order_hint(G)")
    args = ap.parse_args()
    import torch
    import hpcla_amd as hp
    from hpcla_amd import workloads as wl
    from benchmarks.extra_workloads import device_stencil
    backend = hp.backend_rocm_serial(np.float64, np.int32)
    s = torch.cuda.current_stream().cuda_stream
    names = ["prod"] + [f"exp{v}" for v in args.variants.split(",") if v]
    libs = {"prod": hp._capi.load()}
    orders = [int(g) for g in args.orders.split(",") if g]      # the shipped library under explicit block orders ("g<G>")
    for nm in names[1:]:
        lib = ctypes.CDLL(os.path.join(ROOT, "benchmarks", "tune", f"libhpcla_spmv_{nm}.so"), mode=ctypes.RTLD_LOCAL)
        for fn in ("hpcla_spmv_csr_f64_i32", "hpcla_spmv_dist_dot_f64_i32", "hpcla_spmv_split_f64_i32"):
            getattr(lib, fn).argtypes = getattr(libs["prod"], fn).argtypes
            getattr(lib, fn).restype = ctypes.c_int
        libs[nm] = lib
    out = {}
    for dim in [d for d in args.dims.split(",") if d]:
        if dim == "r":                       # config 5's matrix times ONE vector: 2 097 152 rows x 29.8 random columns of 2^24
            dims, n, ncols = "sprand", 2_097_152, 2_097_152 * 8
            gen = torch.Generator(device="cuda")
            gen.manual_seed(0xA11CE)
            counts = torch.poisson(torch.full((n,), 29.8, dtype=torch.float64, device="cuda"), generator=gen).to(torch.int64)
            rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
            torch.cumsum(counts, 0, out=rowptr[1:])
            nnz_r = int(rowptr[-1].item())
            cols = torch.randint(0, ncols, (nnz_r,), generator=gen, device="cuda", dtype=torch.int64)
            rowid = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int64), counts)
            key = torch.sort(rowid * ncols + cols).values
            cols = key - rowid * ncols
            del key, rowid, counts
            vals = torch.rand(nnz_r, generator=gen, device="cuda", dtype=torch.float64)
            A = hp.HPCSparseMatrix_local_device(rowptr, cols, vals, ncols, backend, col_window=(0, ncols - 1))
            del cols
            x = hp.HPCVector.zeros(np.array([0, ncols]), backend)
            hp._capi.call("hpcla_fill_uniform_f64", x.v.data_ptr(), 0, ncols, wl.SEED_X, s)
        else:
            dims = {"2": (4096, 4096), "3": (512, 512, 64), "8": (8192, 8192), "1": (1000, 16000), "4": (256, 256, 256)}[dim]
            n = int(np.prod(dims))
            A = device_stencil(hp, torch, backend, dims, 0, n)
            x = hp.HPCVector.zeros(A.row_partition, backend)
            hp._capi.call("hpcla_fill_uniform_f64", x.v.data_ptr(), 0, n, wl.SEED_X, s)
        plan = hp.get_vector_plan(A, x)
        cv = plan.colval_split
        nnz = A.nnz
        y_ref = torch.empty(n, dtype=torch.float64, device="cuda")
        y = torch.empty_like(y_ref)
        dot_out = torch.zeros(1, dtype=torch.float64, device="cuda")
        dot_work = torch.empty(libs["prod"].hpcla_spmv_dot_work_bytes(n) // 8 + 1, dtype=torch.float64, device="cuda")
        b_alg = wl.spmv_algorithmic_bytes(nnz, n, A.ncols_compressed, 4)

        ghost = torch.zeros(16, dtype=torch.float64, device="cuda")

        hint = {"g": None}

        def launch(nm, fused, yy):
            if nm.startswith("g"):
                G = int(nm[1:])
                if hint["g"] != G:
                    libs["prod"].hpcla_spmv_block_order_hint(ctypes.c_void_p(A.rowptr_target.data_ptr()), G)
                    hint["g"] = G
                nm = "prod"
            elif nm == "prod" and hint["g"] != "plan":
                libs["prod"].hpcla_spmv_block_order_hint(ctypes.c_void_p(A.rowptr_target.data_ptr()), plan.block_group)
                hint["g"] = "plan"
            lib = libs[nm]
            if fused == "split":             # the split-column kernel (ghost select per entry), every column owned
                return lib.hpcla_spmv_split_f64_i32(A.rowptr_target.data_ptr(), cv.data_ptr(), A.nzval.data_ptr(), x.v.data_ptr(),
                                                    ghost.data_ptr(), n, yy.data_ptr(), n, nnz, 0, None, 0, s)
            if fused:
                return lib.hpcla_spmv_dist_dot_f64_i32(None, None, A.rowptr_target.data_ptr(), cv.data_ptr(), A.nzval.data_ptr(),
                                                       x.v.data_ptr(), n, yy.data_ptr(), n, nnz, 0, None, 0, None, 0,
                                                       dot_out.data_ptr(), dot_work.data_ptr(), s)
            if x_gathered is not None:
                return lib.hpcla_spmv_csr_f64_i32(A.rowptr_target.data_ptr(), A.colval_target().data_ptr(), A.nzval.data_ptr(),
                                                  x_gathered.data_ptr(), yy.data_ptr(), n, nnz, 0, s)
            return lib.hpcla_spmv_csr_f64_i32(A.rowptr_target.data_ptr(), cv.data_ptr(), A.nzval.data_ptr(), x.v.data_ptr(),
                                              yy.data_ptr(), n, nnz, 0, s)
        print(f"# {dim}: {dims} n={n} nnz={nnz} B_alg={b_alg}; the plan chose block group {plan.block_group} (prod runs with it)")
        print(f"{'variant':>8} {'entry':>6} {'median_ms':>10} {'min_ms':>10} {'frac_8TB':>9} exact")
        x_gathered = None                    # (one rank: the plan's split column space IS x.v's index space)
        for fused in ((False,) if dim == "r" else (False, True, "split")):
            assert launch("prod", fused, y_ref) == 0
            torch.cuda.synchronize()
            dref = float(dot_out.item())
            run_names = names + [f"g{G}" for G in orders]
            exact, times = {}, {nm: [] for nm in run_names}
            for nm in run_names:
                y.fill_(float("nan"))
                rc = launch(nm, fused, y)
                assert rc == 0, (nm, rc)
                torch.cuda.synchronize()
                exact[nm] = bool(torch.equal(y, y_ref)) and (fused is not True or float(dot_out.item()) == dref)
            for _ in range(args.rounds):
                for nm in run_names:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(args.reps):
                        launch(nm, fused, y)
                    b.record()
                    torch.cuda.synchronize()
                    times[nm].append(a.elapsed_time(b) / args.reps)
            for nm in run_names:
                med, mn = float(np.median(times[nm])), float(np.min(times[nm]))
                ename = "split" if fused == "split" else "dot" if fused else "plain"
                out[f"{dim}d/{ename}/{nm}"] = dict(median_ms=med, min_ms=mn, exact=exact[nm])
                print(f"{nm:>8} {ename:>6} {med:>10.4f} {mn:>10.4f} {b_alg / med / 1e6 / 8000:>9.3f} {exact[nm]}")
        del A, x, plan, y, y_ref
        hp.clear_plan_cache()
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
