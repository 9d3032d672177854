"""BASELINE configs 4 and 5 (and per-kernel vector-op rates): separate harness behind
``bench.py --workload {poisson3d_cg,sprand_spmm}``.  Weak scaling per GPU like bench.py:

* poisson3d_cg  -- 3-D 7-point Poisson, 512 x 512 x (64*N) grid (config 4's per-GPU share: 64 planes
  = 16 777 216 rows, ~117 M nonzeros per GPU), exactly --steps CG iterations from x0 = 0 (default 100),
  no convergence exit; reports ms/iteration, algorithmic GB/s per GPU and the final residual.
* sprand_spmm   -- unstructured sprand-like matrix, 2 097 152 rows per GPU, ~29.8 nnz/row, columns
  uniform over the global 2 097 152*N columns, B with k = 16 dense columns (row-major on device).
"""
import json
import os
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


def _sync_barrier(job, closing=False):
    job.barrier(device_only=closing)      # closing bracket of a timed region: device rendezvous only (bench.py Job)


def _measure_spmm(args, job, hp, wl, A, B, k, world, dev, setup_s, metric, workload):
    C = A @ B
    for _ in range(args.warmup):
        C = A @ B
    _sync_barrier(job)
    steps = min(args.steps, 50)
    t0 = time.perf_counter()
    for _ in range(steps):
        C = A @ B
    _sync_barrier(job, closing=True)
    elapsed = time.perf_counter() - t0
    elapsed = job.max(elapsed)
    ms = elapsed / steps * 1e3
    b_alg = wl.spmm_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, k, 4)
    b_gather = A.nnz * (12 + 8 * k) + 4 * A.nrows_local + 8 * k * A.nrows_local    # every B row read per entry
    out = {
        "metric": metric, "value": round(2.0 * k * A.nnz * world / (ms * 1e-3) / 1e9, 1),
        "unit": "GFLOP/s", "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload,
                   "ncols_compressed": A.ncols_compressed},
        "roofline": {"bound": "hbm", "achieved": round(b_alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": b_alg,
                     "gather_bytes_per_launch": b_gather,
                     "gather_gbs": round(b_gather / (ms * 1e-3) / 1e9, 1),
                     "note": "algorithmic bytes count each touched B row once; a random-column matrix re-reads B rows "
                             "(gather_bytes = one 128-byte line per stored entry), which is what HBM actually serves"},
        "setup_s": round(setup_s, 2),
    }
    return out


def device_stencil(hp, torch, backend, dims, lo, hi):
    """Rows [lo, hi) of the 5-point (dims = (nx, ny)) or 7-point (dims = (nx, ny, nz)) Laplacian, generated and
    column-compressed ON THE DEVICE (hpcla_gen_poisson2d/3d + hpcla_compress_columns_*)."""
    s0 = torch.cuda.current_stream().cuda_stream
    lib = hp._capi.load()
    n_glob = int(np.prod(dims))
    if len(dims) == 2:
        nnz = lib.hpcla_poisson2d_nnz(dims[0], dims[1], lo, hi)
        reach = dims[0]
    else:
        nnz = lib.hpcla_poisson3d_nnz(dims[0], dims[1], dims[2], lo, hi)
        reach = dims[0] * dims[1]
    rp = torch.empty(hi - lo + 1, dtype=torch.int64, device="cuda")
    ci = torch.empty(nnz, dtype=torch.int64, device="cuda")
    va = torch.empty(nnz, dtype=torch.float64, device="cuda")
    if len(dims) == 2:
        hp._capi.call("hpcla_gen_poisson2d", dims[0], dims[1], lo, hi, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), s0)
    else:
        hp._capi.call("hpcla_gen_poisson3d", dims[0], dims[1], dims[2], lo, hi, rp.data_ptr(), ci.data_ptr(),
                      va.data_ptr(), s0)
    return hp.HPCSparseMatrix_local_device(rp, ci, va, n_glob, backend,
                                           col_window=(max(lo - reach, 0), min(hi + reach, n_glob) - 1))


def run(args, backend, rank, world, job):
    out = run_record(args, backend, rank, world, job)
    return json.dumps(out) if rank == 0 else None          # bench.py prints it, then tears down


def run_record(args, backend, rank, world, job):
    import torch
    import hpcla_amd as hp
    from hpcla_amd import workloads as wl
    dev = backend.torch_device
    out = {}
    if args.workload == "poisson3d_cg":
        N = args.size or 512
        planes = (N // 8) if N >= 64 else N
        nz = planes * world
        n_glob = N * N * nz
        lo, hi = rank * N * N * planes, (rank + 1) * N * N * planes
        t0 = time.perf_counter()
        A = device_stencil(hp, torch, backend, (N, N, nz), lo, hi)
        b = hp.HPCVector.zeros(A.row_partition, backend)
        hp._capi.call("hpcla_fill_uniform_f64", b.v.data_ptr(), lo, hi - lo, wl.SEED_RHS,
                      torch.cuda.current_stream().cuda_stream)
        setup_s = time.perf_counter() - t0
        iters = args.steps if args.steps != 200 else 100
        fused = os.environ.get("HPCLA_CG_UNFUSED", "") != "1"
        graph = os.environ.get("HPCLA_CG_GRAPH", "") == "1"     # replay a captured pair of iterations
        hp.cg_fixed_iterations(A, b, max(args.warmup // 4, 2), record_history=False, fused=fused)   # warm-up
        _sync_barrier(job)
        t0 = time.perf_counter()
        x, hist = hp.cg_fixed_iterations(A, b, iters, record_history=True, fused=fused, graph=graph)
        _sync_barrier(job, closing=True)
        elapsed = time.perf_counter() - t0
        elapsed = job.max(elapsed)
        n_loc, nnz_loc = A.nrows_local, A.nnz
        b_spmv = wl.spmv_algorithmic_bytes(nnz_loc, n_loc, A.ncols_compressed, 4)
        b_iter = b_spmv + 96 * n_loc        # textbook unfused CG = SpMV + 96 n bytes
        ms_iter = elapsed / iters * 1e3
        out = {
            "metric": "CG ms/iteration, 3-D 7-pt Poisson, fp64", "value": round(ms_iter, 4), "unit": "ms/iter",
            "n_gpus": world, "steps": iters, "warmup": args.warmup, "ms_per_step": round(ms_iter, 4),
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"poisson3d 7-pt {N}x{N}x{planes} per GPU ({N}x{N}x{nz} global), {iters} CG iterations, "
                                   f"{'fused: SpMV+p.Ap, r-update+r.r, x/p-update' if fused else 'one kernel per reference operator'}"
                                   f"{', HIP graph replay' if graph else ''}",
                       "global_rows": n_glob, "nnz_per_gpu": nnz_loc},
            "roofline": {"bound": "hbm", "achieved": round(b_iter / (ms_iter * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(b_iter / (ms_iter * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_iteration": b_iter,
                         "moved_bytes_per_iteration": b_spmv + (64 if fused else 96) * n_loc,
                         "moved_gbs": round((b_spmv + (64 if fused else 96) * n_loc) / (ms_iter * 1e-3) / 1e9, 1),
                         "note": "whole iteration (SpMV, 2 reductions, the x / r / p updates), wall clock; bytes = the textbook unfused count of SURVEY 8d (SpMV + 96 n), the fused form moves SpMV + 64 n"},
            "residual_first": hist[0], "residual_last": hist[-1], "setup_s": round(setup_s, 2),
            "exchange_timed_out": bool(job.max(1.0 if hp.get_vector_plan(A, b).timed_out() else 0.0)),
        }
    elif args.workload == "poisson2d_spmm":
        # structured counterpart of config 5: the 5-point matrix times 16 dense columns.  Every B row is
        # needed by <= 5 matrix rows that sit close together, so the algorithmic byte count (each B row
        # once) is attainable -- this is the line that shows the SpMM kernel's own efficiency.
        k = 16
        N = args.size or 4096
        nx, ny_loc = N, N // 2
        ny = ny_loc * world
        lo, hi = rank * nx * ny_loc, (rank + 1) * nx * ny_loc
        t0 = time.perf_counter()
        A = device_stencil(hp, torch, backend, (nx, ny), lo, hi)
        b_rows = hi - lo
        Bl = torch.empty((b_rows, k), dtype=torch.float64, device=dev)
        hp._capi.call("hpcla_fill_uniform_f64", Bl.data_ptr(), lo * k, b_rows * k, wl.SEED_X,
                      torch.cuda.current_stream().cuda_stream)
        B = hp.HPCMatrix_local(Bl, backend)
        setup_s = time.perf_counter() - t0
        out = _measure_spmm(args, job, hp, wl, A, B, k, world, dev, setup_s,
                            "SpMM GFLOP/s (2*k*nnz/t), 2-D 5-pt Poisson, k=16, fp64",
                            f"poisson2d 5-pt {nx}x{ny_loc} slab per GPU, nnz/GPU={A.nnz}, k={k}, C = A*B")
    elif args.workload == "sprand_spmm":
        k = 16
        rows_loc = args.size or 2_097_152
        # HPCLA_SPMM_COLS_MULT=8 on ONE GPU reproduces config 5's per-GPU access pattern (B has
        # 8 x 2 097 152 rows = 2.1 GB, far beyond the 256 MiB Infinity Cache) without the exchange
        mult = int(os.environ.get("HPCLA_SPMM_COLS_MULT", "1"))
        ncols = rows_loc * world * mult
        mean_nnz = 29.8
        # generated ON THE DEVICE (torch is plumbing here): counts ~ Poisson(29.8) (= Binomial(ncols, 29.8/ncols)
        # to 1e-6), columns uniform, sorted within rows by one 64-bit key sort; then the library's device-side
        # column compression.  The numpy version of this setup (binomial + lexsort of 6e7 keys) took 17 s.
        t0 = time.perf_counter()
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xA11CE + rank)
        counts = torch.poisson(torch.full((rows_loc,), mean_nnz, dtype=torch.float64, device=dev), generator=gen).to(torch.int64)
        rowptr = torch.zeros(rows_loc + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=rowptr[1:])
        nnz = int(rowptr[-1].item())
        cols = torch.randint(0, ncols, (nnz,), generator=gen, device=dev, dtype=torch.int64)
        rowid = torch.repeat_interleave(torch.arange(rows_loc, device=dev, dtype=torch.int64), counts)
        key = torch.sort(rowid * ncols + cols).values
        cols = key - rowid * ncols
        del key, rowid, counts
        vals = torch.rand(nnz, generator=gen, device=dev, dtype=torch.float64)
        A = hp.HPCSparseMatrix_local_device(rowptr, cols, vals, ncols, backend, col_window=(0, ncols - 1))
        del cols
        if os.environ.get("HPCLA_SPRAND_SPMV", "") == "1":
            # the same unstructured matrix times ONE vector (single GPU only): the x gather is one
            # 64-byte sector per stored entry, far from the each-value-once algorithmic count
            xv = hp.HPCVector.zeros(np.array([0, ncols]), backend)
            hp._capi.call("hpcla_fill_uniform_f64", xv.v.data_ptr(), 0, ncols, wl.SEED_X,
                          torch.cuda.current_stream().cuda_stream)
            yv = A @ xv
            for _ in range(args.warmup):
                hp.mul_(yv, A, xv)
            _sync_barrier(job)
            steps = min(args.steps, 50)
            t0 = time.perf_counter()
            for _ in range(steps):
                hp.mul_(yv, A, xv)
            _sync_barrier(job)
            ms = (time.perf_counter() - t0) / steps * 1e3
            b_alg = wl.spmv_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, 4)
            b_sect = A.nnz * (12 + 64) + 12 * A.nrows_local
            line = json.dumps({"metric": "SpMV GFLOP/s, sprand ~29.8 nnz/row", "value": round(2.0 * A.nnz / (ms * 1e-3) / 1e9, 1),
                              "ms_per_step": round(ms, 4), "nnz": A.nnz, "ncols_compressed": A.ncols_compressed,
                              "algorithmic_gbs": round(b_alg / (ms * 1e-3) / 1e9, 1),
                              "frac_of_peak_algorithmic": round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "sector_gather_gbs": round(b_sect / (ms * 1e-3) / 1e9, 1),
                              "note": "sector_gather = 12 B + one 64-byte sector of x per stored entry"})
            hp.clear_plan_cache()
            return json.loads(line)
        b_rows = rows_loc * mult
        Bl = torch.empty((b_rows, k), dtype=torch.float64, device=dev)
        hp._capi.call("hpcla_fill_uniform_f64", Bl.data_ptr(), rank * b_rows * k, b_rows * k, wl.SEED_X,
                      torch.cuda.current_stream().cuda_stream)
        B = hp.HPCMatrix_local(Bl, backend)
        setup_s = time.perf_counter() - t0
        out = _measure_spmm(args, job, hp, wl, A, B, k, world, dev, setup_s,
                            "SpMM GFLOP/s (2*k*nnz/t), sprand ~29.8 nnz/row, k=16, fp64",
                            f"sprand-like {rows_loc} rows per GPU x {ncols} cols, nnz/GPU={A.nnz}, k={k}, C = A*B")
    job.barrier()
    hp.clear_spmm_cache()
    hp.clear_plan_cache()
    torch.cuda.empty_cache()
    return out
