#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: distributed CSR SpMV, 2-D 5-point Poisson, fp64.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload poisson2d] [--size 4096]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one y = A*x (``mul!(y, A, x)``) over the resident matrix: halo exchange (N > 1: RCCL
send/recv group, ahead of one fused launch by default or overlapped with the interior row blocks on a
side stream with HPCLA_HALO_MODE=overlap) + SpMV.  Workload at N = 1: BASELINE
configs[1], the 4096^2 Poisson matrix (n = 16 777 216, nnz = 83 869 696, Int32 indices).  N > 1:
weak scaling -- every GPU owns one 4096 x 4096 slab of a 4096 x (4096*N) grid (same per-GPU work as
N = 1, two 32 KiB halo lines per interior GPU).

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM, algorithmic
bytes / per-launch time measured with HIP events on the launch stream) and `cpu_baseline` (the
oracle's restatement of the reference kernel, timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (set before the HIP runtime initialises)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="poisson2d", choices=["poisson2d", "poisson2d_strong", "poisson3d_cg", "sprand_spmm", "poisson2d_spmm"])
    ap.add_argument("--size", type=int, default=0, help="grid edge N (default: 4096 for poisson2d)")
    ap.add_argument("--index", default="i32", choices=["i32", "i64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-packed", action="store_true", help="skip the extra opt-in packed-copy measurement")
    ap.add_argument("--host-setup", action="store_true", help="generate/compress the matrix on the host (numpy) instead of on the device")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def _device_barrier(torch, dist):
    """Barrier through the device collective path (a 1-element all-reduce on the NCCL/RCCL group):
    unambiguous for mixed gloo+nccl process groups."""
    t = torch.zeros(1, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()


def usable_cores() -> int:
    """Host threads this process may actually run: the affinity mask capped by the cgroup CPU quota
    (the GPU box exposes every host core in the mask but grants a 16-core share per GPU), or
    HPCLA_CPU_THREADS if set."""
    if os.environ.get("HPCLA_CPU_THREADS"):
        return max(1, int(os.environ["HPCLA_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return min(n, 64)


def cpu_baseline_spmv(rowptr, colval, vals, x_gathered, budget_s):
    """cpu_baseline leg: the oracle's restatement of _spmv_kernel! (src/sparse.jl:2055-2066), OpenMP
    static over rows like the reference's KernelAbstractions CPU backend, on all cores this
    process may use, then on 1 core; bounded to ~budget_s seconds."""
    from oracle import oracle as orc
    cores = usable_cores()
    nnz = len(vals)
    # (run BEFORE the OpenMP legs: libgomp's idle threads spin and would slow these single-thread timings)
    # independent cross-check of values and time: scipy's single-thread csr_matvec has the
    # same row-sequential order, so the results must agree bit for bit
    import scipy.sparse as sp
    n_rows = len(rowptr) - 1
    As = sp.csr_matrix((vals, colval, rowptr), shape=(n_rows, len(x_gathered)))
    ts_sp = []
    for _ in range(3):
        t0 = time.perf_counter()
        y_sp = As @ x_gathered
        ts_sp.append(time.perf_counter() - t0)
    # the reference's per-call staging on a GPU backend (execute_plan!, src/vectors.jl:394-463):
    # x to the host and the gathered vector back, 2 * n * 8 B of host copies, emulated with memcpy
    stage_src, stage_dst = np.ones(n_rows), np.empty(n_rows)
    t0 = time.perf_counter()
    for _ in range(5):
        np.copyto(stage_dst, stage_src)
        np.copyto(stage_src, stage_dst)
    staging_ms = (time.perf_counter() - t0) / 5 * 1e3
    out = {}
    for label, nt, share in (("all", cores, 0.6), ("one", 1, 0.4)):
        orc.lib().orc_set_threads(nt)
        orc.spmv(rowptr, colval, vals, x_gathered, nthreads=nt)          # warm-up
        ts = []
        t_end = time.perf_counter() + budget_s * share
        while time.perf_counter() < t_end or len(ts) < 3:
            t0 = time.perf_counter()
            orc.spmv(rowptr, colval, vals, x_gathered, nthreads=nt)
            ts.append(time.perf_counter() - t0)
            if len(ts) >= 200:
                break
        out[label] = (2.0 * nnz / np.median(ts) / 1e9, len(ts), float(np.median(ts)))
    orc.lib().orc_set_threads(cores)
    scipy_same = bool(np.array_equal(y_sp, orc.spmv(rowptr, colval, vals, x_gathered, nthreads=cores)))
    return {
        "scipy_1thread_gflops": round(2.0 * nnz / float(np.median(ts_sp)) / 1e9, 3),
        "scipy_bits_equal_oracle": scipy_same,
        "staging_emulation_ms": round(staging_ms, 3),
        "reference_like_end_to_end_ms": round(out["all"][2] * 1e3 + staging_ms, 3),
        "value": round(out["all"][0], 3), "unit": "GFLOP/s", "cores": cores, "kind": "port",
        "sample": (f"same matrix and x as the GPU run (rank 0 slab), {out['all'][1]} SpMVs on {cores} threads "
                   f"(median {out['all'][2]*1e3:.2f} ms) + {out['one'][1]} on 1 thread"),
        "value_1core": round(out["one"][0], 3), "ms_per_spmv": round(out["all"][2] * 1e3, 3),
    }


def step_breakdown(hp, torch, dist, world, plan, A, x, y, barrier):
    """Device time per call (HIP events, max over ranks) of the pieces of one distributed step."""
    import ctypes
    capi = hp._capi
    sp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    cur = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gp, gn = ctypes.c_void_p(), ctypes.c_int64(0)
    if plan.has_halo:
        capi.call("hpcla_halo_ghost_ptr", plan.halo, ctypes.byref(gp), ctypes.byref(gn))

    def split(blocks, nb):
        capi.call("hpcla_spmv_split_f64_i32", sp(A.rowptr_target), sp(plan.colval_split), sp(A.nzval), sp(x.v),
                  gp if gp.value else None, plan.n_own, sp(y.v), A.nrows_local, A.nnz, 0, sp(blocks), nb, cur())

    def exchange():
        # the step's own exchange and nothing else: the fused entry point over ZERO rows (same stream, same
        # RCCL group, same HPCLA_HALO_MODE as the timed loop -- no communication pattern the loop did not use)
        capi.call("hpcla_spmv_dist_f64_i32", plan.halo, sp(A.rowptr_target), sp(plan.colval_split), sp(A.nzval),
                  sp(x.v), plan.n_own, sp(y.v), 0, 0, 0, None, 0, None, 0, cur())

    legs = [("all_row_blocks_no_exchange", lambda: split(None, 0))]
    if plan.has_halo:
        legs += [("interior_blocks", lambda: split(plan.interior, plan.n_interior)),
                 ("boundary_blocks", lambda: split(plan.boundary, plan.n_boundary)),
                 ("exchange_only", exchange)]
    legs.append(("full_step", lambda: hp.mul_(y, A, x)))
    out = {"n_interior_blocks": plan.n_interior, "n_boundary_blocks": plan.n_boundary, "ghost_values": int(gn.value)}
    for name, fn in legs:
        fn()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        barrier()
        t = torch.tensor([e0.elapsed_time(e1) / 50], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[name + "_ms"] = round(float(t.item()), 5)
    return out


class _StdoutToStderr:
    """While active, file descriptor 1 points at stderr: native libraries (RCCL prints a version banner
    to stdout when a communicator is created) cannot add lines to the ONE JSON line this script owes."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    with _StdoutToStderr():
        out = _run()
    line, verified = out if isinstance(out, tuple) else (out, True)
    if line is not None:
        print(line, flush=True)            # the result is out before any teardown can go wrong
    with _StdoutToStderr():
        _teardown()
    if not verified:
        raise SystemExit(1)


def _teardown():
    """Collective teardown after the result line is printed: plans, then the process group.  The library's
    RCCL communicator is left to process exit on purpose (like the reference's NCCL communicator,
    ext/HPCLinearAlgebraCUDAExt.jl:373, 384-386: no destroy that could block behind a straggler)."""
    try:
        import torch.distributed as dist
        import hpcla_amd as hp
        hp.clear_plan_cache()
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception as exc:                 # teardown problems must not turn a finished run into a failure
        sys.stderr.write(f"bench: teardown: {type(exc).__name__}: {exc}\n")


def _run():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())

    import hpcla_amd as hp
    from hpcla_amd import workloads as wl

    Ti = np.int32 if args.index == "i32" else np.int64
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("cpu:gloo,cuda:nccl")
        backend = hp.backend_rocm_mpi(np.float64, Ti)
    else:
        backend = hp.backend_rocm_serial(np.float64, Ti, device_index=torch.cuda.current_device())

    if args.workload not in ("poisson2d", "poisson2d_strong"):
        from benchmarks import extra_workloads          # configs 4/5: separate harness
        return extra_workloads.run(args, backend, rank, world)       # its JSON line (rank 0) or None

    # ---- build the workload ------------------------------------------------------------------------
    strong = args.workload == "poisson2d_strong"      # BASELINE configs[2]: fixed 8192^2 grid over N GPUs
    N = args.size or (8192 if strong else 4096)
    if strong:
        nx, ny = N, N
        ny_loc = (N + world - 1) // world
        part0 = hp.uniform_partition(nx * ny, world)
        lo, hi = int(part0[rank]), int(part0[rank + 1])
    else:
        nx, ny_loc = N, N
        ny = ny_loc * world
        lo, hi = rank * nx * ny_loc, (rank + 1) * nx * ny_loc
    n_glob = nx * ny
    t0 = time.perf_counter()
    if args.host_setup:
        rowptr, colidx, vals = wl.poisson2d_rows(nx, ny, lo, hi)
        A = hp.HPCSparseMatrix_local(rowptr, colidx, vals, n_glob, backend)
        del colidx
    else:
        # device-side construction: generator kernel + bitmap/scan column compression
        s0 = torch.cuda.current_stream().cuda_stream
        nnz_gen = hp._capi.load().hpcla_poisson2d_nnz(nx, ny, lo, hi)
        rp_d = torch.empty(hi - lo + 1, dtype=torch.int64, device="cuda")
        ci_d = torch.empty(nnz_gen, dtype=torch.int64, device="cuda")
        va_d = torch.empty(nnz_gen, dtype=torch.float64, device="cuda")
        hp._capi.call("hpcla_gen_poisson2d", nx, ny, lo, hi, rp_d.data_ptr(), ci_d.data_ptr(), va_d.data_ptr(), s0)
        A = hp.HPCSparseMatrix_local_device(rp_d, ci_d, va_d, n_glob, backend,
                                            col_window=(max(lo - nx, 0), min(hi + nx, n_glob) - 1))
        del ci_d, rp_d
        vals = None
    part = A.row_partition
    x = hp.HPCVector.zeros(part, backend)
    hp._capi.call("hpcla_fill_uniform_f64", x.v.data_ptr(), lo, hi - lo, wl.SEED_X,
                  torch.cuda.current_stream().cuda_stream)
    y = hp.HPCVector.zeros(part, backend)
    plan = hp.get_vector_plan(A, x)
    setup_s = time.perf_counter() - t0
    nnz_loc, nrows_loc = A.nnz, A.nrows_local
    b_alg_loc = wl.spmv_algorithmic_bytes(nnz_loc, nrows_loc, A.ncols_compressed, np.dtype(Ti).itemsize)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            _device_barrier(torch, dist)
        torch.cuda.synchronize()

    # ---- verification of this rank's result on sampled rows (incl. halo-adjacent rows) -----------------
    hp.mul_(y, A, x)
    torch.cuda.synchronize()
    samp = np.unique(np.concatenate([np.arange(0, min(3 * nx, nrows_loc)),
                                     np.arange(max(0, nrows_loc - 3 * nx), nrows_loc),
                                     np.random.default_rng(rank).integers(0, nrows_loc, 4096)]))
    g = samp + lo
    gi, gj = g % nx, g // nx
    xs = lambda idx: wl.u01(wl.SEED_X, idx)
    want = np.zeros(len(g))
    # same order and rounding as the kernel: ascending column, mul then add
    for col, ok, coef in ((g - nx, gj > 0, -1.0), (g - 1, gi > 0, -1.0), (g, np.ones_like(g, bool), 4.0),
                          (g + 1, gi < nx - 1, -1.0), (g + nx, gj < ny - 1, -1.0)):
        term = coef * xs(np.where(ok, col, 0))
        want = np.where(ok, want + term, want)
    got = y.v[torch.from_numpy(samp).cuda()].cpu().numpy()
    verified = bool(np.array_equal(got, want))
    if not verified:                     # say where, for whoever reads the log of a failed multi-GPU run
        bad = np.flatnonzero(got != want)
        sys.stderr.write(f"bench: rank {rank}: {len(bad)} of {len(g)} sampled rows differ; first global rows "
                         f"{g[bad[:8]].tolist()} got {got[bad[:8]].tolist()} want {want[bad[:8]].tolist()}\n")
    if world > 1:
        flag = torch.tensor([1 if verified else 0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())

    # ---- warm-up, then EXACTLY K timed steps -----------------------------------------------------------
    for _ in range(args.warmup):
        hp.mul_(y, A, x)
    ev_t0, ev_t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev_t0.record()                       # HIP events on the launch stream, bracketing the timed region itself
    for _ in range(args.steps):
        hp.mul_(y, A, x)
    ev_t1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    timed_region_launch_ms = ev_t0.elapsed_time(ev_t1) / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-launch duration with HIP events on the launch stream (roofline.achieved) ----------------
    reps = min(max(args.steps, 20), 200)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    barrier()
    for a, b in evs:
        a.record()
        hp.mul_(y, A, x)
        b.record()
    torch.cuda.synchronize()
    per_launch_ms = np.array([a.elapsed_time(b) for a, b in evs])
    launch_ms = float(np.mean(per_launch_ms))
    # back-to-back launches between ONE event pair (no per-launch event overhead)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hp.mul_(y, A, x)
    b.record()
    torch.cuda.synchronize()
    stream_ms = a.elapsed_time(b) / reps

    # ---- N > 1: where a distributed step spends its time (device time per call, HIP events; every rank makes
    #      the same calls in the same order, and the only communication is the step's own halo exchange) ----
    breakdown = None
    selftest = os.environ.get("HPCLA_BENCH_BREAKDOWN_SELFTEST", "") == "1"      # runs the local legs at N = 1
    if ((world > 1 and plan.has_halo) or selftest) and not plan.is_i64:
        try:
            breakdown = step_breakdown(hp, torch, dist, world, plan, A, x, y, barrier)
        except Exception as exc:        # identical code and call order on every rank: all ranks land here together
            breakdown = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- opt-in packed copy (3 B per stored entry instead of 12; same bits), reported separately --------
    packed = None
    if not args.no_packed and world == 1:        # an optional extra must not stand between an N > 1 run and its result line
        ok = A.enable_packed(x)
        if world > 1:
            f = torch.tensor([1 if ok else 0], device="cuda")
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            ok = bool(f.item())
        if ok:
            hp.mul_(y, A, x)
            torch.cuda.synchronize()
            same = bool(np.array_equal(y.v[torch.from_numpy(samp).cuda()].cpu().numpy(), want))
            for _ in range(args.warmup):
                hp.mul_(y, A, x)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                hp.mul_(y, A, x)
            barrier()
            el = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            pk_bytes = 3 * nnz_loc + 4 * (nrows_loc + 1) + 8 * nrows_loc + 8 * A.ncols_compressed
            packed = {"ms_per_step": round(el / args.steps * 1e3, 5),
                      "speedup_vs_csr": round((elapsed / args.steps) / (el / args.steps), 3),
                      "bytes_moved_per_launch": pk_bytes,
                      "moved_gbs": round(pk_bytes / (el / args.steps) / 1e9, 1),
                      "csr_algorithmic_gbs": round(b_alg_loc / (el / args.steps) / 1e9, 1),
                      "verified_same_bits": same,
                      "note": "16-bit block-relative columns + 8-bit value codes for interior row blocks; "
                              "NOT the CSR headline: fewer bytes are moved, results bit-identical"}
        A.disable_packed()

    nnz_tot = nnz_loc * world            # slabs differ by <= 2*nx nonzeros; rank 0 reports its own * N
    if world > 1:
        t = torch.tensor([float(nnz_loc), float(b_alg_loc)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        nnz_tot, b_alg_tot = int(t[0].item()), int(t[1].item())
    else:
        b_alg_tot = b_alg_loc
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * nnz_tot / (elapsed / args.steps) / 1e9
    # roofline.achieved: algorithmic bytes of one launch / average launch duration over the TIMED region
    # (device time between the two events / K; the per-launch event pairs below it are a cross-check)
    achieved = b_alg_loc / (timed_region_launch_ms * 1e-3) / 1e9

    traffic = None
    tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tj) and args.index == "i32" and N == 4096 and not strong:     # measured for that launch only
        try:
            traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "SpMV GFLOP/s (2*nnz/t), 2-D 5-pt Poisson, fp64",
        "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"poisson2d 5-pt, {nx}x{ny_loc} slab per GPU ({nx}x{ny} global), "
                               f"n={n_glob}, nnz={nnz_tot}, index={args.index}, CSR SpMV y=A*x",
                   "global_rows": n_glob, "nnz": nnz_tot, "index_type": args.index,
                   "parallelism": f"row-slab x{world}, RCCL halo" if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": "hpcla::spmv_rowblock_quad_kernel", "algorithmic_bytes_per_launch": b_alg_loc,
                     "launch_ms_timed_region": round(timed_region_launch_ms, 5),
                     "launch_ms_event_pairs": round(launch_ms, 5), "launch_ms_back_to_back": round(stream_ms, 5),
                     "launch_ms_min": round(float(per_launch_ms.min()), 5),
                     "launch_ms_median": round(float(np.median(per_launch_ms)), 5)},
        "hbm_gbs_whole_job": round(b_alg_tot / (elapsed / args.steps) / 1e9, 1),
        "hbm_frac_of_peak_whole_job": round(b_alg_tot / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
        "verified_vs_closed_form": verified, "setup_s": round(setup_s, 2),
        "packed_csr_opt_in": packed,
    }
    if breakdown is not None:
        result["step_breakdown_ms_max_over_ranks"] = breakdown
    if rank == 0 and not args.no_cpu_baseline:
        xg = x.local_values()
        ghost = np.zeros(0)
        if plan.has_halo:
            ghost_idx = A.col_indices[A.col_indices >= hi]
            ghost_lo = A.col_indices[A.col_indices < lo]
            xfull = np.concatenate([wl.u01(wl.SEED_X, ghost_lo), xg, wl.u01(wl.SEED_X, ghost_idx)])
        else:
            xfull = xg
        if vals is None:
            vals = A.nzval.cpu().numpy()
        result["cpu_baseline"] = cpu_baseline_spmv(A.rowptr, A.colval, vals, xfull, args.cpu_seconds)
    if world > 1:
        _device_barrier(torch, dist)
    if not verified:
        sys.stderr.write("bench: result verification FAILED\n")
    return (json.dumps(result) if rank == 0 else None), verified


if __name__ == "__main__":
    main()
