#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: distributed CSR SpMV, 2-D 5-point Poisson, fp64.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload poisson2d] [--size 4096]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` starts its own N ranks (one process per GPU; the parent spawns them before it
touches the GPU runtime); under torch.distributed.run it uses the ranks it is given.

A "step" is one y = A*x (``mul!(y, A, x)``) over the resident matrix: halo exchange + SpMV.  At N > 1 the
default exchange is the peer-window push over xGMI (a small kernel stores the boundary values straight
into the neighbours' ghost windows; the SpMV's own boundary workgroups wait for them -- DESIGN.md
section 4); HPCLA_HALO_MODE=serial|overlap selects the RCCL send/recv orderings instead, and the line's
`step_breakdown_ms_max_over_ranks` times every ordering.  Workload at N = 1: BASELINE configs[1], the 4096^2
Poisson matrix (n = 16 777 216, nnz = 83 869 696, Int32 indices).  N > 1: weak scaling -- every GPU owns one
4096 x 4096 slab of a 4096 x (4096*N) grid (same per-GPU work as N = 1, two 32 KiB halo lines per interior
GPU); the sub-record `strong_scaling` is BASELINE configs[2], the fixed 8192^2 problem over the N GPUs, with
the same problem on rank 0 alone beside it (speed-up measured inside one run).

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
    ap.add_argument("--strong-size", type=int, default=8192, help="grid edge of the strong-scaling sub-record (BASELINE configs[2])")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling sub-record")
    ap.add_argument("--no-extras", action="store_true", help="skip the CG / SpMM sub-records (BASELINE configs[3], [4])")
    return ap.parse_args()


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
    sig = lambda v: float(f"{v:.6g}")       # 6 SIGNIFICANT digits: a small matrix on a slow host must not round to 0.0
    return {
        "scipy_1thread_gflops": sig(2.0 * nnz / float(np.median(ts_sp)) / 1e9),
        "scipy_bits_equal_oracle": scipy_same,
        "staging_emulation_ms": round(staging_ms, 3),
        "reference_like_end_to_end_ms": round(out["all"][2] * 1e3 + staging_ms, 3),
        "value": sig(out["all"][0]), "unit": "GFLOP/s", "cores": cores, "kind": "port",
        "sample": (f"same matrix and x as the GPU run (rank 0 slab), {out['all'][1]} SpMVs on {cores} threads "
                   f"(median {out['all'][2]*1e3:.2f} ms) + {out['one'][1]} on 1 thread"),
        "value_1core": sig(out["one"][0]), "ms_per_spmv": sig(out["all"][2] * 1e3),
    }


class Job:
    """Host-side collectives of the benchmark itself (barrier, max / sum / min over ranks): gloo on CPU
    scalars, so they work whatever the data path is -- RCCL, or peer windows with ranks sharing a GPU."""

    def __init__(self, torch, dist, world, rank):
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        self._capi = self._comm = self._token = None

    def attach_data_path(self, hp, backend):
        """From now on a barrier ends with a rendezvous ON THE DEVICES: one scalar all-reduce through the
        library's own communicator (peer windows or RCCL).  No rank's GPU leaves that kernel before every rank's
        has entered it, so the ranks leave the barrier within microseconds of each other -- a gloo barrier is a
        TCP round trip per rank (hundreds of microseconds at 8 ranks), which is not small next to 20 steps of
        0.24 ms."""
        if self.world > 1:
            self._capi, self._comm = hp._capi, backend.rccl
            self._token = self.torch.zeros(1, dtype=self.torch.float64, device="cuda")

    def barrier(self, device_only=False):
        """cuda synchronize, barrier over the ranks, cuda synchronize.  device_only=True (the closing bracket of a
        timed region): the device rendezvous alone -- still a barrier over all ranks, without gloo's latency
        inside the timed window."""
        import ctypes
        torch = self.torch
        torch.cuda.synchronize()
        if self.world > 1:
            if not (device_only and self._comm is not None):
                self.dist.barrier()
            if self._comm is not None:
                self._capi.call("hpcla_allreduce_f64", self._comm, ctypes.c_void_p(self._token.data_ptr()), 1, 0,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()

    def _reduce(self, value, op):
        if self.world == 1:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, v):
        return self._reduce(v, self.dist.ReduceOp.MAX if self.world > 1 else None)

    def min(self, v):
        return self._reduce(v, self.dist.ReduceOp.MIN if self.world > 1 else None)

    def sum(self, v):
        return self._reduce(v, self.dist.ReduceOp.SUM if self.world > 1 else None)

    def agree(self, flag) -> bool:
        """Collective yes/no (every rank must say yes): budget decisions are taken by all ranks together, so no
        rank starts a collective stage another rank has skipped."""
        return bool(self.min(1.0 if flag else 0.0))


HALO_MODES = {"serial": 0, "overlap": 1, "push": 2}


def step_breakdown(hp, job, backend, plan, A, x, y):
    """Device time per call (HIP events, max over ranks) of one distributed step in EVERY ordering this job
    can run (push: peer windows attached; serial / overlap: an RCCL communicator exists), plus the pieces of
    the step.  Every rank makes the same calls in the same order; the only communication is the step's own
    halo exchange."""
    import ctypes
    torch = job.torch
    capi = hp._capi
    sp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    cur = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sfx = "i64" if plan.is_i64 else "i32"

    _g = []

    def ghost():                         # fetched ONCE: for a double-buffered plan the call reads the device step counter
        if not _g:
            gp, gn = ctypes.c_void_p(), ctypes.c_int64(0)
            if plan.has_halo:
                capi.call("hpcla_halo_ghost_ptr", plan.halo, ctypes.byref(gp), ctypes.byref(gn))
            _g.append(((gp if gp.value else None), int(gn.value)))
        return _g[0]

    def split(blocks, nb):
        capi.call(f"hpcla_spmv_split_f64_{sfx}", sp(plan.rowptr_of(A)), sp(plan.colval_split), sp(A.nzval), sp(x.v),
                  ghost()[0], plan.n_own, sp(y.v), A.nrows_local, A.nnz, 0, sp(blocks), nb, cur())

    def exchange():
        # the step's own exchange and nothing else: the fused entry point over ZERO rows (same stream, same
        # transport and ordering as the timed loop -- no communication pattern the loop did not use)
        capi.call(f"hpcla_spmv_dist_f64_{sfx}", plan.halo, sp(plan.rowptr_of(A)), sp(plan.colval_split), sp(A.nzval),
                  sp(x.v), plan.n_own, sp(y.v), 0, 0, 0, None, 0, None, 0, cur())

    def timed(fn, reps=50):
        fn()
        job.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        job.barrier()
        return round(job.max(e0.elapsed_time(e1) / reps), 5)

    out = {"n_interior_blocks": plan.n_interior, "n_boundary_blocks": plan.n_boundary, "ghost_values": ghost()[1]}
    out["all_row_blocks_no_exchange_ms"] = timed(lambda: split(None, 0))
    if plan.has_halo:
        out["interior_blocks_ms"] = timed(lambda: split(plan.interior, plan.n_interior))
        out["boundary_blocks_ms"] = timed(lambda: split(plan.boundary, plan.n_boundary))
    modes = []
    if getattr(plan, "push", False):
        modes.append("push")
    if backend.has_rccl:
        modes += ["serial", "overlap"]
    per_mode = {}
    for m in modes:
        capi.call("hpcla_set_halo_mode", HALO_MODES[m])
        per_mode[m] = {"full_step_ms": timed(lambda: hp.mul_(y, A, x)),
                       "exchange_only_ms": timed(exchange) if plan.has_halo else None}
    capi.call("hpcla_set_halo_mode", -1)
    out["modes"] = per_mode
    out["timed_out"] = bool(job.max(1.0 if plan.timed_out() else 0.0))
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


class Outcome:
    """What the headline run hands to main(): rank 0's result record (None on the other ranks), the verdict of
    the closed-form checks, and -- at N > 1 -- the optional transport comparison still to be run."""

    def __init__(self, result, verified, breakdown_fn):
        self.result, self.verified, self.breakdown_fn = result, verified, breakdown_fn


def _guarded_breakdown(fn, result, verified, stdout_fd, limit):
    """Run the transport comparison under a watchdog.  It exercises orderings the timed loop did not use (the
    RCCL send/recv modes next to the peer-window push); if it does not come back within `limit` seconds (what is
    left of the run's budget, at most 90 s -- benchmarks/budget.py; HPCLA_BENCH_BREAKDOWN_TIMEOUT_S overrides)
    every rank gives up on its own timer: rank 0 prints the finished result line without the comparison and the
    process exits -- an optional diagnostic must not be able to cost the measurement."""
    import threading
    lock, state = threading.Lock(), {"over": False}

    def bail():
        with lock:
            if state["over"]:
                return
            state["over"] = True
        sys.stderr.write(f"bench: the transport comparison did not finish within {limit:.0f} s; result printed without it\n")
        if result is not None:
            result["step_breakdown_ms_max_over_ranks"] = {"error": f"did not finish within {limit:.0f} s; omitted"}
            os.write(stdout_fd, (json.dumps(result) + "\n").encode())
        os._exit(0 if verified else 1)

    timer = threading.Timer(limit, bail)
    timer.daemon = True
    timer.start()
    try:
        bd = fn()
    except Exception as exc:             # identical code and call order on every rank: all ranks land here together
        bd = {"error": f"{type(exc).__name__}: {exc}"}
    with lock:
        late = state["over"]
        state["over"] = True
    if late:                             # the watchdog is already printing and exiting: leave it to it
        time.sleep(3600)
    timer.cancel()
    return bd


def _load_budget():
    """benchmarks/budget.py loaded BY PATH (stdlib only; the launching parent imports nothing else)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hpcla_budget", os.path.join(ROOT, "benchmarks", "budget.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _load_launcher():
    """linearalgebrampi.jl_amd/launch.py loaded BY PATH: importing the package would import torch, and the
    launching parent must never touch the GPU runtime."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hpcla_launch", os.path.join(ROOT, "linearalgebrampi.jl_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    args = parse()
    launch = _load_launcher()
    budget = _load_budget().Budget()                 # origin: HPCLA_BENCH_T0 if a parent set it, else now
    budget.export_guards()                           # the ranks inherit the origin and the derived guard defaults
    if args.gpus > 1 and not launch.already_launched():
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (the reference's
        # distributed entry does the same with mpiexec, test/runtests.jl:16-35).  Nothing GPU-related has
        # been imported in this process.
        sys.stderr.write(f"bench: launching {args.gpus} ranks (one process per GPU)\n")
        # the launcher's own limit is derived from the driver's (outer - 30 s), never larger than it
        raise SystemExit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                            timeout=float(os.environ.get("HPCLA_BENCH_TIMEOUT_S", budget.launcher_timeout()))))
    with _StdoutToStderr() as redirected:
        out = _run(args, budget)
        if isinstance(out, Outcome):
            if out.breakdown_fn is not None:
                limit = float(os.environ.get("HPCLA_BENCH_BREAKDOWN_TIMEOUT_S", budget.comparison_timeout()))
                budget.stage(f"transport comparison (watchdog {limit:.0f} s)")
                bd = _guarded_breakdown(out.breakdown_fn, out.result, out.verified, redirected.saved, limit)
                if out.result is not None:
                    out.result["step_breakdown_ms_max_over_ranks"] = bd
                budget.stage("transport comparison done")
            out = (json.dumps(out.result) if out.result is not None else None), out.verified
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
        hp.clear_spmm_cache()
        hp.clear_plan_cache()
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    except Exception as exc:                 # teardown problems must not turn a finished run into a failure
        sys.stderr.write(f"bench: teardown: {type(exc).__name__}: {exc}\n")


def closed_form_check(wl, torch, y, nx, ny, lo, nrows_loc, rank):
    """This rank's result on sampled rows (incl. the halo-adjacent first / last grid lines) against the
    closed form of the 5-point matrix: same order and rounding as the kernel (ascending column, multiply
    then add), so the comparison is bit-exact."""
    samp = np.unique(np.concatenate([np.arange(0, min(3 * nx, nrows_loc)),
                                     np.arange(max(0, nrows_loc - 3 * nx), nrows_loc),
                                     np.random.default_rng(rank).integers(0, nrows_loc, 4096)]))
    g = samp + lo
    gi, gj = g % nx, g // nx
    xs = lambda idx: wl.u01(wl.SEED_X, idx)
    want = np.zeros(len(g))
    for col, ok, coef in ((g - nx, gj > 0, -1.0), (g - 1, gi > 0, -1.0), (g, np.ones_like(g, bool), 4.0),
                          (g + 1, gi < nx - 1, -1.0), (g + nx, gj < ny - 1, -1.0)):
        term = coef * xs(np.where(ok, col, 0))
        want = np.where(ok, want + term, want)
    got = y.v[torch.from_numpy(samp).cuda()].cpu().numpy()
    ok = bool(np.array_equal(got, want))
    if not ok:                     # say where, for whoever reads the log of a failed multi-GPU run
        bad = np.flatnonzero(got != want)
        sys.stderr.write(f"bench: rank {rank}: {len(bad)} of {len(g)} sampled rows differ; first global rows "
                         f"{g[bad[:8]].tolist()} got {got[bad[:8]].tolist()} want {want[bad[:8]].tolist()}\n")
    return ok, samp, want


class PoissonRun:
    """One resident 2-D Poisson problem (matrix, x, y, plan) of `world` ranks and the timing of K steps."""

    def __init__(self, hp, wl, job, backend, args, N, strong, world, rank):
        torch = job.torch
        self.hp, self.wl, self.job, self.backend = hp, wl, job, backend
        Ti = np.int32 if args.index == "i32" else np.int64
        if strong:                                  # BASELINE configs[2]: ONE N x N grid over all ranks
            nx, ny = N, N
            part0 = hp.uniform_partition(nx * ny, world)
            lo, hi = int(part0[rank]), int(part0[rank + 1])
            ny_loc = (hi - lo) // nx
        else:                                       # weak: one N x N slab per rank
            nx, ny_loc = N, N
            ny = ny_loc * world
            lo, hi = rank * nx * ny_loc, (rank + 1) * nx * ny_loc
        self.nx, self.ny, self.ny_loc, self.lo, self.hi, self.n_glob = nx, ny, ny_loc, lo, hi, nx * ny
        t0 = time.perf_counter()
        self.vals = None
        if args.host_setup:
            rowptr, colidx, vals = wl.poisson2d_rows(nx, ny, lo, hi)
            self.A = hp.HPCSparseMatrix_local(rowptr, colidx, vals, self.n_glob, backend)
            self.vals = vals
            del colidx
        else:
            # device-side construction: generator kernel + bitmap/scan column compression
            s0 = torch.cuda.current_stream().cuda_stream
            nnz_gen = hp._capi.load().hpcla_poisson2d_nnz(nx, ny, lo, hi)
            rp_d = torch.empty(hi - lo + 1, dtype=torch.int64, device="cuda")
            ci_d = torch.empty(nnz_gen, dtype=torch.int64, device="cuda")
            va_d = torch.empty(nnz_gen, dtype=torch.float64, device="cuda")
            hp._capi.call("hpcla_gen_poisson2d", nx, ny, lo, hi, rp_d.data_ptr(), ci_d.data_ptr(), va_d.data_ptr(), s0)
            self.A = hp.HPCSparseMatrix_local_device(rp_d, ci_d, va_d, self.n_glob, backend,
                                                     col_window=(max(lo - nx, 0), min(hi + nx, self.n_glob) - 1))
            del ci_d, rp_d
        A = self.A
        self.x = hp.HPCVector.zeros(A.row_partition, backend)
        hp._capi.call("hpcla_fill_uniform_f64", self.x.v.data_ptr(), lo, hi - lo, wl.SEED_X,
                      torch.cuda.current_stream().cuda_stream)
        self.y = hp.HPCVector.zeros(A.row_partition, backend)
        self.plan = hp.get_vector_plan(A, self.x)
        torch.cuda.synchronize()
        self.setup_s = time.perf_counter() - t0
        self.nnz_loc, self.nrows_loc = A.nnz, A.nrows_local
        # bytes per index the KERNEL streams: the plan's (an Int64 matrix whose plan was narrowed streams Int32)
        self.index_bytes = 8 if self.plan.is_i64 else 4
        self.narrowed = bool(getattr(self.plan, "narrowed", False))
        self.b_alg_loc = wl.spmv_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, self.index_bytes)
        self.b_alg_matrix_index_type = wl.spmv_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, np.dtype(Ti).itemsize)
        hp.mul_(self.y, A, self.x)
        torch.cuda.synchronize()
        ok, self.samp, self.want = closed_form_check(wl, torch, self.y, nx, ny, lo, self.nrows_loc, rank)
        self.collective = world == job.world          # False: rank 0 alone inside a multi-rank job
        self.verified = bool(job.min(1.0 if ok else 0.0)) if self.collective else ok

    def time_steps(self, steps, warmup, settle=False):
        """warm-up, then EXACTLY `steps` timed steps bracketed by barrier + synchronize on both sides; returns
        (wall seconds, max over ranks; device ms per launch between two HIP events on the launch stream).
        `settle` (SUB-records only; the headline keeps the contract's W): warm up for at least
        benchmarks.extra_workloads.SETTLE_MS as well, like the other sub-records (clocks: extra_workloads.warm_up);
        the number of warm-up steps actually run is left in self.warmup_run."""
        torch, hp, job = self.job.torch, self.hp, self.job
        collective = self.collective
        sync = job.barrier if collective else torch.cuda.synchronize
        self.warmup_run = warmup
        if settle and collective:
            from benchmarks.extra_workloads import warm_up
            self.warmup_run = warm_up(job, lambda: hp.mul_(self.y, self.A, self.x), warmup)
        elif settle:                                 # rank 0 alone inside a multi-rank job: no collective decision
            from benchmarks.extra_workloads import SETTLE_MS
            t_w = time.perf_counter()
            n_w = 0
            while n_w < warmup or (time.perf_counter() - t_w) * 1e3 < SETTLE_MS:
                hp.mul_(self.y, self.A, self.x)
                n_w += 1
                if n_w % 16 == 0:
                    torch.cuda.synchronize()
            self.warmup_run = n_w
        for _ in range(0 if settle else warmup):
            hp.mul_(self.y, self.A, self.x)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync()
        t0 = time.perf_counter()
        ev0.record()                       # HIP events on the launch stream, bracketing the timed region itself
        for _ in range(steps):
            hp.mul_(self.y, self.A, self.x)
        ev1.record()
        if collective:
            job.barrier(device_only=True)
        else:
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if collective:
            elapsed = job.max(elapsed)
        return elapsed, ev0.elapsed_time(ev1) / steps

    def release(self):
        self.A = self.x = self.y = self.plan = None


def strong_scaling_record(hp, wl, job, backend, args, world, rank, budget=None, size=None, n1_key="strong_scaling_n1"):
    """BASELINE configs[2]: ONE 8192^2 Poisson problem row-partitioned over all ranks (strong scaling), and --
    on rank 0 alone, same box, same run -- the whole problem on one GPU, so the record carries its own
    speed-up.  The N = 1 job reports just the single-GPU point.  `size` = 4096: the second reading of north_star's
    "n ~ 10^8 nnz ... >= 6x at 8 GPUs" -- the headline matrix itself (83.9 M entries) strong-scaled."""
    torch = job.torch
    N = size or args.strong_size
    steps, warmup = min(args.steps, 50), min(args.warmup, 10)
    rec = {"workload": f"poisson2d 5-pt {N}x{N} global over {world} GPU(s), CSR SpMV y=A*x, index={args.index}"}
    run = PoissonRun(hp, wl, job, backend, args, N, True, world, rank)
    el, launch_ms = run.time_steps(steps, warmup, settle=True)
    nnz_tot = int(job.sum(run.nnz_loc))
    b_tot = job.sum(run.b_alg_loc)
    ms = el / steps * 1e3
    rec.update({"n_gpus": world, "steps": steps, "warmup": run.warmup_run, "ms_per_step": round(ms, 5),
                "gflops": round(2.0 * nnz_tot / (ms * 1e-3) / 1e9, 2), "nnz": nnz_tot,
                "hbm_frac_of_peak_whole_job": round(b_tot / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                "verified_vs_closed_form": run.verified, "timed_out": bool(job.max(1.0 if run.plan.timed_out() else 0.0)),
                "halo_mode": ("push" if getattr(run.plan, "push", False) else
                              os.environ.get("HPCLA_HALO_MODE", "serial") if run.plan.has_halo else "none (no neighbours)")})
    verified = run.verified
    run.release()
    hp.clear_plan_cache()
    torch.cuda.empty_cache()
    if world > 1 and budget is not None and not job.agree(budget.allows(n1_key)):
        rec["n1_ms_per_step_rank0_alone"] = rec["speedup_vs_n1"] = None
        rec["n1_skipped"] = "budget"
    elif world > 1:
        n1 = None
        if rank == 0:                                # the other ranks wait at the barrier below
            b1 = hp.backend_rocm_serial(np.float64, np.int32 if args.index == "i32" else np.int64,
                                        device_index=torch.cuda.current_device())
            one = PoissonRun(hp, wl, job, b1, args, N, True, 1, 0)
            el1, _ = one.time_steps(steps, warmup, settle=True)
            n1 = el1 / steps * 1e3
            verified = verified and one.verified
            one.release()
            torch.cuda.empty_cache()
        job.barrier()
        n1 = job.max(n1 if n1 is not None else 0.0)
        rec["n1_ms_per_step_rank0_alone"] = round(n1, 5)
        rec["speedup_vs_n1"] = round(n1 / ms, 3)
    return rec, verified


def spmv_kernel_name(hp=None):
    """the CSR SpMV kernel of the library's aligned launches (the row gather; the quad kernel of rounds 1-3 and its switch
    were retired in round 6)"""
    return "spmv_rowgather_kernel"


def spmv_kernel_instance(hp=None, is_i64=False, split=False, wait=False):
    """The launched instantiation as rocprofv3 prints it: <index type, SPLIT, WAIT, LONGR = false> (round 5 added the
    opt-in long-row flag)."""
    args = ["long" if is_i64 else "int", "true" if split else "false", "true" if wait else "false", "false"]
    return "hpcla::%s<%s>" % (spmv_kernel_name(), ", ".join(args))


def f32_kernel_name(kc, index="int", split=False):
    """Float32 lanes = rows kernel as rocprofv3 prints it: the element type is a template argument of the shared row-gather
    template since round 4 (csrc/rowgather_t.h: rowgather_kernel<T, I, SPLIT, KC, URX>), not a kernel of its own."""
    return "hpcla::rowgather_kernel<float, %s, %s, %d, 0>" % (index, "true" if split else "false", kc)


def headline_traffic(block_group, applicable=True):
    """HBM bytes per launch of the headline kernel (config 2's slab, Int32 kernel) from the builder's stored rocprofv3
    --pmc passes (profiles/traffic_latest.json), under the block order this run's plan chose.  The same per-GPU share
    at N > 1: one 4096^2 slab per rank is the single-rank pass's workload plus two ghost rows."""
    tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not applicable or not os.path.exists(tj):
        return None, "no stored PMC measurement for this shape (profiles/traffic_latest.json)"
    try:
        rec = json.load(open(tj))
        traffic = rec.get("hbm_bytes_per_launch")
        source = ("NOT measured by this run: PMC counters cannot be read from inside the process; value "
                  "stored by the builder's rocprofv3 passes of this same command -- " + rec.get("source", tj))
        # the counter depends on the block order the plan's measurement chose for this run (neighbouring XCD groups
        # share x lines; FETCH_SIZE counts them once per L2 although the Infinity Cache serves the repeats)
        per_order = rec.get("by_block_order", {}).get(str(block_group))
        if per_order:
            traffic = per_order["hbm_bytes_per_launch"]
            source += ("; the pass under this run's block order (" +
                       ("natural" if block_group <= 1 else f"XCD groups of {block_group} row blocks") + ")")
        return traffic, source
    except Exception as exc:
        return None, f"profiles/traffic_latest.json unreadable ({type(exc).__name__})"


def int64_record(hp, wl, job, args, N, steps, warmup):
    """The headline matrix with Int64 indices -- the reference's default `Ti = Int` (src/backends.jl:348, 369) -- as a
    driver-timed sub-record (N = 1).  Round 4: the plan NARROWS such a matrix (nnz and the split column space fit
    Int32: sparse.can_narrow_indices), so the launches take the Int32 kernel and stream 12 B per stored entry; the
    record says so (`narrowed`, `index_bytes_streamed`) and prices the launch by the bytes it streams, with the Int64
    count (1 744 568 328 B at 4096^2) beside it.  HPCLA_NARROW_INDICES=0 gives the Int64 kernel back."""
    import copy
    torch = job.torch
    a2 = copy.copy(args)
    a2.index = "i64"
    b64 = hp.backend_rocm_serial(np.float64, np.int64, device_index=torch.cuda.current_device())
    run = PoissonRun(hp, wl, job, b64, a2, N, False, 1, 0)
    el, launch_ms = run.time_steps(steps, warmup, settle=True)
    ms = el / steps * 1e3
    from benchmarks.extra_workloads import stored_traffic
    if run.narrowed:
        traffic, traffic_source = headline_traffic(run.plan.block_group, N == 4096)
        if traffic_source:
            traffic_source += "; this Int64 matrix runs the headline's Int32 kernel on a narrowed plan, so the headline's passes apply"
    else:
        traffic, traffic_source = stored_traffic("poisson2d_spmv_int64", N == 4096)
    rec = {"workload": f"poisson2d 5-pt {N}x{N}, CSR SpMV y=A*x, index=i64 (reference default Ti=Int)",
           "narrowed": run.narrowed, "index_bytes_streamed": run.index_bytes,
           "steps": steps, "warmup": run.warmup_run, "ms_per_step": round(ms, 5),
           "gflops": round(2.0 * run.nnz_loc / (ms * 1e-3) / 1e9, 2),
           "roofline": {"bound": "hbm", "achieved": round(run.b_alg_loc / (launch_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(run.b_alg_loc / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "traffic_source": traffic_source,
                        "kernel": spmv_kernel_instance(hp, run.plan.is_i64),
                        "block_order_group": run.plan.block_group,
                        "algorithmic_bytes_per_launch": run.b_alg_loc,
                        "algorithmic_bytes_if_int64_were_streamed": run.b_alg_matrix_index_type,
                        "launch_ms_timed_region": round(launch_ms, 5)},
           "verified_vs_closed_form": run.verified}
    ok = run.verified
    run.release()
    hp.clear_plan_cache()
    torch.cuda.empty_cache()
    return rec, ok


def float32_record(hp, wl, job, args, N, steps, warmup):
    """The headline matrix with Float32 values -- the reference's other GPU configuration (CUDA x {Float32, Float64},
    test/test_utils.jl:62-80) -- through the host layer's Float32 backend (csrc/f32.hip), as a driver-timed sub-record
    (N = 1).  Algorithmic bytes are SURVEY 8d's formula with sizeof(T) = 4: 8 B per stored entry, 4 + 4 B per row, 4 B per
    column.  Verified like the headline, bit for bit, against the closed form evaluated in Float32."""
    torch = job.torch
    b32 = hp.backend_rocm_serial(np.float32, np.int32, device_index=torch.cuda.current_device())
    nx = ny = N
    n = nx * ny
    s0 = torch.cuda.current_stream().cuda_stream
    nnz = hp._capi.load().hpcla_poisson2d_nnz(nx, ny, 0, n)
    rp_d = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ci_d = torch.empty(nnz, dtype=torch.int64, device="cuda")
    va_d = torch.empty(nnz, dtype=torch.float64, device="cuda")
    hp._capi.call("hpcla_gen_poisson2d", nx, ny, 0, n, rp_d.data_ptr(), ci_d.data_ptr(), va_d.data_ptr(), s0)
    A = hp.HPCSparseMatrix_local_device(rp_d, ci_d, va_d, n, b32, col_window=(0, n - 1))
    del rp_d, ci_d, va_d
    x64 = torch.empty(n, dtype=torch.float64, device="cuda")
    hp._capi.call("hpcla_fill_uniform_f64", x64.data_ptr(), 0, n, wl.SEED_X, s0)
    x = hp.HPCVector.zeros(A.row_partition, b32)
    x.v.copy_(x64)                                   # rounded to Float32 once, like an upload of Float32 data
    del x64
    y = hp.HPCVector.zeros(A.row_partition, b32)
    step = lambda: hp.mul_(y, A, x)
    step()
    torch.cuda.synchronize()
    # closed form in Float32: same order and rounding as the kernel (ascending column, multiply then add, all in float)
    samp = np.unique(np.concatenate([np.arange(0, 3 * nx), np.arange(n - 3 * nx, n), np.random.default_rng(0).integers(0, n, 4096)]))
    gi, gj = samp % nx, samp // nx
    xs = lambda idx: wl.u01(wl.SEED_X, idx).astype(np.float32)
    want = np.zeros(len(samp), dtype=np.float32)
    for col, ok, coef in ((samp - nx, gj > 0, -1.0), (samp - 1, gi > 0, -1.0), (samp, np.ones_like(samp, bool), 4.0),
                          (samp + 1, gi < nx - 1, -1.0), (samp + nx, gj < ny - 1, -1.0)):
        term = np.float32(coef) * xs(np.where(ok, col, 0))
        want = np.where(ok, want + term, want).astype(np.float32)
    verified = bool(np.array_equal(y.v[torch.from_numpy(samp).cuda()].cpu().numpy(), want))
    from benchmarks.extra_workloads import warm_up
    w_run = warm_up(job, step, warmup)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    launch_ms = ev0.elapsed_time(ev1) / steps
    b_alg = nnz * 8 + (n + 1) * 4 + 4 * n + 4 * A.ncols_compressed
    from benchmarks.extra_workloads import stored_traffic
    traffic, traffic_source = stored_traffic("poisson2d_spmv_float32", N == 4096)
    rec = {"workload": f"poisson2d 5-pt {N}x{N}, CSR SpMV y=A*x, Float32 values, index=i32",
           "steps": steps, "warmup": w_run, "ms_per_step": round(ms, 5), "gflops": round(2.0 * nnz / (ms * 1e-3) / 1e9, 2),
           "dtype": "f32",
           "roofline": {"bound": "hbm", "achieved": round(b_alg / (launch_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(b_alg / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "traffic_source": traffic_source,
                        "kernel": f32_kernel_name(1), "algorithmic_bytes_per_launch": b_alg,
                        "launch_ms_timed_region": round(launch_ms, 5)},
           "verified_vs_closed_form": verified}
    A = x = y = None
    hp.clear_plan_cache()
    torch.cuda.empty_cache()
    return rec, verified


def configs_digest(result):
    """The other BASELINE configurations of this line as numbers at its top level (ms per step; None = not run or
    skipped), so that a parsed record carries them and not only the sub-records' key names: config 3 (8192^2 over the N
    GPUs; at N = 1 the whole problem on one GPU), config 4 (CG ms per iteration), config 5 (SpMM k = 16: row-major host
    layer, and as a column-major caller gets it), and the headline matrix with the reference's default Int64 indices."""
    oc = result.get("other_configs") or {}
    get = lambda d, *ks: (get(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    return {
        "cfg3_poisson8192_spmv_ms": get(result, "strong_scaling", "ms_per_step"),
        "cfg4_cg_ms_per_iter": get(oc, "poisson3d_cg", "ms_per_step"),
        "cfg5_spmm_rowmajor_ms": get(oc, "sprand_spmm", "ms_per_step"),
        "cfg5_spmm_colmajor_caller_ms": get(oc, "sprand_spmm", "column_major_caller", "via_b_conversion_and_colmajor_store_ms"),
        "headline_int64_ms": get(oc, "int64", "ms_per_step"),
        # round 6: the 5-point matrix x 16 (run tiles) and x 15 (odd k on the padded pitch; was 2 x the time of 16 before)
        "stencil_spmm_k16_ms": get(oc, "poisson2d_spmm", "device_ms_per_step"),
        "stencil_spmm_k15_ms": get(oc, "poisson2d_spmm", "odd_k", "device_ms_per_step"),
    }


def _run(args, budget):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev and os.environ.get("HPCLA_ALLOW_SHARED_GPU", "") != "1":
        raise SystemExit(f"bench.py --gpus {world}: only {ndev} GPU(s) visible (HPCLA_ALLOW_SHARED_GPU=1 lets ranks "
                         "share a device for a functional rehearsal; its timings mean nothing)")
    torch.cuda.set_device(local_rank % ndev)
    budget.stage(f"rank {rank}/{world}: torch imported, device set")
    if os.environ.get("HPCLA_BENCH_VERBOSE", "") == "1":
        import faulthandler                      # diagnostics: where is a rank when a stage takes long?
        faulthandler.dump_traceback_later(float(os.environ.get("HPCLA_BENCH_DUMP_S", "40")), repeat=True, file=sys.stderr)

    import hpcla_amd as hp
    from hpcla_amd import workloads as wl

    Ti = np.int32 if args.index == "i32" else np.int64
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # host-side collectives (plan construction, this script's barriers and maxima) run over gloo; the
        # data path is the library's own: peer windows over xGMI and/or its RCCL communicator
        dist.init_process_group("gloo")
        backend = hp.backend_rocm_mpi(np.float64, Ti)
    else:
        backend = hp.backend_rocm_serial(np.float64, Ti, device_index=torch.cuda.current_device())
    job = Job(torch, dist, world, rank)
    job.attach_data_path(hp, backend)
    stage = budget.stage if rank == 0 else (lambda _name: None)
    stage(f"backend ready (rccl={'yes' if backend.has_rccl else 'no'}, peer windows={'yes' if backend.peer_windows else 'no'})")

    if args.workload not in ("poisson2d", "poisson2d_strong"):
        from benchmarks import extra_workloads          # configs 4/5: separate harness
        return extra_workloads.run(args, backend, rank, world, job)     # its JSON line (rank 0) or None

    # ---- headline: weak scaling, one 4096^2 slab per GPU (N = 1: BASELINE configs[1]) ------------------
    strong = args.workload == "poisson2d_strong"      # whole line on the fixed 8192^2 grid instead
    N = args.size or (args.strong_size if strong else 4096)
    run = PoissonRun(hp, wl, job, backend, args, N, strong, world, rank)
    A, x, y, plan = run.A, run.x, run.y, run.plan
    verified = run.verified
    nnz_loc, nrows_loc, b_alg_loc = run.nnz_loc, run.nrows_loc, run.b_alg_loc
    stage("headline problem built and verified")

    # ---- per-launch duration with HIP events on the launch stream (cross-check of roofline.achieved) ----
    # Taken BEFORE the contract's timed region, on purpose: the first ~30 launches after load begins run 2-4 % slower
    # than the steady state (per-launch device times in profiles/r02_warmup_transient.log: 0.244-0.247 ms for launches
    # 10-30, then 0.235-0.240 -- power management settling, with or without idling first), and `--steps 20 --warmup 5`
    # would time exactly that window.  SURVEY 8d lists "clocks" among the purposes of the warm-up; these 2 x `reps`
    # launches are diagnostics the line reports anyway (launch_ms_event_pairs / _back_to_back / _min / _median), so they
    # are simply run first.  The contract region below is unchanged: W untimed steps, then exactly K timed steps between
    # barrier + synchronize on both sides.
    # BOTH windows are on the line.  `cold_window` is the contract's bracket run FIRST, right behind the plan
    # build: W untimed steps, K timed -- what `--steps 20 --warmup 5` times on a plan that has seen nothing but its own
    # construction (1 verifying launch + the 64 launches of the plan-time block-order measurement); the headline `value`
    # is the same bracket run after the diagnostics below, i.e. in the steady state.
    cold_elapsed, cold_launch_ms = run.time_steps(args.steps, args.warmup)
    stage("cold window timed")
    reps = min(max(args.steps, 20), 200)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    job.barrier()
    for a, b in evs:
        a.record()
        hp.mul_(y, A, x)
        b.record()
    torch.cuda.synchronize()
    per_launch_ms = np.array([a.elapsed_time(b) for a, b in evs])
    launch_ms = float(np.mean(per_launch_ms))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):                # back-to-back launches between ONE event pair
        hp.mul_(y, A, x)
    b.record()
    torch.cuda.synchronize()
    stream_ms = a.elapsed_time(b) / reps
    stage("per-launch cross-check done")

    elapsed, timed_region_launch_ms = run.time_steps(args.steps, args.warmup)
    stage("headline timed")
    timed_out = bool(job.max(1.0 if plan.timed_out() else 0.0))

    # ---- N > 1: every ordering of the distributed step + its pieces -- measured LAST (see _guarded_breakdown:
    #      it drives transports the timed loop did not use, and must not be able to cost the measured result) ----
    selftest = os.environ.get("HPCLA_BENCH_BREAKDOWN_SELFTEST", "") == "1"      # runs the local legs at N = 1
    want_breakdown = (world > 1 and bool(job.max(1.0 if plan.has_halo else 0.0))) or selftest

    # ---- opt-in packed copy (3 B per stored entry instead of 12; same bits), reported separately --------
    packed = None
    if not args.no_packed and world == 1 and budget.allows("packed"):   # an optional extra must not stand between an N > 1 run and its result line
        if A.enable_packed(x):
            hp.mul_(y, A, x)
            torch.cuda.synchronize()
            same = bool(np.array_equal(y.v[torch.from_numpy(run.samp).cuda()].cpu().numpy(), run.want))
            el, _ = run.time_steps(args.steps, args.warmup)
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

    nnz_tot = int(job.sum(nnz_loc))
    b_alg_tot = job.sum(b_alg_loc)
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * nnz_tot / (elapsed / args.steps) / 1e9
    # roofline.achieved / frac: algorithmic bytes of one launch / `ms_per_step` -- the SAME clock as `value` (wall time of
    # the timed region between the barriers, max over ranks; VERDICT r4: the line used to divide `value` by wall time and
    # `frac` by the HIP-event time of the same region, 0.7698 printed against 0.749 recomputed).  The device-event figure
    # (events on the launch stream around the same K launches) stays beside it as achieved_device_events / frac_device_events.
    achieved = b_alg_loc / (ms_per_step * 1e-3) / 1e9
    achieved_events = b_alg_loc / (timed_region_launch_ms * 1e-3) / 1e9

    # (an Int64 matrix on a narrowed plan runs the Int32 kernel: the same stored passes apply)
    traffic, traffic_source = headline_traffic(run.plan.block_group,
                                               run.index_bytes == 4 and N == 4096 and not strong)
    if world > 1 and traffic is not None:
        traffic_source += (f"; N = {world}: the SINGLE-RANK passes of the same per-GPU share (one {N}x{N} slab; the "
                           "distributed launch adds two ghost rows of x and the push stores, < 0.01 % of the bytes)")

    kernel = spmv_kernel_instance(hp, plan.is_i64, plan.has_halo, getattr(plan, "push", False))
    nx, ny, ny_loc, n_glob = run.nx, run.ny, run.ny_loc, run.n_glob
    result = {
        "metric": "SpMV GFLOP/s (2*nnz/t), 2-D 5-pt Poisson, fp64",
        "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"poisson2d 5-pt, {nx}x{ny_loc} slab per GPU ({nx}x{ny} global), "
                               f"n={n_glob}, nnz={nnz_tot}, index={args.index}, CSR SpMV y=A*x",
                   "global_rows": n_glob, "nnz": nnz_tot, "index_type": args.index,
                   "index_bytes_streamed": run.index_bytes, "narrowed": run.narrowed,
                   "parallelism": (f"row-slab x{world}, halo: " + ("peer-window push over xGMI" if getattr(plan, "push", False)
                                                                   else "RCCL send/recv")) if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "clock": "ms_per_step (wall time of the timed region / K, max over ranks): the clock of `value`",
                     "achieved_device_events": round(achieved_events, 1),
                     "frac_device_events": round(achieved_events / HBM_PEAK_GBS, 4),
                     "kernel": kernel, "algorithmic_bytes_per_launch": b_alg_loc,
                     "launch_ms_timed_region": round(timed_region_launch_ms, 5),
                     "launch_ms_event_pairs": round(launch_ms, 5), "launch_ms_back_to_back": round(stream_ms, 5),
                     "launches_before_timed_region": (1 + 2 * reps + 2 * args.warmup + args.steps +
                                                      (64 if run.plan.block_group_measured else 0)),
                     "cold_window": {"ms_per_step": round(cold_elapsed / args.steps * 1e3, 5),
                                     "launch_ms": round(cold_launch_ms, 5),
                                     "frac": round(b_alg_loc / (cold_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "launches_before": 1 + args.warmup + (64 if run.plan.block_group_measured else 0),
                                     "note": "the same bracket (W untimed + K timed steps between barrier + synchronize), run FIRST "
                                             "behind the plan build; the first ~30 launches after load begins run 2-4 % slower than "
                                             "the steady state the headline window measures"},
                     "block_order_group": run.plan.block_group,
                     "launch_ms_min": round(float(per_launch_ms.min()), 5),
                     "launch_ms_median": round(float(np.median(per_launch_ms)), 5)},
        "hbm_gbs_whole_job": round(b_alg_tot / (elapsed / args.steps) / 1e9, 1),
        "hbm_frac_of_peak_whole_job": round(b_alg_tot / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
        "verified_vs_closed_form": verified, "exchange_timed_out": timed_out, "setup_s": round(run.setup_s, 2),
        "packed_csr_opt_in": packed,
    }
    if world > 1:
        # the strong-scaling speed-up (BASELINE: ">= 6x at 8 GPUs over 1") is filled in below; placed at the TOP level
        # next to the transport facts so a reader of `value` (weak scaling) cannot miss it
        result["halo_mode"] = ("push" if getattr(plan, "push", False) else os.environ.get("HPCLA_HALO_MODE", "serial"))
        result["n_ranks_rccl"] = world if backend.has_rccl else 0
        result["peer_windows"] = bool(backend.peer_windows)
        result["strong_scaling_speedup_vs_n1"] = None
    if world == 1 and not args.no_cpu_baseline and budget.allows("cpu_baseline", args.cpu_seconds + 18):   # the contract: rank 0 at N = 1 only
        lo, hi = run.lo, run.hi
        xg = x.local_values()
        if plan.has_halo:
            ghost_idx = A.col_indices[A.col_indices >= hi]
            ghost_lo = A.col_indices[A.col_indices < lo]
            xfull = np.concatenate([wl.u01(wl.SEED_X, ghost_lo), xg, wl.u01(wl.SEED_X, ghost_idx)])
        else:
            xfull = xg
        vals = run.vals if run.vals is not None else A.nzval.cpu().numpy()
        result["cpu_baseline"] = cpu_baseline_spmv(A.rowptr, A.colval, vals, xfull, args.cpu_seconds)
        del vals, xfull
        stage("cpu baseline timed")

    # ---- BASELINE configs[2], the strong-scaling problem, as a sub-record of the same line ----------------
    # Optional stages run in order of value and only while the budget covers them (benchmarks/budget.py): a stage
    # that would not fit is recorded as {"skipped": "budget"}, decided by all ranks together.
    from benchmarks.budget import SKIPPED
    if not strong and not args.no_strong:
        run.release()
        del A, x, y, plan
        hp.clear_plan_cache()
        torch.cuda.empty_cache()
        job.barrier()
        if job.agree(budget.allows("strong_scaling")):
            try:
                rec, ok = strong_scaling_record(hp, wl, job, backend, args, world, rank, budget)
                verified = verified and ok and not rec.get("timed_out", False)
            except Exception as exc:                     # same code on every rank: all ranks land here together
                rec = {"error": f"{type(exc).__name__}: {exc}"}
        else:
            rec = dict(SKIPPED)
        result["strong_scaling"] = rec
        if world > 1:
            result["strong_scaling_speedup_vs_n1"] = rec.get("speedup_vs_n1")
        stage("strong-scaling sub-record done")
        if world > 1:
            # the OTHER reading of BASELINE's ">= 6x at 8 GPUs over 1" (north_star: "n ~ 10^8 nnz"): the 4096^2 headline
            # matrix (83.9 M entries) strong-scaled, next to the 8192^2 one of configs[2] -- both on record, each with its
            # own speedup_vs_n1 (at N = 1 it would be the headline itself)
            hp.clear_plan_cache()
            torch.cuda.empty_cache()
            job.barrier()
            if job.agree(budget.allows("strong_scaling_4096")):
                try:
                    rec4, ok4 = strong_scaling_record(hp, wl, job, backend, args, world, rank, budget, size=4096,
                                                      n1_key="strong_scaling_4096_n1")
                    verified = verified and ok4 and not rec4.get("timed_out", False)
                except Exception as exc:
                    rec4 = {"error": f"{type(exc).__name__}: {exc}"}
            else:
                rec4 = dict(SKIPPED)
            result["strong_scaling_4096"] = rec4
            result["strong_scaling_4096_speedup_vs_n1"] = rec4.get("speedup_vs_n1")
            stage("strong-scaling (4096^2) sub-record done")
    # ---- the other BASELINE configs as sub-records of the same line (driver-timed): configs[3]'s per-GPU share
    #      (3-D Poisson, 100 CG iterations) and configs[4] (SpMM, k = 16): config 5 at its own gather set (B = 2^24 x 16
    #      rows whatever N is: at N = 1 all of it local -- the HBM side of the 8-GPU job), the 5-point matrix x 16
    #      (the kernel's own efficiency), and the Infinity-Cache-sized B of round 2 labelled as such ----------------
    if not strong and not args.no_extras:
        import copy
        from benchmarks import extra_workloads
        extras = {}
        if world == 1:
            if budget.allows("int64"):
                try:
                    extras["int64"], ok = int64_record(hp, wl, job, args, N, min(args.steps, 50), min(args.warmup, 10))
                    verified = verified and ok
                except Exception as exc:
                    extras["int64"] = {"error": f"{type(exc).__name__}: {exc}"}
            else:
                extras["int64"] = dict(SKIPPED)
            stage("int64 sub-record done")
            if budget.allows("float32"):
                try:
                    extras["float32"], ok = float32_record(hp, wl, job, args, N, min(args.steps, 50), min(args.warmup, 10))
                    verified = verified and ok
                except Exception as exc:
                    extras["float32"] = {"error": f"{type(exc).__name__}: {exc}"}
            else:
                extras["float32"] = dict(SKIPPED)
            stage("float32 sub-record done")
        todo = [("poisson3d_cg", "poisson3d_cg", 100, 0), ("sprand_spmm", "sprand_spmm", 10, max(1, 8 // world)),
                ("poisson2d_spmm", "poisson2d_spmm", 20, 0)]
        if world == 1:
            todo.append(("sprand_spmm_mall_sized", "sprand_spmm", 10, 1))
        else:
            # config 5 once more in the opt-in PANEL order (exchange overlapped chunk by chunk), so that
            # one N > 1 run carries both orders of the same step and their xGMI rooflines side by side
            todo.append(("sprand_spmm_panel_order", "sprand_spmm", 10, max(1, 8 // world)))
        only = [t for t in os.environ.get("HPCLA_BENCH_EXTRAS", "").split(",") if t]       # diagnostics: a subset, in this order
        if only:
            todo = [t for nm in only for t in todo if t[0] == nm]
        for name, workload, wsteps, mult in todo:
            a2 = copy.copy(args)
            a2.workload, a2.steps, a2.warmup, a2.size, a2.cols_mult = workload, wsteps, 5, 0, mult
            a2.spmm_order = "panel" if name == "sprand_spmm_panel_order" else None
            if not job.agree(budget.allows(name)):
                extras[name] = dict(SKIPPED)
                continue
            try:
                extras[name] = extra_workloads.run_record(a2, backend, rank, world, job)
            except Exception as exc:                 # same code on every rank: all ranks land here together
                extras[name] = {"error": f"{type(exc).__name__}: {exc}"}
            stage(f"{name} sub-record done")
        result["other_configs"] = extras
    result["configs_digest"] = configs_digest(result)
    verified = verified and not timed_out
    result["verified_vs_closed_form"] = verified
    job.barrier()
    if not verified:
        sys.stderr.write("bench: result verification FAILED\n")
    breakdown_fn = None
    if want_breakdown and not job.agree(budget.allows("comparison")):
        if rank == 0:
            result["step_breakdown_ms_max_over_ranks"] = dict(SKIPPED)
        want_breakdown = False
    result["budget"] = {"outer_limit_s": budget.outer, "elapsed_s": round(budget.elapsed(), 1), "skipped": list(budget.skipped)}
    if want_breakdown:
        def breakdown_fn():              # on a fresh copy of the headline problem (the plan caches were cleared above)
            hp.clear_plan_cache()
            run2 = PoissonRun(hp, wl, job, backend, args, N, strong, world, rank)
            return step_breakdown(hp, job, backend, run2.plan, run2.A, run2.x, run2.y)
    return Outcome(result if rank == 0 else None, verified, breakdown_fn)


if __name__ == "__main__":
    main()
