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
import sys
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


def stored_traffic(key, applicable=True, world=1, share_note=""):
    """HBM bytes per launch / iteration measured by the builder's separate rocprofv3 --pmc passes of the same
    workload (profiles/traffic_latest.json, written by benchmarks/collect_profiles.py; corrected for gfx950 as
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside the process, so the value is a
    stored measurement and labelled as such; (None, reason) when this run's shape differs from the stored one.
    At N > 1 the value is the SINGLE-RANK pass of the same per-GPU share (the records are weak-scaled: every rank runs
    the stored workload's kernel on its own share) and the source says so; `share_note` names any difference."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_latest.json")
    try:
        rec = json.load(open(path)).get("workloads", {}).get(key)
    except Exception:
        rec = None
    if not applicable or not rec:
        return None, "no stored PMC measurement for this shape (profiles/traffic_latest.json)"
    src = ("NOT measured by this run: PMC counters cannot be read from inside the process; value stored by "
           "the builder's rocprofv3 passes of this workload -- " + rec.get("source", path))
    if world > 1:
        src += f"; N = {world}: the SINGLE-RANK passes of the same per-GPU share (per GPU, like `achieved`)"
    if share_note:
        src += "; " + share_note
    return rec["hbm_bytes"], src


TRAFFIC_CONVENTION = ("traffic = 2 x FETCH_SIZE + WRITE_SIZE: the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte "
                      "requests at 64 B), calibrated on WIDE streaming reads.  A random gather fetches one 64-byte sector per "
                      "gathered value, for which the doubling overcounts: the bytes actually moved lie between "
                      "traffic_undoubled_fetch (= FETCH_SIZE + WRITE_SIZE as counted) and traffic")


def stored_traffic_undoubled(key, applicable=True):
    """FETCH_SIZE + WRITE_SIZE of the stored passes AS COUNTED (no gfx950 doubling), or None: the lower bracket of a
    gather-dominated record's HBM bytes."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_latest.json")
    try:
        rec = json.load(open(path)).get("workloads", {}).get(key)
        if not applicable or not rec:
            return None
        return int(round((rec["FETCH_SIZE_KB_mean"] + rec["WRITE_SIZE_KB_mean"]) * 1024))
    except Exception:
        return None


_T0 = time.perf_counter()


def _stamp(msg):
    """HPCLA_BENCH_VERBOSE=1: per-rank progress lines on stderr (where does a record's wall time go)."""
    if os.environ.get("HPCLA_BENCH_VERBOSE", "") == "1":
        sys.stderr.write(f"[extra +{time.perf_counter() - _T0:7.2f}s rank {os.environ.get('RANK', '0')}] {msg}\n")
        sys.stderr.flush()


def _sync_stamp(msg):
    """Verbose runs: synchronise the device, then stamp (which piece of GPU work does a stall sit in?)."""
    if os.environ.get("HPCLA_BENCH_VERBOSE", "") == "1":
        import torch
        torch.cuda.synchronize()
        _stamp(msg)


def _sync_barrier(job, closing=False):
    job.barrier(device_only=closing)      # closing bracket of a timed region: device rendezvous only (bench.py Job)


XGMI_LINK_GBS = 153.0        # MI355X_MICROARCH.md / SURVEY 5: 7 xGMI links per GPU x ~153 GB/s each way, point to point
XGMI_LINKS = 7


SETTLE_MS = float(os.environ.get("HPCLA_BENCH_SETTLE_MS", "40"))


def warm_up(job, step, w_min):
    """Untimed warm-up of a sub-record: at least `w_min` steps AND at least SETTLE_MS of them (HPCLA_BENCH_SETTLE_MS, default
    40 ms).  SURVEY 8d lists "clocks" among the purposes of the warm-up: the first ~10 ms after load begins run at clocks
    that are still ramping (profiles/r02_warmup_transient.log: 2-4 % on the SpMV; profiles/r04_spmm_warmup_ramp.log: the
    LDS-heavy run-tile SpMM 0.500 ms after 5 warm-up products, 0.474 after 100, while the latency-bound gather kernel reads
    0.525 either way) -- five launches of a 0.5 ms kernel end inside that ramp.  Returns the number of steps run (reported
    as the record's `warmup`); every rank runs the same count (decided by rank 0's clock through job.max)."""
    import torch
    n = 0
    for _ in range(w_min):
        step()
        n += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-6)
    n += 1
    more = int(job.max(max(0.0, SETTLE_MS * 1e-3 / one - n)))
    for _ in range(min(more, 2000)):
        step()
    torch.cuda.synchronize()
    return n + min(more, 2000)


def _measure_spmm(args, job, hp, wl, A, B, k, world, dev, setup_s, metric, workload, traffic_key=None, share_note=""):
    import torch
    _stamp("spmm: operands ready")
    C = A @ B
    if os.environ.get("HPCLA_BENCH_VERBOSE", "") == "1":
        torch.cuda.synchronize()
        _stamp("spmm: first product (plan built)")
    keep = [C]

    def one_product():
        keep[0] = A @ B
    n_warm = warm_up(job, one_product, max(args.warmup, 5))
    steps = min(args.steps, 50)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _sync_barrier(job)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        C = A @ B
    ev1.record()
    _sync_barrier(job, closing=True)
    elapsed = time.perf_counter() - t0
    elapsed = job.max(elapsed)
    device_ms = job.max(ev0.elapsed_time(ev1) / steps)
    ms = elapsed / steps * 1e3
    b_alg = wl.spmm_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, k, 4)
    b_gather = A.nnz * (12 + 8 * k) + 4 * A.nrows_local + 8 * k * A.nrows_local    # every B row read per entry
    traffic, traffic_source = stored_traffic(traffic_key, traffic_key is not None, world, share_note)
    out = {
        "metric": metric, "value": round(2.0 * k * A.nnz * world / (ms * 1e-3) / 1e9, 1),
        "unit": "GFLOP/s", "n_gpus": world, "steps": steps, "warmup": n_warm, "ms_per_step": round(ms, 4),
        "device_ms_per_step": round(device_ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload,
                   "ncols_compressed": A.ncols_compressed, "ncols": int(B.row_partition[-1])},
        "roofline": {"bound": "hbm", "achieved": round(b_alg / (device_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(b_alg / (device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     **({"traffic_undoubled_fetch": stored_traffic_undoubled(traffic_key), "traffic_convention": TRAFFIC_CONVENTION}
                        if (traffic_key or "").startswith("sprand") and traffic is not None else {}),
                     "algorithmic_bytes_per_launch": b_alg,
                     "block_order_group": int(job.max(hp.spmm_block_order_of(A, B))),
                     "run_tiles": (lambda f: {"used": False, "kernel": "hpcla::spmm_rowblock_vec_kernel",
                                              "note": "run descriptors not built for this product (k != 16, Float32, or the panel order)"} if f is None else
                                   {"blocks_that_fit": f[0], "blocks": f[1], "used": bool(f[0] >= 0.99 * f[1]),
                                    "kernel": "hpcla::spmm_rowblock_runs_kernel" if f[0] >= 0.99 * f[1] else "hpcla::spmm_rowblock_vec_kernel"})(
                                       hp.spmm_runs_fit_of(A, B)),
                     "gather_bytes_per_launch": b_gather,
                     "gather_gbs": round(b_gather / (device_ms * 1e-3) / 1e9, 1),
                     "note": "achieved = algorithmic bytes / device time per step (HIP events on the launch stream); algorithmic "
                             "bytes count each touched B row once; a random-column matrix re-reads B rows "
                             "(gather_bytes = one 128-byte line per stored entry), which is what HBM actually serves"},
        "setup_s": round(setup_s, 2),
    }
    if world == 1 and os.environ.get("HPCLA_BENCH_COLMAJOR", "1") != "0":
        # banded structure: the direct column-major product AND the conversions; unstructured: the conversions only (the direct
        # form touches a line per (entry, column) pair there -- the Julia extension never takes it)
        out["column_major_caller"] = _colmajor_cost(hp, A, B, k, b_alg, direct_too="5-pt" in workload)
    if world > 1:
        # the exchange side of the step ("report against both HBM and xGMI rooflines"):
        # ghost rows of B that cross xGMI per step, against 7 point-to-point links of ~153 GB/s per direction
        xin, xout, n_peers_in, n_peers_out = hp.spmm_exchange_bytes(A, B)
        peak_in = XGMI_LINK_GBS * max(min(n_peers_in, XGMI_LINKS), 1)
        t_comm_bound = job.max(max(xin / (XGMI_LINK_GBS * max(min(n_peers_in, XGMI_LINKS), 1)),
                                   xout / (XGMI_LINK_GBS * max(min(n_peers_out, XGMI_LINKS), 1))) / 1e9 * 1e3)    # ms
        t_hbm_bound = job.max(b_alg / (HBM_PEAK_GBS * 1e9) * 1e3)
        out["roofline_xgmi"] = {
            "bound": "xgmi", "bytes_in_per_gpu_per_step": int(job.max(xin)), "bytes_out_per_gpu_per_step": int(job.max(xout)),
            "peers_in": int(job.max(n_peers_in)), "peers_out": int(job.max(n_peers_out)),
            "achieved_in_gbs": round(job.max(xin) / (ms * 1e-3) / 1e9, 1), "link_gbs": XGMI_LINK_GBS, "links": XGMI_LINKS,
            "peak_in_gbs": round(job.max(peak_in), 1),
            "frac": round(job.max(xin) / (ms * 1e-3) / 1e9 / job.max(peak_in), 4) if xin or world > 1 else None,
            "comm_bound_ms": round(t_comm_bound, 4), "hbm_bound_ms": round(t_hbm_bound, 4),
            "step_sits_on": "xgmi" if t_comm_bound > t_hbm_bound else "hbm",
            "order": os.environ.get("HPCLA_SPMM_ORDER", "sequential"),
            "note": "bytes = ghost rows of B (8k bytes each) received / sent by the busiest rank per step; peak = one xGMI link "
                    "per peer (point to point), at most 7; frac = achieved ingress / that peak over the WALL time of a step "
                    "(exchange and kernel together); comm_bound / hbm_bound = the step's two lower bounds"}
    return out


def _spmm_order_in_force(plan, rowptr):
    """The SpMM block-order group the host layer has set in the library for this rowptr (0 = natural)."""
    return int(plan.__dict__.get("_spmm_order_in_force", {}).get(rowptr.data_ptr(), 0))


def _colmajor_cost(hp, A, B, k, b_alg, direct_too=True):
    """What the same product costs a COLUMN-major caller -- Julia's Matrix, the layout of the reference's dense block
    (src/dense.jl:63) -- through the raw C ABI on the plan's own arrays: (a) the product on the column-major blocks as they are
    (csrc/colmajor.hip: hpcla_spmm_csr_f64_* with both layouts HPCLA_LAYOUT_COL), which is what the Julia extension calls for
    banded matrices, against (b) the two layout conversions around the row-major product it used before."""
    import torch
    from hpcla_amd.sparse import get_vector_plan
    from hpcla_amd.vectors import HPCVector, current_stream_ptr, dptr
    from hpcla_amd.partition import compute_partition_hash
    capi = hp._capi
    probe = HPCVector(compute_partition_hash(B.row_partition), B.row_partition, B.A[:, 0], A.backend)
    plan = get_vector_plan(A, probe)
    sfx = "i64" if plan.is_i64 else "i32"
    rp, cv = plan.rowptr_of(A), plan.colval_split
    n, nb = A.nrows_local, int(B.A.shape[0])
    Bc = B.A.t().contiguous()                                   # k x nb: the column-major block
    Cc = torch.empty((k, n), dtype=torch.float64, device=B.A.device)
    Br, Cr = torch.empty_like(B.A), torch.empty((n, k), dtype=torch.float64, device=B.A.device)
    ROW, COL = capi.LAYOUT_ROW, capi.LAYOUT_COL
    s = current_stream_ptr()

    def direct():
        capi.call(f"hpcla_spmm_csr_f64_{sfx}", dptr(rp), dptr(cv), dptr(A.nzval), dptr(Bc), nb, COL, dptr(Cc), n, COL, n, A.nnz, k, 0, s)

    def converted():
        capi.call("hpcla_transpose_f64", dptr(Bc), nb, COL, dptr(Br), k, ROW, nb, k, s)
        capi.call(f"hpcla_spmm_csr_f64_{sfx}", dptr(rp), dptr(cv), dptr(A.nzval), dptr(Br), k, ROW, dptr(Cr), k, ROW, n, A.nnz, k, 0, s)
        capi.call("hpcla_transpose_f64", dptr(Cr), k, ROW, dptr(Cc), n, COL, n, k, s)

    def b_only():
        capi.call("hpcla_transpose_f64", dptr(Bc), nb, COL, dptr(Br), k, ROW, nb, k, s)

    def product_only():
        capi.call(f"hpcla_spmm_csr_f64_{sfx}", dptr(rp), dptr(cv), dptr(A.nzval), dptr(Br), k, ROW, dptr(Cr), k, ROW, n, A.nnz, k, 0, s)

    def ccol():
        # round 5: B converted once, the product writes C column-major itself (csrc/spmm.hip CCOL) -- what the Julia extension
        # calls for an unstructured matrix
        capi.call("hpcla_transpose_f64", dptr(Bc), nb, COL, dptr(Br), k, ROW, nb, k, s)
        capi.call(f"hpcla_spmm_csr_f64_{sfx}", dptr(rp), dptr(cv), dptr(A.nzval), dptr(Br), k, ROW, dptr(Cc), n, COL, n, A.nnz, k, 0, s)

    def timed(fn, reps=20):
        t_end = time.perf_counter() + SETTLE_MS * 1e-3
        while time.perf_counter() < t_end:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    converted()
    ref = Cc.clone()
    ccol()
    same_ccol = bool(torch.equal(ref, Cc))
    if not direct_too:
        ms_c, ms_cc, ms_b, ms_p = timed(converted, 10), timed(ccol, 10), timed(b_only, 10), timed(product_only, 10)
        return {"via_b_conversion_and_colmajor_store_ms": round(ms_cc, 4),
                "frac_of_peak": round(b_alg / (ms_cc * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "same_bits": same_ccol,
                "b_conversion_ms": round(ms_b, 4), "rowmajor_product_ms": round(ms_p, 4),
                "via_two_layout_conversions_ms": round(ms_c, 4),
                "kernel": "hpcla::spmm_rowblock_vec_kernel<..., CCOL = true>",
                "note": "the product as a column-major caller (Julia's Matrix) gets it on an UNSTRUCTURED matrix: B converted to "
                        "row-major rows, then the row-major-B product that stores C column-major itself (round 5; before: C "
                        "converted back by a second transposition = via_two_layout_conversions_ms); the direct column-major form "
                        "would touch a line per (entry, column) pair here; the record's own ms_per_step is the row-major host layer"}
    direct()
    same = bool(torch.equal(ref, Cc))
    ms_d, ms_c = timed(direct), timed(converted)
    ms_cc = timed(ccol)
    out = {"direct_ms": round(ms_d, 4), "direct_frac_of_peak": round(b_alg / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "kernel": "hpcla::rowgather_kernel<double, int, false, 16, 2>",
           "via_two_layout_conversions_ms": round(ms_c, 4), "same_bits": same and same_ccol,
           "via_b_conversion_and_colmajor_store_ms": round(ms_cc, 4),
           "note": "the product as a column-major caller (Julia's Matrix) gets it: on the blocks as they are (lanes = rows kernel; "
                   "run_tiles_ms: the run tiles on the column-major blocks, round 5 -- what the Julia extension calls at k = 16 on a "
                   "structure whose blocks fit) against transpose + row-major product + transpose; the record's own ms_per_step is "
                   "the row-major host layer"}
    # round 5: the run tiles on the column-major blocks (k = 16, every column owned here, B's columns on the 16-byte grid)
    from hpcla_amd.dense import _spmm_runs
    desc = _spmm_runs(A, plan, rp, cv, plan.is_i64) if k == 16 else None
    if desc is not None and nb % 2 == 0 and Bc.data_ptr() % 16 == 0 and plan.n_own == nb:
        def runs_direct():
            capi.call(f"hpcla_spmm_runs_colmajor_k16_f64_{sfx}", dptr(rp), dptr(cv), dptr(A.nzval), dptr(Bc), nb, None, 0, nb, dptr(Cc), n, n,
                      A.nnz, 0, dptr(desc), None, 0, s)
        import ctypes
        in_force = _spmm_order_in_force(plan, rp)
        chosen = ctypes.c_int(1)
        Cc.fill_(float("nan"))
        # plan time, as the Julia extension does it: the launch timed under five block orders, the fastest stays set for rowptr
        capi.call(f"hpcla_spmm_runs_colmajor_tune_block_order_f64_{sfx}", dptr(rp), dptr(cv), dptr(A.nzval), dptr(Bc), nb, None, 0, nb,
                  dptr(Cc), n, n, A.nnz, 0, dptr(desc), None, 0, s, ctypes.byref(chosen))
        out["same_bits"] = out["same_bits"] and bool(torch.equal(ref, Cc))
        Cc.fill_(float("nan"))
        runs_direct()
        out["same_bits"] = out["same_bits"] and bool(torch.equal(ref, Cc))
        ms_r = timed(runs_direct)
        capi.call("hpcla_spmm_block_order_hint", dptr(rp), in_force)      # the host layer's own (row-major) order back in force
        out.update({"run_tiles_ms": round(ms_r, 4), "run_tiles_block_order_group": int(chosen.value), "run_tiles_frac_of_peak": round(b_alg / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "run_tiles_kernel": "hpcla::spmm_runs_colmajor_kernel<int, false>" if sfx == "i32" else "hpcla::spmm_runs_colmajor_kernel<long, false>"})
    return out


def _sprand_spmv(hp, wl, job, A, ncols, backend, args):
    """The unstructured matrix of config 5 times ONE vector (N = 1): the shape of the reference's own single-rank SpMV
    benchmark (tools/benchmark_single_rank.jl:48-71: random columns, ~10 per row; here config 5's ~30), on the default
    row-gather kernel.  Every stored entry gathers one x value from a random place: 8 useful bytes out of a 64-byte sector
    (128-byte line), so the each-value-once algorithmic count is far from what the memory system serves -- the record carries
    both, and the stored PMC traffic."""
    import torch
    xv = hp.HPCVector.zeros(np.array([0, ncols]), backend)
    hp._capi.call("hpcla_fill_uniform_f64", xv.v.data_ptr(), 0, ncols, wl.SEED_X, torch.cuda.current_stream().cuda_stream)
    yv = A @ xv
    step = lambda: hp.mul_(yv, A, xv)
    n_warm = warm_up(job, step, max(args.warmup, 5))
    steps = min(args.steps, 50)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    dev_ms = ev0.elapsed_time(ev1) / steps
    plan = hp.get_vector_plan(A, xv)
    b_alg = wl.spmv_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, 4)
    b_sect = A.nnz * (12 + 64) + 12 * A.nrows_local
    mall = ncols * 8 <= (1 << 25)           # the small sibling: x of 17 MB (8 L2s x 4 MiB hold most of it)
    tkey = "sprand_spmv_mall_sized" if mall else "sprand_spmv_b2e24"
    stored_shape = A.nrows_local == 2_097_152 and ncols in (2_097_152, 16_777_216)
    traffic, traffic_source = stored_traffic(tkey, stored_shape)
    return {"metric": "SpMV GFLOP/s (2*nnz/t), sprand ~29.8 nnz/row, fp64", "value": round(2.0 * A.nnz / (ms * 1e-3) / 1e9, 1),
            "unit": "GFLOP/s", "steps": steps, "warmup": n_warm, "ms_per_step": round(ms, 4), "device_ms_per_step": round(dev_ms, 4),
            "config": {"workload": f"sprand-like {A.nrows_local} x {ncols}, nnz={A.nnz}, CSR SpMV y=A*x, index=i32; x = {ncols * 8 / 1e6:.0f} MB "
                                   + ("(about the size of the eight L2s together)" if mall else "(beyond the L2s, inside the 256 MiB Infinity Cache)"),
                       "ncols_compressed": A.ncols_compressed},
            "roofline": {"bound": "hbm", "achieved": round(b_alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_undoubled_fetch": stored_traffic_undoubled(tkey, stored_shape), "traffic_convention": TRAFFIC_CONVENTION,
                         "kernel": "hpcla::spmv_rowgather_kernel<int, false, false, false>", "algorithmic_bytes_per_launch": b_alg,
                         "block_order_group": int(getattr(plan, "block_group", 1)),
                         "sector_gather_bytes_per_launch": b_sect, "sector_gather_gbs": round(b_sect / (ms * 1e-3) / 1e9, 1),
                         "note": "achieved = algorithmic bytes (each x value once) / WALL time per step; sector_gather = 12 B + one 64-byte "
                                 "sector of x per stored entry -- what a random gather asks of the memory system"}}


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
        native = os.environ.get("HPCLA_CG_PYTHON_LOOP", "") != "1"   # default: hpcla_cg_iterations_* (one host call)
        # Everything that is not an iteration stays OUTSIDE the timed region: the workspace (x, r, p, Ap, history),
        # the plan, x0 = 0 / r0 = p0 = b / sum r0^2 (cg_setup), and the history read-back.  Warm-up: >= 5
        # iterations through the SAME code path that is then timed.
        ws = hp.CGWorkspace(b, max(iters, 8) + 2)
        plan, fused_eff = hp.cg_setup(A, b, ws, fused)
        n_warm = max(args.warmup, 5)
        gp = None
        if graph:
            hp.cg_iterate(A, ws, plan, fused_eff, 2, native_loop=False)
            gp = hp.CGGraphPair(A, ws, plan, fused_eff)
            gp.replay(max(n_warm // 2, 3))
        else:
            # at least n_warm iterations and at least SETTLE_MS of them (clocks: see warm_up), restarting the solve whenever
            # the workspace's history is used up
            done = [0]

            def warm_iterations():
                if ws.done + 5 > ws.max_iters:
                    hp.cg_setup(A, b, ws, fused)
                hp.cg_iterate(A, ws, plan, fused_eff, 5, native_loop=native)
                done[0] += 5
            warm_up(job, warm_iterations, (n_warm + 4) // 5)
            n_warm = done[0]
        torch.cuda.synchronize()
        plan, fused_eff = hp.cg_setup(A, b, ws, fused)          # restart from x0 = 0
        if graph:
            hp.cg_iterate(A, ws, plan, fused_eff, 2, native_loop=False)
            iters_timed = (iters - 2) // 2 * 2
        else:
            iters_timed = iters
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _sync_barrier(job)
        t0 = time.perf_counter()
        ev0.record()                       # HIP events on the launch stream, bracketing the iterations themselves
        if graph:
            gp.replay(iters_timed // 2)
        else:
            hp.cg_iterate(A, ws, plan, fused_eff, iters_timed, native_loop=native)
        ev1.record()
        t_enqueued = time.perf_counter() - t0          # host time to enqueue all iterations (nothing waited for)
        _sync_barrier(job, closing=True)
        elapsed = time.perf_counter() - t0
        elapsed = job.max(elapsed)
        device_ms_iter = job.max(ev0.elapsed_time(ev1) / iters_timed)
        host_enqueue_ms_iter = job.max(t_enqueued / iters_timed * 1e3)
        if ws.done < iters:                                     # graph mode: the odd remainder, untimed
            hp.cg_iterate(A, ws, plan, fused_eff, iters - ws.done, native_loop=False)
        hist = ws.hist[:iters + 1].sqrt().cpu().tolist()       # read-back after the timed region
        n_loc, nnz_loc = A.nrows_local, A.nnz
        b_spmv = wl.spmv_algorithmic_bytes(nnz_loc, n_loc, A.ncols_compressed, 4)
        b_iter = b_spmv + 96 * n_loc        # textbook unfused CG = SpMV + 96 n bytes
        b_moved = b_spmv + (64 if fused_eff else 96) * n_loc
        ms_iter = elapsed / iters_timed * 1e3
        traffic, traffic_source = stored_traffic("poisson3d_cg_iteration", N == 512 and fused_eff, world,
                                                 "the distributed step adds two ghost planes (2 x 2 MiB) per iteration" if world > 1 else "")
        out = {
            "metric": "CG ms/iteration, 3-D 7-pt Poisson, fp64", "value": round(ms_iter, 4), "unit": "ms/iter",
            "n_gpus": world, "steps": iters_timed, "warmup": n_warm, "ms_per_step": round(ms_iter, 4),
            "wall_ms_per_iter": round(ms_iter, 4), "device_ms_per_iter": round(device_ms_iter, 4),
            "host_enqueue_ms_per_iter": round(host_enqueue_ms_iter, 5),
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"poisson3d 7-pt {N}x{N}x{planes} per GPU ({N}x{N}x{nz} global), {iters_timed} CG iterations, "
                                   f"{'fused: SpMV+p.Ap, r-update+r.r, x/p-update' if fused_eff else 'one kernel per reference operator'}"
                                   f"{', HIP graph replay' if graph else (', one host call for all iterations (hpcla_cg_iterations)' if native and fused_eff else ', one host call per kernel')}",
                       "global_rows": n_glob, "nnz_per_gpu": nnz_loc},
            "roofline": {"bound": "hbm", "achieved": round(b_iter / (ms_iter * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(b_iter / (ms_iter * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_iteration": b_iter,
                         "moved_bytes_per_iteration": b_moved,
                         "moved_gbs": round(b_moved / (ms_iter * 1e-3) / 1e9, 1),
                         "frac_moved_bytes": round(b_moved / (ms_iter * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "whole iteration (SpMV, 2 reductions, the x / r / p updates); achieved = bytes / WALL clock per iteration "
                                 "(max over ranks); device_ms_per_iter = HIP events on the launch stream around the same iterations; "
                                 "bytes = the textbook unfused count of SURVEY 8d (SpMV + 96 n), the fused form moves SpMV + 64 n"},
            "block_order_group": int(getattr(plan, "block_group", 1)),
            "residual_first": hist[0], "residual_last": hist[-1], "setup_s": round(setup_s, 2),
            "exchange_timed_out": bool(job.max(1.0 if hp.get_vector_plan(A, b).timed_out() else 0.0)),
        }
        del ws, gp
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
                            f"poisson2d 5-pt {nx}x{ny_loc} slab per GPU, nnz/GPU={A.nnz}, k={k}, C = A*B",
                            traffic_key="poisson2d_spmm" if N == 4096 else None)
        # ODD k (round 6): the same matrix times 15 columns through the host layer -- B's rows on the padded pitch 16
        # (dense.spmm_pitch), the vector kernel with the last column pair's second half masked; before round 6 every odd k ran
        # on the one-column-per-lane kernel (0.95 ms on this shape where 16 columns took 0.52 on the same gather kernel)
        try:
            k15 = 15
            B15 = hp.HPCMatrix_local(Bl[:, :k15].contiguous(), backend)
            C15 = A @ B15                                    # plan build + (first product of a result on the padded pitch)
            B15p = hp.HPCMatrix(B15.row_partition, B15.col_partition, hp.dense._rows_on_pitch(B15.A, hp.dense.spmm_pitch(A, k15))[:, :k15],
                                backend)                     # as a chained product finds it: already on the pitch, no copy
            for _ in range(3):
                C15 = A @ B15p
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n15 = max(5, min(args.steps, 20))
            job.barrier()
            ev0.record()
            for _ in range(n15):
                C15 = A @ B15p
            ev1.record()
            torch.cuda.synchronize()
            ms15 = job.max(ev0.elapsed_time(ev1) / n15)
            b15 = wl.spmm_algorithmic_bytes(A.nnz, A.nrows_local, A.ncols_compressed, k15, 4)
            out["odd_k"] = {"k": k15, "row_pitch": int(B15p.A.stride(0)), "device_ms_per_step": round(ms15, 4),
                            "frac": round(b15 / (ms15 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": b15,
                            "vs_k16_this_record": round(ms15 / out["device_ms_per_step"], 3),
                            "kernel": "hpcla::spmm_rowblock_vec_kernel (K16 shape, kr = 15)",
                            "note": "k = 16 above takes the run-tile kernel; on the same gather kernel 16 columns take what 15 take "
                                    "(profiles/r06_spmm_odd_k.log: 0.518 / 0.519 ms at 4096 x 2048 rows)"}
            del B15, B15p, C15
        except Exception as exc:                              # an extra: never takes the record down with it
            out["odd_k"] = {"error": f"{type(exc).__name__}: {exc}"}
    elif args.workload == "sprand_spmm":
        k = 16
        rows_loc = args.size or 2_097_152
        # HPCLA_SPMM_COLS_MULT=8 on ONE GPU reproduces config 5's per-GPU access pattern (B has
        # 8 x 2 097 152 rows = 2.1 GB, far beyond the 256 MiB Infinity Cache) without the exchange
        mult = int(getattr(args, "cols_mult", 0) or os.environ.get("HPCLA_SPMM_COLS_MULT", "1"))
        ncols = rows_loc * world * mult
        mean_nnz = 29.8
        # generated ON THE DEVICE (torch is plumbing here): counts ~ Poisson(29.8) (= Binomial(ncols, 29.8/ncols)
        # to 1e-6), columns uniform, sorted within rows by one 64-bit key sort; then the library's device-side
        # column compression.  The numpy version of this setup (binomial + lexsort of 6e7 keys) took 17 s.
        t0 = time.perf_counter()
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xA11CE + rank)
        # Ranks that SHARE a GPU (HPCLA_ALLOW_SHARED_GPU=1 rehearsals) generate one after the other: four processes'
        # torch.sort calls (rocPRIM onesweep, decoupled look-back) running on one GPU at once made no visible progress for
        # 40-300 s (all ranks inside the sort's synchronisation, GPU 100 % busy, no memory traffic; gpurun_out/r03_reh5.log).
        # One process per GPU -- the real layout -- never has two sorts on one device.
        take_turns = world > 1 and os.environ.get("HPCLA_ALLOW_SHARED_GPU", "") == "1"
        for turn in range(rank if take_turns else 0):
            job.barrier()
        _sync_stamp("sprand: start (device idle)")
        counts = torch.poisson(torch.full((rows_loc,), mean_nnz, dtype=torch.float64, device=dev), generator=gen).to(torch.int64)
        _sync_stamp("sprand: poisson counts")
        rowptr = torch.zeros(rows_loc + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=rowptr[1:])
        nnz = int(rowptr[-1].item())
        _sync_stamp("sprand: cumsum")
        cols = torch.randint(0, ncols, (nnz,), generator=gen, device=dev, dtype=torch.int64)
        _sync_stamp("sprand: randint")
        rowid = torch.repeat_interleave(torch.arange(rows_loc, device=dev, dtype=torch.int64), counts)
        _sync_stamp("sprand: repeat_interleave")
        key = torch.sort(rowid * ncols + cols).values
        _sync_stamp("sprand: sort")
        cols = key - rowid * ncols
        del key, rowid, counts
        vals = torch.rand(nnz, generator=gen, device=dev, dtype=torch.float64)
        if take_turns:
            torch.cuda.synchronize()
            for turn in range(rank, world):
                job.barrier()
        _sync_stamp("sprand: entries generated")
        A = hp.HPCSparseMatrix_local_device(rowptr, cols, vals, ncols, backend, col_window=(0, ncols - 1))
        del cols
        if os.environ.get("HPCLA_SPRAND_SPMV", "") == "1":       # the SpMV record alone (PMC passes: benchmarks/collect_profiles.py)
            rec = _sprand_spmv(hp, wl, job, A, ncols, backend, args)
            hp.clear_plan_cache()
            return rec
        b_rows = rows_loc * mult
        Bl = torch.empty((b_rows, k), dtype=torch.float64, device=dev)
        hp._capi.call("hpcla_fill_uniform_f64", Bl.data_ptr(), rank * b_rows * k, b_rows * k, wl.SEED_X,
                      torch.cuda.current_stream().cuda_stream)
        B = hp.HPCMatrix_local(Bl, backend)
        setup_s = time.perf_counter() - t0
        order = getattr(args, "spmm_order", None)
        saved_order = os.environ.get("HPCLA_SPMM_ORDER")
        if order:
            os.environ["HPCLA_SPMM_ORDER"] = order          # every rank runs this record: the choice is collective
        regime = ("B = %d rows x 16 = %.2f GB: config 5's gather set, far beyond the 256 MiB Infinity Cache" % (ncols, ncols * 128 / 1e9)
                  if ncols * 128 > (1 << 29) else
                  "B = %d rows x 16 = %.0f MB: Infinity-Cache-sized B (MALL-assisted gathers), NOT config 5's regime" % (ncols, ncols * 128 / 1e6))
        out = _measure_spmm(args, job, hp, wl, A, B, k, world, dev, setup_s,
                            "SpMM GFLOP/s (2*k*nnz/t), sprand ~29.8 nnz/row, k=16, fp64",
                            f"sprand-like {rows_loc} rows per GPU x {ncols} cols, nnz/GPU={A.nnz}, k={k}, C = A*B; {regime}",
                            traffic_key=("sprand_spmm_b2e24" if rows_loc == 2_097_152 and ncols * 128 > (1 << 29) else
                                         "sprand_spmm_mall_sized" if (rows_loc, ncols) == (2_097_152, 2_097_152) else None),
                            share_note=("" if ncols in (16_777_216, 2_097_152) else
                                        f"the stored pass has B = 16777216 rows (the 8-GPU gather set), this run {ncols}: both far "
                                        "beyond the Infinity Cache, one 128-byte line per stored entry either way"))
        if world == 1 and not order and os.environ.get("HPCLA_BENCH_SPRAND_SPMV", "1") != "0":
            # the same matrix times one vector: the unstructured SpMV's current number
            try:
                out["spmv_same_matrix"] = _sprand_spmv(hp, wl, job, A, ncols, backend, args)
            except Exception as exc:
                out["spmv_same_matrix"] = {"error": f"{type(exc).__name__}: {exc}"}
        if order:
            out["config"]["workload"] += f"; HPCLA_SPMM_ORDER={order}"
            if saved_order is None:
                os.environ.pop("HPCLA_SPMM_ORDER", None)
            else:
                os.environ["HPCLA_SPMM_ORDER"] = saved_order
    _stamp("record measured")
    job.barrier()
    hp.clear_spmm_cache()
    hp.clear_plan_cache()
    torch.cuda.empty_cache()
    _stamp("plans destroyed, cache emptied")
    return out
