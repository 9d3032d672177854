#!/usr/bin/env python3
"""Overhead of the distributed step on ONE GPU: a 4096 x 4096 Poisson slab whose upper and lower
ghost lines (2 x 32 KiB) are exchanged with ITSELF through a one-rank communicator (HPCLA_FORCE_RCCL=1),
i.e. the launch sequence of an interior rank in bench.py --gpus N, in every ordering of the step
(hpcla_set_halo_mode): push (peer-window push kernel + ONE fused launch whose boundary workgroups wait
in-kernel), serial (RCCL group, then one launch) and overlap (RCCL group + boundary blocks on a side
stream), each against the plain kernel."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HPCLA_FORCE_RCCL"] = "1"


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--slabs":
        # round 5: the per-GPU shares of the strong-scaled problems -- e.g. 8192x1024 = config 3 at 8 GPUs --
        # one after the other: plain kernel, the step with the self-exchange in every ordering; one summary line each
        for spec in sys.argv[2].split(","):
            nx, ny = (int(v) for v in spec.lower().split("x"))
            run(nx, ny, False, summary=True)
        return
    run(4096, 4096, len(sys.argv) > 1 and sys.argv[1] == "--dim3")


def run(nx_arg, ny_arg, dim3, summary=False):
    import torch
    import hpcla_amd as hp
    from hpcla_amd import workloads as wl
    backend = hp.backend_rocm_serial(np.float64, np.int32)
    capi, lib = hp._capi, hp._capi.load()
    s = torch.cuda.current_stream().cuda_stream
    if dim3:
        # config 4's per-GPU share: 512 x 512 x 64 slab of the 7-point matrix, ghost PLANES of 262 144 values (2 MiB)
        N, planes = 512, 64
        nx = N * N                                           # "reach": one plane
        nloc = nx * planes
        rowptr, colidx, vals = wl.poisson3d_rows(N, N, 3 * planes, nloc, 2 * nloc)
        print(f"3-D slab {N}x{N}x{planes}: ghost planes of {nx} values ({nx * 8 / 2**20:.1f} MiB each)")
    else:
        nx, ny = nx_arg, ny_arg
        nloc = nx * ny
        # middle slab of a 3-slab grid: ghosts below (nx) and above (nx)
        rowptr, colidx, vals = wl.poisson2d_rows(nx, 3 * ny, nloc, 2 * nloc)
    colidx = colidx - nloc                                   # own columns 0..nloc-1, ghosts <0 and >= nloc
    split = np.where(colidx < 0, nloc + (colidx + nx), np.where(colidx >= nloc, nloc + nx + (colidx - nloc), colidx))
    d_rp = torch.from_numpy(rowptr.astype(np.int32)).cuda()
    d_cv = torch.from_numpy(split.astype(np.int32)).cuda()
    d_nz = torch.from_numpy(vals).cuda()
    x = torch.rand(nloc, dtype=torch.float64, device="cuda")
    y = torch.empty(nloc, dtype=torch.float64, device="cuda")
    rpb = lib.hpcla_spmv_rows_per_block()
    nblk = (nloc + rpb - 1) // rpb
    flags = torch.empty(nblk, dtype=torch.int32, device="cuda")
    capi.call("hpcla_classify_blocks_i32", d_rp.data_ptr(), d_cv.data_ptr(), nloc, 0, nloc, rpb, flags.data_ptr(), s)
    interior = torch.nonzero(flags == 0).flatten().to(torch.int32)
    boundary = torch.nonzero(flags != 0).flatten().to(torch.int32)
    # one "neighbour", rank 0 itself: lower ghost = my last line, upper ghost = my first line (a rank appears
    # once per list in a real plan; the two lines travel as one 64 KiB message here)
    send_idx = torch.cat([torch.arange(nloc - nx, nloc), torch.arange(0, nx)]).to(torch.int32).cuda()
    plan = ctypes.c_void_p()
    ranks = (ctypes.c_int32 * 1)(0)
    counts = (ctypes.c_int64 * 1)(2 * nx)
    torch.cuda.synchronize()
    capi.check("create", lib.hpcla_halo_plan_create(ctypes.byref(plan), backend.rccl, 1, ranks, counts,
                                                   send_idx.data_ptr(), 0, 1, ranks, counts, 1))
    from hpcla_amd.backends import attach_halo_windows
    pushable = attach_halo_windows(backend, plan)
    modes = (["push"] if pushable else []) + ["serial", "overlap"]

    def dist():
        capi.call("hpcla_spmv_dist_f64_i32", plan, d_rp.data_ptr(), d_cv.data_ptr(), d_nz.data_ptr(), x.data_ptr(),
                  nloc, y.data_ptr(), nloc, len(vals), 0, interior.data_ptr(), interior.numel(),
                  boundary.data_ptr(), boundary.numel(), s)

    ghost = torch.zeros(2 * nx, dtype=torch.float64, device="cuda")

    def plain():
        capi.call("hpcla_spmv_split_f64_i32", d_rp.data_ptr(), d_cv.data_ptr(), d_nz.data_ptr(), x.data_ptr(),
                  ghost.data_ptr(), nloc, y.data_ptr(), nloc, len(vals), 0, None, 0, s)

    def halo_only():
        capi.call("hpcla_halo_begin", plan, x.data_ptr(), s)
        capi.call("hpcla_halo_end", plan, s)

    def interior_only():
        capi.call("hpcla_spmv_split_f64_i32", d_rp.data_ptr(), d_cv.data_ptr(), d_nz.data_ptr(), x.data_ptr(),
                  ghost.data_ptr(), nloc, y.data_ptr(), nloc, len(vals), 0, interior.data_ptr(), interior.numel(), s)

    def boundary_only():
        capi.call("hpcla_spmv_split_f64_i32", d_rp.data_ptr(), d_cv.data_ptr(), d_nz.data_ptr(), x.data_ptr(),
                  ghost.data_ptr(), nloc, y.data_ptr(), nloc, len(vals), 0, boundary.data_ptr(), boundary.numel(), s)

    def interior_then_boundary():
        interior_only()
        boundary_only()

    import time

    def measure(name, fn):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        res = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 200 * 1e3)
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        host = (time.perf_counter() - t0) / 200 * 1e3
        torch.cuda.synchronize()
        print(f"{name:44s} {np.median(res):8.4f} ms/step (min {np.min(res):.4f})   host enqueue {host:.4f} ms/step", flush=True)
        return float(np.median(res))

    if summary:
        MODE = {"serial": 0, "overlap": 1, "push": 2}
        res = {}
        for rnd in range(2):
            res.setdefault("plain", []).append(measure(f"[{nx}x{ny}] plain split kernel", plain))
            for m in modes:
                capi.call("hpcla_set_halo_mode", MODE[m])
                res.setdefault(m, []).append(measure(f"[{nx}x{ny}] [{m}] halo + interior + boundary", dist))
        capi.call("hpcla_set_halo_mode", -1)
        med = {k: float(np.min(v)) for k, v in res.items()}
        print(f"SLAB {nx}x{ny} rows={nloc} plain_ms={med['plain']:.4f} " +
              " ".join(f"{m}_ms={med[m]:.4f} {m}_overhead_us={1e3 * (med[m] - med['plain']):+.1f}" for m in modes), flush=True)
        torch.cuda.synchronize()
        capi.call("hpcla_halo_plan_destroy", plan)
        return
    base = measure("plain split kernel", plain)
    measure("interior blocks only (list)", interior_only)
    measure("boundary blocks only", boundary_only)
    measure("interior + boundary, no halo", interior_then_boundary)
    MODE = {"serial": 0, "overlap": 1, "push": 2}
    for rnd in range(2):                                   # two interleaved rounds: box drift shows up as disagreement
        for m in modes:
            capi.call("hpcla_set_halo_mode", MODE[m])
            t = measure(f"[{m}] halo + interior + boundary", dist)
            measure(f"[{m}] halo exchange only", halo_only)
            print(f"    -> [{m}] overhead vs plain kernel: {1e3 * (t - base):+.1f} us", flush=True)
        base = measure("plain split kernel (again)", plain)
    capi.call("hpcla_set_halo_mode", -1)
    st = ctypes.c_int(0)
    capi.call("hpcla_halo_status", plan, ctypes.byref(st))
    print(f"interior blocks {interior.numel()}, boundary blocks {boundary.numel()}, timed_out={st.value}")
    torch.cuda.synchronize()
    capi.call("hpcla_halo_plan_destroy", plan)


if __name__ == "__main__":
    main()
