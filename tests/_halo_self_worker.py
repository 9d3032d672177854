"""GPU worker: drives hpcla_halo_* with a ONE-rank communicator that sends to itself
(HPCLA_FORCE_RCCL=1), so the ncclGroup send/recv, pack kernel, side stream and event ordering of the
multi-GPU path -- or, with HPCLA_HALO_MODE=push, the peer-window push kernel, flags, acks and the
in-kernel wait (the "peer" window being this rank's own) -- run on a single MI355X.  Also runs the fused
hpcla_spmv_dist with that plan.  Real two-process exchanges: tests/test_gpu_multirank.py."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import hpcla_amd as hp
    from oracle import oracle as orc

    assert os.environ.get("HPCLA_FORCE_RCCL") == "1"
    from hpcla_amd.backends import attach_halo_windows
    push = os.environ.get("HPCLA_HALO_MODE", "") == "push"
    backend = hp.backend_rocm_serial(np.float64, np.int32)
    assert backend.peer_windows
    capi = hp._capi
    lib = capi.load()
    s = torch.cuda.current_stream().cuda_stream
    n = 100_000
    x = torch.from_numpy(orc.fill_uniform(0, n, 42)).cuda()

    def run(idx_np, width, xsrc, bad_probe=False):
        plan = ctypes.c_void_p()
        idx = torch.from_numpy(idx_np.astype(np.int32)).cuda()
        ranks = (ctypes.c_int32 * 1)(0)
        counts = (ctypes.c_int64 * 1)(len(idx_np))
        torch.cuda.synchronize()
        capi.check("create", lib.hpcla_halo_plan_create(ctypes.byref(plan), backend.rccl, 1, ranks, counts,
                                                       idx.data_ptr(), 0, 1, ranks, counts, width))
        # the attach ends with the plan's connection test (two real exchanges checked against these lists); a
        # plan whose test fails is detached and exchanges over RCCL instead
        probe = (xsrc.numel() // width, width, [(0, idx_np + (1 if bad_probe else 0))])
        assert attach_halo_windows(backend, plan, probe) == (push and not bad_probe)
        ghost = ctypes.c_void_p(); ng = ctypes.c_int64()
        for rep in range(5):                       # repeated use: WAR ordering of the ghost buffer(s)
            xs = xsrc * (rep + 1.0)
            capi.call("hpcla_halo_begin", plan, xs.data_ptr(), s)
            busy = torch.ones(1 << 20, device="cuda").cumsum(0)      # overlapping work on the main stream
            capi.call("hpcla_halo_end", plan, s)
            capi.call("hpcla_halo_ghost_ptr", plan, ctypes.byref(ghost), ctypes.byref(ng))   # buffer of THIS exchange
            assert ng.value == len(idx_np)
            # read the ghost buffer on the main stream (ordered after halo_end)
            tmp = torch.empty(len(idx_np) * width, dtype=torch.float64, device="cuda")
            ident = torch.arange(len(idx_np) * width, dtype=torch.int64, device="cuda")
            capi.call("hpcla_gather_f64_i64", ghost, ident.data_ptr(), None, tmp.data_ptr(), ident.numel(), 0, s)
            torch.cuda.synchronize()
            want = xs.view(-1, width)[torch.from_numpy(idx_np).cuda()].reshape(-1)
            assert torch.equal(tmp, want), f"ghost mismatch (width={width}, rep={rep})"
        st = ctypes.c_int(0)
        capi.call("hpcla_halo_status", plan, ctypes.byref(st))
        assert st.value == 0, "push/wait timed out"
        capi.call("hpcla_halo_plan_destroy", plan)

    rng = np.random.default_rng(0)
    run(np.sort(rng.choice(n, size=5000, replace=False)), 1, x)          # scattered -> pack kernel
    run(np.arange(777, 777 + 8192), 1, x)                                 # contiguous -> direct send
    k = 16
    xk = torch.from_numpy(orc.fill_uniform(0, 4096 * k, 43)).cuda()
    run(np.sort(rng.choice(4096, size=300, replace=False)), k, xk)       # SpMM ghost rows, width 16
    run(np.arange(100, 400), k, xk)
    if push:
        run(np.arange(50, 900), 1, x, bad_probe=True)                      # failed connection test -> RCCL, same values

    # chained plans (the chunk-sets of a panel-ordered SpMM) share ONE exchange stream by reference count: destroying
    # the LEADER first must leave the follower fully usable (it used to borrow the leader's stream untracked)
    def make(idx_np, width, xsrc):
        plan = ctypes.c_void_p()
        idx = torch.from_numpy(idx_np.astype(np.int32)).cuda()
        ranks = (ctypes.c_int32 * 1)(0)
        counts = (ctypes.c_int64 * 1)(len(idx_np))
        torch.cuda.synchronize()
        capi.check("create_ex", lib.hpcla_halo_plan_create_ex(ctypes.byref(plan), backend.rccl, 1, ranks, counts, idx.data_ptr(),
                                                              0, 1, ranks, counts, width, capi.HALO_SINGLE_BUFFER))
        attach_halo_windows(backend, plan, (xsrc.numel() // width, width, [(0, idx_np)]))
        return plan, idx

    ia, ib = np.arange(10, 200), np.arange(300, 700)
    leader, keep_a = make(ia, k, xk)
    follower, keep_b = make(ib, k, xk)
    capi.call("hpcla_halo_plan_chain", follower, leader)
    capi.call("hpcla_halo_begin", leader, xk.data_ptr(), s)
    capi.call("hpcla_halo_begin", follower, xk.data_ptr(), s)
    capi.call("hpcla_halo_end", leader, s)
    capi.call("hpcla_halo_end", follower, s)
    torch.cuda.synchronize()
    capi.call("hpcla_halo_plan_destroy", leader)                       # leader FIRST
    ghost = ctypes.c_void_p(); ng = ctypes.c_int64()
    for rep in range(3):                                               # the follower still exchanges on the shared stream
        xs = xk * (rep + 2.0)
        capi.call("hpcla_halo_begin", follower, xs.data_ptr(), s)
        capi.call("hpcla_halo_end", follower, s)
        capi.call("hpcla_halo_ghost_ptr", follower, ctypes.byref(ghost), ctypes.byref(ng))
        tmp = torch.empty(len(ib) * k, dtype=torch.float64, device="cuda")
        ident = torch.arange(len(ib) * k, dtype=torch.int64, device="cuda")
        capi.call("hpcla_gather_f64_i64", ghost, ident.data_ptr(), None, tmp.data_ptr(), ident.numel(), 0, s)
        torch.cuda.synchronize()
        assert torch.equal(tmp, xs.view(-1, k)[torch.from_numpy(ib).cuda()].reshape(-1)), "follower after its leader was destroyed"
    capi.call("hpcla_halo_plan_destroy", follower)

    # all-reduce on the one-rank communicator is the identity
    t = torch.tensor([3.25, -1.0], dtype=torch.float64, device="cuda")
    capi.call("hpcla_allreduce_f64", backend.rccl, t.data_ptr(), 2, 0, s)
    capi.call("hpcla_allreduce_f64", backend.rccl, t.data_ptr(), 2, 1, s)
    torch.cuda.synchronize()
    assert t.tolist() == [3.25, -1.0]

    # fused distributed SpMV: a "periodic" 1-rank problem -- rank 0's ghosts are its own last rows.
    # rows: 2-D Poisson slab whose upper neighbour line is fetched through the halo from itself.
    nx, ny = 512, 40
    nloc = nx * ny
    rows = orc.poisson2d_rows(nx, ny + 1, 0, nloc)            # local rows of a taller grid
    ci, cv = orc.compress_columns(rows)                        # columns reach nloc .. nloc+nx-1 (ghosts)
    n_ghost = int((ci >= nloc).sum())
    assert n_ghost == nx
    xg = orc.fill_uniform(0, nloc, 44)
    # ghost g (global col nloc+g) is served by "neighbour" rank 0 from its local index g + 3*nx
    send_idx = np.arange(nx) + 3 * nx
    x_ext = np.concatenate([xg, xg[send_idx]])
    want = orc.spmv(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals, x_ext[ci])
    plan = ctypes.c_void_p()
    d_idx = torch.from_numpy(send_idx.astype(np.int32)).cuda()
    ranks = (ctypes.c_int32 * 1)(0)
    counts = (ctypes.c_int64 * 1)(nx)
    torch.cuda.synchronize()
    capi.check("create", lib.hpcla_halo_plan_create(ctypes.byref(plan), backend.rccl, 1, ranks, counts,
                                                   d_idx.data_ptr(), 0, 1, ranks, counts, 1))
    assert attach_halo_windows(backend, plan) == push
    d_rp = torch.from_numpy(rows.rowptr.astype(np.int32)).cuda()
    d_cv = torch.from_numpy(ci[cv].astype(np.int32)).cuda()    # split columns: own < nloc, ghosts nloc+g
    d_nz = torch.from_numpy(rows.vals).cuda()
    d_x = torch.from_numpy(xg).cuda()
    rpb = lib.hpcla_spmv_rows_per_block()
    nblk = (nloc + rpb - 1) // rpb
    flags = torch.empty(nblk, dtype=torch.int32, device="cuda")
    capi.call("hpcla_classify_blocks_i32", d_rp.data_ptr(), d_cv.data_ptr(), nloc, 0, nloc, rpb, flags.data_ptr(), s)
    interior = torch.nonzero(flags == 0).flatten().to(torch.int32)
    boundary = torch.nonzero(flags != 0).flatten().to(torch.int32)
    assert boundary.numel() == nx // rpb and interior.numel() == nblk - nx // rpb
    y = torch.full((nloc,), float("nan"), dtype=torch.float64, device="cuda")
    for rep in range(5):
        capi.call("hpcla_spmv_dist_f64_i32", plan, d_rp.data_ptr(), d_cv.data_ptr(), d_nz.data_ptr(),
                  d_x.data_ptr(), nloc, y.data_ptr(), nloc, rows.nnz, 0, interior.data_ptr(), interior.numel(),
                  boundary.data_ptr(), boundary.numel(), s)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), want), "fused distributed SpMV mismatch"
    # same exchange with the opt-in packed copy for the interior row blocks (+ fused x.y)
    packed = ctypes.c_void_p()
    capi.check("packed_create", lib.hpcla_packed_create_i32(ctypes.byref(packed), d_rp.data_ptr(), d_cv.data_ptr(),
                                                          d_nz.data_ptr(), nloc, rows.nnz, nloc, 0,
                                                          interior.data_ptr(), interior.numel(), s))
    work = torch.empty(lib.hpcla_spmv_dot_work_bytes(nloc) // 8 + 1, dtype=torch.float64, device="cuda")
    dot_out = torch.zeros(1, dtype=torch.float64, device="cuda")
    y2 = torch.full((nloc,), float("nan"), dtype=torch.float64, device="cuda")
    for rep in range(2):
        capi.call("hpcla_spmv_dist_packed_f64_i32", plan, backend.rccl, packed, d_rp.data_ptr(), d_cv.data_ptr(),
                  d_nz.data_ptr(), d_x.data_ptr(), nloc, y2.data_ptr(), nloc, rows.nnz, 0, interior.data_ptr(),
                  interior.numel(), boundary.data_ptr(), boundary.numel(), dot_out.data_ptr(), work.data_ptr(), s)
    torch.cuda.synchronize()
    assert np.array_equal(y2.cpu().numpy(), want), "packed distributed SpMV mismatch"
    ref_dot = float(np.dot(xg, want))
    assert abs(dot_out.item() - ref_dot) <= 1e-12 * float(np.abs(xg) @ np.abs(want)), (dot_out.item(), ref_dot)
    capi.call("hpcla_packed_destroy", packed)
    capi.call("hpcla_halo_plan_destroy", plan)
    # contiguous-range exchange (repartition plans): two messages to myself + the part that stays,
    # widths 1 (vector / nzval) and 5 (row-major dense rows)
    from hpcla_amd.repartition import exchange_ranges
    for width in (1, 5):
        src = torch.from_numpy(orc.fill_uniform(0, 3000 * width, 77)).cuda()
        dst = torch.full((2600 * width,), float("nan"), dtype=torch.float64, device="cuda")
        exchange_ranges(backend, src, dst, [0, 0], [100, 2000], [900, 1000], [0, 0], [0, 1600], [900, 1000],
                        1000, 900, 700, width)
        torch.cuda.synchronize()
        sv, dv = src.view(-1, width), dst.view(-1, width)
        assert torch.equal(dv[0:900], sv[100:1000]) and torch.equal(dv[900:1600], sv[1000:1700]) \
            and torch.equal(dv[1600:2600], sv[2000:3000]), f"exchange_ranges mismatch (width={width})"

    if push:
        # bounded spins: a wait whose producer never comes must give up after HPCLA_PUSH_TIMEOUT_S, flag the plan and
        # let the grid drain (the test sets 2 s) -- never hang the GPU
        import time
        plan = ctypes.c_void_p()
        idx = torch.arange(64, dtype=torch.int32, device="cuda")
        ranks = (ctypes.c_int32 * 1)(0)
        counts = (ctypes.c_int64 * 1)(64)
        torch.cuda.synchronize()
        capi.check("create", lib.hpcla_halo_plan_create(ctypes.byref(plan), backend.rccl, 1, ranks, counts,
                                                       idx.data_ptr(), 0, 1, ranks, counts, 1))
        assert attach_halo_windows(backend, plan)
        t0 = time.perf_counter()
        capi.call("hpcla_halo_end", plan, s)                 # a wait with no push before it
        torch.cuda.synchronize()
        waited = time.perf_counter() - t0
        st = ctypes.c_int(0)
        capi.call("hpcla_halo_status", plan, ctypes.byref(st))
        limit = float(os.environ.get("HPCLA_PUSH_TIMEOUT_S", "300"))
        assert st.value == 1, "the orphan wait did not report a timeout"
        assert 0.5 * limit <= waited <= limit + 5.0, f"orphan wait took {waited:.2f} s (limit {limit} s)"
        # ... and the expired wait POISONED the ghost buffer of its epoch: whatever is launched behind it computes NaN,
        # never a plausible result from stale values.  (This plan is a double-buffered vector plan; the orphan wait
        # was for epoch 1, i.e. buffer 1 = 64 doubles behind buffer 0 -- 64 * 8 B is already a multiple of the
        # window's 256-byte buffer alignment.  No push ever ran, so the step counter still says 0 and
        # hpcla_halo_ghost_ptr names buffer 0.)
        gp, gn = ctypes.c_void_p(), ctypes.c_int64(0)
        capi.call("hpcla_halo_ghost_ptr", plan, ctypes.byref(gp), ctypes.byref(gn))
        assert gn.value == 64
        both = torch.zeros(128, dtype=torch.float64, device="cuda")
        capi.call("hpcla_scale_f64", 1.0, gp, ctypes.c_void_p(both.data_ptr()), 128, s)     # copy out through the library
        torch.cuda.synchronize()
        assert bool(torch.isnan(both[64:]).all()), "the expired wait left stale values in the ghost buffer of its epoch"
        assert not bool(torch.isnan(both[:64]).any()), "the other buffer was touched"
        capi.call("hpcla_halo_plan_destroy", plan)
        print(f"orphan wait gave up after {waited:.2f} s, status flagged, ghost poisoned")

    print("halo self-exchange OK")


if __name__ == "__main__":
    main()
