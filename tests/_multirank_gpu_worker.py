"""Worker for tests/test_gpu_multirank.py: ONE process per rank (started by hpcla_amd launch.spawn_ranks),
every rank on GPU ``LOCAL_RANK % device_count`` -- with fewer GPUs than ranks the ranks SHARE a GPU, which
the peer-window transport supports (RCCL does not: "Duplicate GPU detected").

Checks, each against the CPU oracle on identical inputs and therefore against the 1-rank result too
(the 1-rank GPU product is bit-equal to the oracle, tests/test_gpu_parity.py):
  * y = A*x distributed, bit-exact, for a 2-D stencil slab (contiguous sends), a 3-D slab whose halo planes
    take several push chunks each, an unstructured matrix (scattered sends, all-to-all neighbour sets), a
    one-directional band (each rank receives from the next and sends to the previous one only: push targets and
    wait sources differ), a matrix with fewer rows than ranks (an empty rank) and x partitioned differently
    from A's rows;
  * 16 dependent steps x <- A*x/8 (free-running: exercises epochs, double buffering and acks);
  * dot / norm: 1e-12 relative to the oracle AND bit-identical on all ranks;
  * CG, 8 iterations, fused and unfused, residual history vs the oracle -- eagerly and replayed from a captured
    HIP graph (bit-identical);
  * A*B with k = 16 and k = 3 dense columns (distributed SpMM), Int32 and Int64;
  * mul_dot_ (fused SpMV + p.Ap);
  * products of all the matrices interleaved (several cached plans in flight, no host sync in between);
  * no push/wait timed out.
Exit code 0 = all passed on this rank."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    import hpcla_amd as hp
    from hpcla_amd import backends as B
    from oracle import oracle as orc

    dist.init_process_group("gloo")
    rank, nranks = dist.get_rank(), dist.get_world_size()
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % ndev)
    mode = os.environ.get("HPCLA_HALO_MODE", "(default)")

    def allgather_f64(v):
        flat = B.comm_allgather(comm, np.ascontiguousarray(v, dtype=np.float64).view(np.int64))
        return flat.view(np.float64)

    if os.environ.get("HPCLA_MR_TIMEOUT_CASE", "") == "1":
        return timeout_case(torch, dist, hp, orc, rank, nranks)

    # HPCLA_MR_TYPES=i32,i64 (default both): the test file gives the larger rank counts one index type each --
    # ranks that share a GPU time-slice it, so the suite's wall time grows with ranks x cases
    # "i64": Int64 matrices on NARROWED plans (the default: Int32 kernels); "i64wide": HPCLA_NARROW_INDICES=0, the
    # Int64 kernels themselves
    types = [t for t in os.environ.get("HPCLA_MR_TYPES", "i32,i64").split(",") if t]
    for tname in types:
        assert tname in ("i32", "i64", "i64wide"), tname
        Ti = np.int32 if tname == "i32" else np.int64
        os.environ["HPCLA_NARROW_INDICES"] = "0" if tname == "i64wide" else "1"
        backend = hp.backend_rocm_mpi(np.float64, Ti)
        comm = backend.comm
        tag = f"[rank {rank}/{nranks} {tname} mode={mode} windows={backend.peer_windows}]"

        cases = []
        # stencil slab: rows partitioned uniformly, 2 boundary lines per interior rank
        nx, ny = 512, 6 * nranks + 3
        n = nx * ny
        cases.append(("poisson2d", n, lambda lo, hi: orc.poisson2d_rows(nx, ny, lo, hi),
                      orc.uniform_partition(n, nranks), orc.uniform_partition(n, nranks)))
        # 3-D slab: one 96 x 96 plane (9216 doubles = 72 KiB) per neighbour, i.e. a push of SEVERAL 32 KiB
        # chunks per neighbour (arrival counter, last chunk publishes) -- config 4's shape in small
        mx, my, mz = 96, 96, 5 * nranks + 1
        n3d = mx * my * mz
        cases.append(("poisson3d", n3d, lambda lo, hi: orc.poisson3d_rows(mx, my, mz, lo, hi),
                      orc.uniform_partition(n3d, nranks), orc.uniform_partition(n3d, nranks)))
        # unstructured: every rank talks to every rank, scattered indices
        n2 = 30000
        cases.append(("sprand", n2, lambda lo, hi: orc.sprand_rows(n2, 0.0015, lo, hi),
                      orc.uniform_partition(n2, nranks), orc.uniform_partition(n2, nranks)))
        # one-directional dependencies: row i needs columns i and i + K with K ~ one rank's rows, so rank r
        # RECEIVES from rank r+1 only and SENDS to rank r-1 only -- push targets and wait sources differ, the
        # last rank only sends, the first only receives (the ack lines, not a symmetric handshake, protect the
        # ghost buffers here)
        n4 = 4000 * nranks
        K4 = 4000 - 37

        def upper_rows(lo, hi, n4=n4, K4=K4):
            import scipy.sparse as sp
            i = np.arange(n4)
            M = sp.csr_matrix((np.concatenate([2.0 + (i % 7), -1.0 - (i[:n4 - K4] % 3)]),
                               (np.concatenate([i, i[:n4 - K4]]), np.concatenate([i, i[:n4 - K4] + K4]))), shape=(n4, n4))
            M.sort_indices()
            loc = M[lo:hi]
            return orc.LocalRows(loc.indptr.astype(np.int64), loc.indices.astype(np.int64), loc.data.astype(np.float64), n4)
        cases.append(("upper", n4, upper_rows, orc.uniform_partition(n4, nranks), orc.uniform_partition(n4, nranks)))
        # fewer rows than ranks: the last rank owns nothing (empty local matrix, no plan of its own, still collective)
        n5 = max(nranks - 1, 1)

        def tiny_rows(lo, hi, n5=n5):
            import scipy.sparse as sp
            M = (sp.identity(n5, format="csr") * 3.0 + sp.diags([np.ones(max(n5 - 1, 0))], [1], shape=(n5, n5), format="csr")).tocsr()
            M.sort_indices()
            loc = M[lo:hi]
            return orc.LocalRows(loc.indptr.astype(np.int64), loc.indices.astype(np.int64), loc.data.astype(np.float64), n5)
        cases.append(("tiny", n5, tiny_rows, orc.uniform_partition(n5, nranks), orc.uniform_partition(n5, nranks)))
        # x partitioned differently from the rows (A*x accepts any partition of x, src/sparse.jl:2096-2128)
        n3 = 20000
        xp3 = np.array([0] + [min(n3, 700 + (n3 * r) // nranks) for r in range(1, nranks)] + [n3])
        cases.append(("sprand_xpart", n3, lambda lo, hi: orc.sprand_rows(n3, 0.002, lo, hi),
                      orc.uniform_partition(n3, nranks), xp3))

        # the pin at the reference's own sizes (tests/golden/pin_large.npz): laplacian_2d_sparse(10^4) and the
        # generate_sparse(1000)-shaped matrix; x = u01(SEED_X) is the x of every case here, so `got` is also held to
        # the fixture's exact-rational product below
        with np.load(os.path.join(ROOT, "tests", "golden", "pin_large.npz"), allow_pickle=False) as z:
            pin = {k: z[k] for k in z.files}
        for which in ("lap", "gs"):
            npin = int(pin[f"{which}_n"])
            cases.append((f"pin_{which}", npin,
                          lambda lo, hi, w=which, m=npin: orc.rows_from_coo(pin[f"{w}_I"], pin[f"{w}_J"], pin[f"{w}_V"], m, m, lo, hi),
                          orc.uniform_partition(npin, nranks), orc.uniform_partition(npin, nranks)))

        # HPCLA_MR_CASES=name,name (default: all): the runs with the larger rank counts / the panel-order run take a subset -- ranks
        # that share a GPU time-slice it, and the whole suite has to stay inside the driver's limit.  HPCLA_MR_PARTS names the
        # parts of a case to run (default all): steps (dependent steps), reductions, cg, spmm
        only = [c for c in os.environ.get("HPCLA_MR_CASES", "").split(",") if c]
        if only:
            assert set(only) <= {c[0] for c in cases}, only
            cases = [c for c in cases if c[0] in only]
        parts = set(p for p in os.environ.get("HPCLA_MR_PARTS", "steps,reductions,cg,spmm").split(",") if p)
        assert parts <= {"steps", "reductions", "cg", "spmm"}, parts
        kept = []                                   # (name, A, x, y, want): for the interleaved-plans check below
        for name, ng, gen, rp, xp in cases:
            lo, hi = int(rp[rank]), int(rp[rank + 1])
            rows = gen(lo, hi)
            A = hp.HPCSparseMatrix_local(rows.rowptr, rows.colidx, rows.vals, ng, backend)
            assert np.array_equal(A.row_partition, rp)
            xg = orc.fill_uniform(0, ng, orc.SEED_X)
            x = hp.HPCVector.from_global(xg, backend, partition=xp)
            ci, cv = orc.compress_columns(rows)
            want = orc.spmv(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals, xg[ci])
            y = A @ x
            torch.cuda.synchronize()
            got = y.local_values()
            assert np.array_equal(got, want), f"{tag} {name}: A*x differs, max err {np.abs(got - want).max()}"
            if name.startswith("pin_"):             # ... and against the exact-rational product of the fixture
                which = name[4:]
                assert np.array_equal(xg, pin[f"{which}_x_u01"])
                kk = np.diff(rows.rowptr).astype(np.float64)
                absAx = orc.abs_spmv(rows.rowptr, rows.colidx, rows.vals, xg)
                err = np.abs(got - pin[f"{which}_y_u01"][lo:hi])
                u = 2.0 ** -53
                assert np.all(err <= (kk * u / (1.0 - kk * u) + u) * absAx) and np.all(err <= 1e-12 * absAx), (tag, name)
            plan = hp.get_vector_plan(A, x)
            assert plan.narrowed == (tname == "i64") and plan.is_i64 == (tname == "i64wide"), (tag, name)
            assert plan.colval_split.dtype == (torch.int64 if tname == "i64wide" else torch.int32), (tag, name)
            if nranks > 1 and mode in ("(default)", "push") and backend.peer_windows:
                assert plan.push or not plan.has_halo, f"{tag} {name}: push transport not attached"
            # repeated in-place products (same x): ghost buffers are reused / double-buffered
            for _ in range(5):
                hp.mul_(y, A, x)
            torch.cuda.synchronize()
            assert np.array_equal(y.local_values(), want), f"{tag} {name}: repeated mul! differs"
            assert not plan.timed_out(), f"{tag} {name}: a push/wait timed out"
            kept.append((name, A, x, y, want))

            if name in ("sprand_xpart", "tiny"):
                continue
            # dependent steps: x_{k+1} = A x_k / 8, no host sync in between
            xs = hp.HPCVector.from_global(xg, backend, partition=rp)
            ys = xs.similar()
            steps = 16 if "steps" in parts else 0
            for _ in range(steps):
                hp.mul_(ys, A, xs)
                xs.v.copy_(ys.v)
                xs.v.mul_(0.125)
            torch.cuda.synchronize()
            assert not hp.get_vector_plan(A, xs).timed_out(), f"{tag} {name}: timed out in the dependent loop"
            # oracle: the same recurrence on the GLOBAL vector (every rank can afford it at this size)
            rows_all = gen(0, ng)
            ci_all, cv_all = orc.compress_columns(rows_all)
            xr = xg.copy()
            for _ in range(steps):
                xr = orc.spmv(rows_all.rowptr.astype(Ti), cv_all.astype(Ti), rows_all.vals, xr[ci_all]) * 0.125
            assert np.array_equal(xs.local_values(), xr[lo:hi]), f"{tag} {name}: dependent steps differ"

            # dot / norm: tolerance vs the oracle, identical bits across ranks
            yv = hp.HPCVector.from_global(orc.fill_uniform(0, ng, orc.SEED_RHS), backend, partition=rp)
            xv = hp.HPCVector.from_global(xg, backend, partition=rp)
            if "reductions" in parts:
                d = hp.dot(xv, yv)
                nr = hp.norm(xv)
                yg = orc.fill_uniform(0, ng, orc.SEED_RHS)
                d_ref = orc.dot([xg], [yg])
                n_ref = orc.norm([xg])
                assert abs(d - d_ref) <= 1e-12 * abs(d_ref), (tag, name, d, d_ref)
                assert abs(nr - n_ref) <= 1e-12 * abs(n_ref), (tag, name, nr, n_ref)
                pg = 1.0 + 1e-3 * xg                               # prod: the all-reduce with op = product
                pr = hp.prod(hp.HPCVector.from_global(pg, backend, partition=rp))
                assert abs(pr - np.exp(np.sum(np.log(pg)))) <= 1e-9 * abs(pr), (tag, name, pr)
                alld = allgather_f64(np.array([d, nr]))
                assert all(alld[2 * r] == d and alld[2 * r + 1] == nr for r in range(nranks)), \
                    f"{tag} {name}: dot/norm not uniform across ranks: {alld}"

            if name in ("poisson2d", "poisson3d") and "cg" in parts:
                # CG (SPD matrix): 8 iterations, both forms, vs the oracle's restatement
                bg = orc.fill_uniform(0, ng, orc.SEED_RHS)
                b = hp.HPCVector.from_global(bg, backend, partition=rp)
                _, hist_ref = orc.cg(rows_all.rowptr.astype(Ti), cv_all.astype(Ti), rows_all.vals, bg, 8)
                for fused in (True, False):
                    xc, hist = hp.cg_fixed_iterations(A, b, 8, fused=fused)
                    assert np.allclose(hist, hist_ref, rtol=1e-12, atol=0), (tag, fused, hist, hist_ref)
                    if fused:
                        # all iterations enqueued by ONE library call (hpcla_cg_iterations_*, the default) vs one
                        # call per kernel from Python: same launches, same arguments, same bits
                        xp_, hist_p = hp.cg_fixed_iterations(A, b, 8, fused=True, native_loop=False)
                        assert hist_p == hist, (tag, "python-loop CG differs from the native loop", hist_p, hist)
                        assert np.array_equal(xp_.local_values(), xc.local_values())
                        xc, hist = hp.cg_fixed_iterations(A, b, 8, fused=True)      # (xp_ shares no storage with xc)
                    # the same iterations replayed from a captured HIP graph: the push-mode step keeps its epoch
                    # in device memory, so a distributed step is capturable; bit-identical to the eager loop
                    xg2, hist_g = hp.cg_fixed_iterations(A, b, 8, fused=fused, graph=True)
                    assert hist_g == hist, (tag, fused, "graph replay differs", hist_g, hist)
                    assert np.array_equal(xg2.local_values(), xc.local_values())
                    assert not hp.get_vector_plan(A, b).timed_out()
                # fused SpMV + p.Ap
                out = torch.zeros(1, dtype=torch.float64, device="cuda")
                yy = xv.similar()
                hp.mul_dot_(yy, A, xv, out)
                torch.cuda.synchronize()
                y_all = orc.spmv(rows_all.rowptr.astype(Ti), cv_all.astype(Ti), rows_all.vals, xg[ci_all])
                assert np.array_equal(yy.local_values(), y_all[lo:hi])
                pAp_ref = orc.dot([xg], [y_all])
                assert abs(out.item() - pAp_ref) <= 1e-12 * float(np.abs(xg) @ np.abs(y_all)), (out.item(), pAp_ref)

            # distributed SpMM: k dense columns, row-major on the device.  k = 1: a width-1 exchange plan driven
            # through halo_begin / halo_end -- it must be SINGLE-buffered (round-2 defect: the ghost pointer was
            # taken before the exchange completed and named the PREVIOUS exchange's buffer); two DIFFERENT B in a
            # row, without a host sync in between, so that a stale buffer cannot pass
            for k in ((16, 3, 1) if "spmm" in parts else ()):
                Bg = orc.fill_uniform(0, ng * k, 4711).reshape(ng, k)
                Bl = torch.from_numpy(np.ascontiguousarray(Bg[lo:hi])).cuda()
                Bm = hp.HPCMatrix_local(Bl, backend)
                Bg2 = orc.fill_uniform(0, ng * k, 1234).reshape(ng, k) - 0.5
                Bm2 = hp.HPCMatrix_local(torch.from_numpy(np.ascontiguousarray(Bg2[lo:hi])).cuda(), backend)
                C = A @ Bm
                C2 = A @ Bm2                                     # cached plan, second exchange, different values
                C3 = A @ Bm
                torch.cuda.synchronize()
                Cw = orc.spmm(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals, np.ascontiguousarray(Bg[ci]))
                Cw2 = orc.spmm(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals, np.ascontiguousarray(Bg2[ci]))
                assert np.array_equal(C.A.cpu().numpy(), Cw), f"{tag} {name}: A*B (k={k}) differs"
                assert np.array_equal(C2.A.cpu().numpy(), Cw2), f"{tag} {name}: second A*B (k={k}, other B) differs"
                assert np.array_equal(C3.A.cpu().numpy(), Cw), f"{tag} {name}: third A*B (k={k}) differs"
                if os.environ.get("HPCLA_SPMM_ORDER_TEST", "") == "1":
                    # opt-in panel order (exchange overlapped chunk by chunk): a different summation ORDER, so
                    # BASELINE's 1e-12 relative and the componentwise |A||B| bound of SURVEY 8d instead of bits
                    os.environ["HPCLA_SPMM_ORDER"] = "panel"
                    try:
                        Cp = (A @ Bm2).A.cpu().numpy()
                        Cp1 = (A @ Bm).A.cpu().numpy()
                    finally:
                        os.environ.pop("HPCLA_SPMM_ORDER", None)
                    for got_p, want_p, Bref in ((Cp, Cw2, Bg2), (Cp1, Cw, Bg)):
                        bound = orc.spmm(rows.rowptr.astype(Ti), cv.astype(Ti), np.abs(rows.vals), np.ascontiguousarray(np.abs(Bref[ci])))
                        assert np.all(np.abs(got_p - want_p) <= 1e-12 * bound), f"{tag} {name}: panel-order A*B (k={k}) outside the |A||B| bound"
                        nrm = np.linalg.norm(want_p)
                        assert np.linalg.norm(got_p - want_p) <= 1e-12 * max(nrm, 1e-300), f"{tag} {name}: panel-order A*B (k={k}) norm error"
        # several cached plans in flight at once: products of DIFFERENT matrices interleaved without a host sync
        # (every plan has its own windows, epochs and acks; their pushes and waits must not disturb each other)
        for _ in range(4):
            for name, A, x, y, want in kept:
                y.v.zero_()
                hp.mul_(y, A, x)
        torch.cuda.synchronize()
        for name, A, x, y, want in kept:
            assert np.array_equal(y.local_values(), want), f"{tag} {name}: interleaved products differ"
            assert not hp.get_vector_plan(A, x).timed_out(), f"{tag} {name}: timed out in the interleaved loop"
        del kept
        flag = __import__("ctypes").c_int(0)
        hp._capi.call("hpcla_comm_status", backend.rccl, __import__("ctypes").byref(flag))
        assert flag.value == 0, f"{tag}: a window all-reduce timed out"
        hp.clear_spmm_cache()
        hp.clear_plan_cache()
        print(f"{tag} OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def timeout_case(torch, dist, hp, orc, rank, nranks):
    """An exchange whose neighbour never shows up (HPCLA_PUSH_TIMEOUT_S = 2): rank 0 takes one step more than the
    other ranks.  Its boundary workgroups give up after the bound and the step must be IMPOSSIBLE TO MISS: the rows
    that needed ghost values are NaN (never computed from stale ghosts), interior rows are right, the plan's status
    is set, and a scalar that reaches the host raises ExchangeTimeout.  The grid drains (the process exits)."""
    Ti = np.int32
    backend = hp.backend_rocm_mpi(np.float64, Ti)
    nx, ny = 512, 6 * nranks + 3
    n = nx * ny
    rp = orc.uniform_partition(n, nranks)
    lo, hi = int(rp[rank]), int(rp[rank + 1])
    rows = orc.poisson2d_rows(nx, ny, lo, hi)
    A = hp.HPCSparseMatrix_local(rows.rowptr, rows.colidx, rows.vals, n, backend)
    xg = orc.fill_uniform(0, n, orc.SEED_X)
    x = hp.HPCVector.from_global(xg, backend, partition=rp)
    ci, cv = orc.compress_columns(rows)
    want = orc.spmv(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals, xg[ci])
    y = A @ x                                        # a good step on every rank
    torch.cuda.synchronize()
    assert np.array_equal(y.local_values(), want)
    plan = hp.get_vector_plan(A, x)
    assert plan.push, "timeout case needs the push transport"
    dist.barrier()
    if rank == 0:
        y.v.zero_()
        hp.mul_(y, A, x)                             # nobody else takes this step
        torch.cuda.synchronize()                     # returns after ~HPCLA_PUSH_TIMEOUT_S: the grid drained
        got = y.local_values()
        bad = np.isnan(got)
        assert bad.any(), "a timed-out wait left no NaN in y"
        assert np.array_equal(got[~bad], want[~bad]), "rows that needed no ghost must still be right"
        ghost_rows = np.flatnonzero(np.diff(rows.rowptr) > 0)
        needs_ghost = np.array([np.any((rows.colidx[rows.rowptr[r]:rows.rowptr[r + 1]] < lo) |
                                       (rows.colidx[rows.rowptr[r]:rows.rowptr[r + 1]] >= hi)) for r in ghost_rows])
        assert bad[ghost_rows[needs_ghost]].all(), "a row that needs ghost values was computed from stale ghosts"
        assert plan.timed_out()
        # a DEAD plan must drain at once.  The status word is sticky and every wait reads it on its first
        # failed poll, so steps enqueued behind the expiry cost (almost) nothing -- before, each of them spun out the
        # full bound again (8 steps = 8 timeouts; 100 enqueued CG iterations = 300).
        import time
        bound = float(os.environ["HPCLA_PUSH_TIMEOUT_S"])
        t0 = time.perf_counter()
        for _ in range(8):
            hp.mul_(y, A, x)
        torch.cuda.synchronize()
        dt_steps = time.perf_counter() - t0
        assert dt_steps < 0.5 * bound, f"8 steps on a dead plan took {dt_steps:.2f} s (bound {bound} s): not draining at once"
        assert np.isnan(y.local_values()[ghost_rows[needs_ghost]]).all()
        # ... and the same for a whole CG call: 6 iterations = 6 halo waits + 12 all-reduces (+ 1 in the setup) enqueued
        # with no host in the loop.  The communicator's first all-reduce spins out ONE bound (nobody answers), sets the
        # communicator's sticky word, and everything behind it returns poison at once.
        ws = hp.CGWorkspace(x, 8)
        t0 = time.perf_counter()
        cplan, fused = hp.cg_setup(A, x, ws)
        hp.cg_iterate(A, ws, cplan, fused, 6)
        torch.cuda.synchronize()
        dt_cg = time.perf_counter() - t0
        assert dt_cg < 1.8 * bound, f"a 6-iteration CG call on a dead plan took {dt_cg:.2f} s (bound {bound} s)"
        assert np.isnan(ws.hist[1:7].cpu().numpy()).all(), "CG history behind an expired exchange must be NaN"
        print(f"[rank 0] dead plan drains: 8 steps {dt_steps * 1e3:.1f} ms, 6 CG iterations {dt_cg:.2f} s "
              f"(bound {bound} s)", flush=True)
        # every scalar that reaches the host from the poisoned vector raises -- the max reductions included (their
        # comparator used to drop NaN, so norm(y, Inf) / maximum(y) returned a number after a timed-out exchange)
        for what, fn in (("dot", lambda: hp.dot(y, y)), ("norm(y, Inf)", lambda: hp.norm(y, np.inf)),
                         ("maximum", lambda: hp.maximum(y)), ("minimum", lambda: hp.minimum(y))):
            try:
                fn()
            except hp.ExchangeTimeout as exc:
                print(f"[rank 0] {what} raised as it must: {exc}", flush=True)
            else:
                raise AssertionError(f"{what} of a poisoned vector returned without raising ExchangeTimeout")
    dist.barrier()
    print(f"[rank {rank}/{nranks}] timeout case OK", flush=True)
    # the plan is dead: no collective teardown of its windows (a dead plan's peers may not show up either)
    dist.destroy_process_group()
    os._exit(0)


if __name__ == "__main__":
    main()
