"""Worker for tests/test_float32.py::test_float32_backend_across_ranks: ONE process per rank (launch.spawn_ranks), ranks may
share a GPU (peer-window transport).  The Float32 backend (csrc/f32.hip) through the host layer, each check against the
oracle's Float32 loop on identical inputs -- hence against the 1-rank result too:
  * y = A*x and repeated mul!(y, A, x): bit-exact, for a 2-D slab, a 3-D slab (several push chunks per neighbour), an
    unstructured matrix (scattered sends -- the widening pack), an empty rank, x partitioned unlike A's rows;
  * 12 dependent steps x <- A*x/8 without a host sync (the staging vector and the single ghost buffer are reused);
  * dot / norm / sum: formed in double, 1e-6 relative to a double reference (the reference's Float32 tolerance is 1e-4)
    and bit-identical on all ranks;
  * A*B with 16, 3 and 1 dense columns, two different B back to back;
  * nothing timed out.
Exit code 0 = all passed on this rank."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
F32 = np.float32


def main():
    import torch
    import torch.distributed as dist
    import hpcla_amd as hp
    from hpcla_amd import backends as B
    from oracle import oracle as orc

    dist.init_process_group("gloo")
    rank, nranks = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
    # HPCLA_MR_TYPES=i32,i64 (default both): the test file gives each rank count one index type (ranks that share a GPU
    # time-slice it, so wall time grows with ranks x cases)
    types = [t for t in os.environ.get("HPCLA_MR_TYPES", "i32,i64").split(",") if t]
    for Ti in [np.int32 if t == "i32" else np.int64 for t in types]:
        backend = hp.backend_rocm_mpi(F32, Ti)
        comm = backend.comm
        tag = f"[f32 rank {rank}/{nranks} {np.dtype(Ti).name} windows={backend.peer_windows}]"
        cases = []
        nx, ny = 512, 6 * nranks + 3
        cases.append(("poisson2d", nx * ny, lambda lo, hi: orc.poisson2d_rows(nx, ny, lo, hi), None))
        mx, my, mz = 96, 96, 5 * nranks + 1
        cases.append(("poisson3d", mx * my * mz, lambda lo, hi: orc.poisson3d_rows(mx, my, mz, lo, hi), None))
        n2 = 30000
        cases.append(("sprand", n2, lambda lo, hi: orc.sprand_rows(n2, 0.0015, lo, hi), None))
        n5 = max(nranks - 1, 1)

        def tiny_rows(lo, hi, n5=n5):
            import scipy.sparse as sp
            M = (sp.identity(n5, format="csr") * 3.0 + sp.diags([np.ones(max(n5 - 1, 0))], [1], shape=(n5, n5), format="csr")).tocsr()
            M.sort_indices()
            loc = M[lo:hi]
            return orc.LocalRows(loc.indptr.astype(np.int64), loc.indices.astype(np.int64), loc.data.astype(np.float64), n5)
        cases.append(("tiny", n5, tiny_rows, None))
        n3 = 20000
        xp3 = np.array([0] + [min(n3, 700 + (n3 * r) // nranks) for r in range(1, nranks)] + [n3])
        cases.append(("sprand_xpart", n3, lambda lo, hi: orc.sprand_rows(n3, 0.002, lo, hi), xp3))

        for name, ng, gen, xp in cases:
            rp = orc.uniform_partition(ng, nranks)
            xp = rp if xp is None else xp
            lo, hi = int(rp[rank]), int(rp[rank + 1])
            rows = gen(lo, hi)
            vals32 = rows.vals.astype(F32)
            A = hp.HPCSparseMatrix_local(rows.rowptr, rows.colidx, rows.vals, ng, backend)
            assert A.nzval.dtype == torch.float32
            xg = orc.fill_uniform(0, ng, orc.SEED_X).astype(F32)
            x = hp.HPCVector.from_global(xg, backend, partition=xp)
            assert x.v.dtype == torch.float32
            ci, cv = orc.compress_columns(rows)
            want = orc.spmv(rows.rowptr.astype(Ti), cv.astype(Ti), vals32, xg[ci])
            y = A @ x
            torch.cuda.synchronize()
            assert y.v.dtype == torch.float32 and y.backend is backend
            got = y.local_values()
            assert np.array_equal(got, want), f"{tag} {name}: A*x differs, max err {np.abs(got - want).max() if len(want) else 0}"
            plan = hp.get_vector_plan(A, x)
            assert plan.is_f32
            for _ in range(4):
                hp.mul_(y, A, x)
            torch.cuda.synchronize()
            assert np.array_equal(y.local_values(), want), f"{tag} {name}: repeated mul! differs"
            assert not plan.timed_out(), f"{tag} {name}: a push/wait timed out"
            if name in ("sprand_xpart", "tiny"):
                continue
            xs = hp.HPCVector.from_global(xg, backend, partition=rp)
            ys = xs.similar()
            steps = 12
            for _ in range(steps):
                hp.mul_(ys, A, xs)
                xs.v.copy_(ys.v)
                xs.v.mul_(0.125)
            torch.cuda.synchronize()
            rows_all = gen(0, ng)
            ci_all, cv_all = orc.compress_columns(rows_all)
            xr = xg.copy()
            for _ in range(steps):
                xr = orc.spmv(rows_all.rowptr.astype(Ti), cv_all.astype(Ti), rows_all.vals.astype(F32), xr[ci_all]) * F32(0.125)
            assert xr.dtype == F32
            assert np.array_equal(xs.local_values(), xr[lo:hi]), f"{tag} {name}: dependent steps differ"

            yg = orc.fill_uniform(0, ng, orc.SEED_RHS).astype(F32)
            xv = hp.HPCVector.from_global(xg, backend, partition=rp)
            yv = hp.HPCVector.from_global(yg, backend, partition=rp)
            d, nr, sm = hp.dot(xv, yv), hp.norm(xv), hp.vsum(xv)
            x64, y64 = xg.astype(np.float64), yg.astype(np.float64)
            for gotv, ref in ((d, float(x64 @ y64)), (nr, float(np.sqrt(x64 @ x64))), (sm, float(x64.sum()))):
                assert gotv == float(F32(gotv)), (tag, name, "not rounded to Float32", gotv)
                assert abs(gotv - ref) <= 1e-6 * abs(ref), (tag, name, gotv, ref)
            assert hp.norm(xv, np.inf) == float(np.abs(xg).max()) and hp.maximum(xv) == float(xg.max()) and hp.minimum(xv) == float(xg.min())
            allv = B.comm_allgather(comm, np.array([d, nr, sm], dtype=np.float64).view(np.int64)).view(np.float64)
            assert all(np.array_equal(allv[3 * r:3 * r + 3], [d, nr, sm]) for r in range(nranks)), f"{tag} {name}: reductions not uniform"
            # u + v, u - v, a*v, v/a: numpy's float32 arithmetic, bit for bit
            assert np.array_equal((xv + yv).local_values(), xg[lo:hi] + yg[lo:hi])
            assert np.array_equal((xv - yv).local_values(), xg[lo:hi] - yg[lo:hi])
            assert np.array_equal((xv * 1.7).local_values(), F32(1.7) * xg[lo:hi])
            assert np.array_equal((xv / 1.7).local_values(), xg[lo:hi] / F32(1.7))

            for k in (16, 3, 1):
                Bg = orc.fill_uniform(0, ng * k, 4711).reshape(ng, k).astype(F32)
                Bg2 = (orc.fill_uniform(0, ng * k, 1234).reshape(ng, k) - 0.5).astype(F32)
                Bm = hp.HPCMatrix_local(torch.from_numpy(np.ascontiguousarray(Bg[lo:hi])).cuda(), backend)
                Bm2 = hp.HPCMatrix_local(torch.from_numpy(np.ascontiguousarray(Bg2[lo:hi])).cuda(), backend)
                C, C2, C3 = A @ Bm, A @ Bm2, A @ Bm
                torch.cuda.synchronize()
                assert C.A.dtype == torch.float32
                Cw = orc.spmm(rows.rowptr.astype(Ti), cv.astype(Ti), vals32, np.ascontiguousarray(Bg[ci]))
                Cw2 = orc.spmm(rows.rowptr.astype(Ti), cv.astype(Ti), vals32, np.ascontiguousarray(Bg2[ci]))
                assert np.array_equal(C.A.cpu().numpy(), Cw), f"{tag} {name}: A*B (k={k}) differs"
                assert np.array_equal(C2.A.cpu().numpy(), Cw2), f"{tag} {name}: second A*B (k={k}) differs"
                assert np.array_equal(C3.A.cpu().numpy(), Cw), f"{tag} {name}: third A*B (k={k}) differs"
        # the widened rows stay Float64 entries: a clear error, not a wrong answer
        try:
            hp.transpose(A) @ x
            raise AssertionError("transpose(A)*x accepted a Float32 backend")
        except TypeError:
            pass
        hp.check_exchange_health(backend, always=True)
        print(f"{tag} ok", flush=True)
    hp.clear_plan_cache()
    hp.clear_spmm_cache()
    dist.barrier()
    return 0


if __name__ == "__main__":
    sys.exit(main())
