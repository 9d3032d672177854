"""One-off stress of the push transport's epoch / ack / double-buffer protocol: thousands of dependent steps
x <- A x / 8 with a window all-reduce (norm) every step and NO host synchronisation, on N ranks (sharing the GPU if
there are fewer GPUs), checked at the end against the oracle's recurrence.
Run:  python -c "import hpcla_launch..."  -- see benchmarks/EXPERIMENTS.md"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    import hpcla_amd as hp
    from oracle import oracle as orc
    dist.init_process_group("gloo")
    rank, nranks = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
    steps = int(os.environ.get("STRESS_STEPS", "3000"))
    backend = hp.backend_rocm_mpi(np.float64, np.int32)
    # scale: keeps the recurrence from overflowing (row sums: 8 for the stencil, ~12 for the random matrix)
    for name, gen, n, scale in (("poisson2d", lambda lo, hi: orc.poisson2d_rows(256, 5 * nranks + 2, lo, hi), 256 * (5 * nranks + 2), 0.125),
                                ("sprand", lambda lo, hi: orc.sprand_rows(6000, 0.004, lo, hi), 6000, 0.078125)):
        if os.environ.get("STRESS_ONLY", name) != name:
            continue
        rp = orc.uniform_partition(n, nranks)
        lo, hi = int(rp[rank]), int(rp[rank + 1])
        rows = gen(lo, hi)
        import time
        t_case = time.perf_counter()
        A = hp.HPCSparseMatrix_local(rows.rowptr, rows.colidx, rows.vals, n, backend)
        xg = orc.fill_uniform(0, n, orc.SEED_X)
        xs = hp.HPCVector.from_global(xg, backend, partition=rp)
        ys = xs.similar()
        nrm = torch.zeros(steps, dtype=torch.float64, device="cuda")
        for k in range(steps):
            hp.mul_(ys, A, xs)
            xs.v.copy_(ys.v)
            xs.v.mul_(scale)
            hp.norm(xs, 2, out=nrm[k:k + 1])          # window all-reduce, result stays on the device
            if rank == 0 and k % 250 == 249:
                print(f"  {name}: {k + 1} steps enqueued", flush=True)
        torch.cuda.synchronize()
        assert not hp.get_vector_plan(A, xs).timed_out(), "timed out"
        rows_all = gen(0, n)
        ci, cv = orc.compress_columns(rows_all)
        xr = xg.copy()
        want_n = np.empty(steps)
        for k in range(steps):
            xr = orc.spmv(rows_all.rowptr.astype(np.int32), cv.astype(np.int32), rows_all.vals, xr[ci]) * scale
            want_n[k] = float(xr @ xr)
        assert np.array_equal(xs.local_values(), xr[lo:hi]), f"rank {rank} {name}: x differs after {steps} steps"
        got_n = nrm.cpu().numpy()
        rel = np.abs(got_n - want_n) / np.maximum(want_n, 1e-300)
        assert np.all((rel <= 1e-12) | (want_n == 0)), f"rank {rank} {name}: norms differ, max rel {rel.max()}"
        print(f"rank {rank}/{nranks} {name}: {steps} dependent steps + {steps} all-reduces OK "
              f"({time.perf_counter() - t_case:.1f} s)", flush=True)
    hp.clear_plan_cache()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
