"""Worker for tests/test_gpu_colmajor.py::test_colmajor_product_across_ranks: ONE process per rank, ranks may share a GPU.
The distributed A * B of a column-major caller (what integration/HPCLinearAlgebraROCmExt.jl does for banded matrices), driven
through the raw C ABI on the host layer's plans: own block and result COLUMN-major, the exchange posted from the column-major
block (hpcla_halo_begin_strided_*: the rows the plan sends are staged row-major), interior 256-row blocks while it is in
flight, boundary blocks behind hpcla_halo_end with the plan's row-major ghost segment.  Float64 and Float32, Int32 and Int64
kernel indices, k = 16 and 3, two different B back to back; at k = 16 in Float64 also through the run tiles on the column-major
blocks (round 5); bit-exact against the oracle's column loop."""
import ctypes
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
    from hpcla_amd import dense
    from hpcla_amd.vectors import current_stream_ptr, dptr
    from oracle import oracle as orc
    capi = hp._capi

    dist.init_process_group("gloo")
    rank, nranks = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
    rpb = capi.load().hpcla_spmv_rows_per_block()
    for T, Ti in ((np.float64, np.int32), (np.float32, np.int64), (np.float64, np.int64)):
        if Ti == np.int64 and T == np.float64:
            os.environ["HPCLA_NARROW_INDICES"] = "0"        # the Int64 kernels themselves
        backend = hp.backend_rocm_mpi(T, Ti)
        dt = "f64" if T == np.float64 else "f32"
        tT = torch.float64 if T == np.float64 else torch.float32
        tag = f"[colmajor rank {rank}/{nranks} {dt} {np.dtype(Ti).name}]"
        nx, ny = 512, 6 * nranks + 3
        cases = [("poisson2d", nx * ny, lambda lo, hi: orc.poisson2d_rows(nx, ny, lo, hi)),
                 ("sprand", 20000, lambda lo, hi: orc.sprand_rows(20000, 0.002, lo, hi))]
        for name, ng, gen in cases:
            rp_part = orc.uniform_partition(ng, nranks)
            lo, hi = int(rp_part[rank]), int(rp_part[rank + 1])
            nloc = hi - lo
            rows = gen(lo, hi)
            A = hp.HPCSparseMatrix_local(rows.rowptr, rows.colidx, rows.vals, ng, backend)
            ci, cv = orc.compress_columns(rows)
            for k in (16, 3):
                Bgs = [orc.fill_uniform(0, ng * k, seed).reshape(ng, k).astype(T) - T(0.25) for seed in (4711, 1234)]
                # the host layer's plan for (A, row partition of B, k): halo handle, split columns, ghost segment
                probe = hp.HPCMatrix_local(torch.from_numpy(np.ascontiguousarray(Bgs[0][lo:hi])).cuda(), backend)
                plan, ent = dense._spmm_plan(A, probe, width=k)      # (column-major blocks: the exchange carries exactly k values per row)
                assert ent is not None and ent[0] is not None, f"{tag} {name}: no exchange entry"
                halo, _i, _b, _s, colval_split, ghost = ent[:6]
                is64 = bool(ent[10])
                sfx = "i64" if is64 else "i32"
                rowptr = dense._entry_rowptr(A, plan, is64)
                # 256-row blocks: with / without ghost columns
                nblk = (A.nrows_local + rpb - 1) // rpb
                flags = torch.zeros(max(nblk, 1), dtype=torch.int32, device="cuda")
                s = current_stream_ptr()
                capi.call(f"hpcla_classify_blocks_{sfx}", dptr(rowptr), dptr(colval_split), A.nrows_local, 0, plan.n_own, rpb,
                          dptr(flags), s)
                interior = torch.nonzero(flags[:nblk] == 0).flatten().to(torch.int32).contiguous()
                boundary = torch.nonzero(flags[:nblk] != 0).flatten().to(torch.int32).contiguous()
                stage = torch.empty(max(plan.n_own * k, 1), dtype=torch.float64, device="cuda")
                ld = max(nloc, 1)
                fn = f"hpcla_spmm_split_colmajor_{dt}_{sfx}"
                outs = []
                for Bg in Bgs:                                # two different B without a host sync in between
                    Bc = torch.from_numpy(np.ascontiguousarray(Bg[lo:hi].T)).cuda()           # column-major nloc x k
                    C = torch.full((k, max(A.nrows_local, 1)), float("nan"), dtype=tT, device="cuda")
                    capi.call(f"hpcla_halo_begin_strided_{dt}", halo, dptr(Bc), 1, ld, dptr(stage), s)
                    if interior.numel():
                        capi.call(fn, dptr(rowptr), dptr(colval_split), dptr(A.nzval), dptr(Bc), ld, None, k, plan.n_own, dptr(C),
                                  max(A.nrows_local, 1), A.nrows_local, A.nnz, k, 0, dptr(interior), int(interior.numel()), s)
                    capi.call("hpcla_halo_end", halo, s)
                    if boundary.numel():
                        capi.call(fn, dptr(rowptr), dptr(colval_split), dptr(A.nzval), dptr(Bc), ld, ghost, k, plan.n_own, dptr(C),
                                  max(A.nrows_local, 1), A.nrows_local, A.nnz, k, 0, dptr(boundary), int(boundary.numel()), s)
                    outs.append((Bc, C))
                torch.cuda.synchronize()
                for Bg, (_, C) in zip(Bgs, outs):
                    want = orc.spmm(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals.astype(T), np.ascontiguousarray(Bg[ci]))
                    got = C.cpu().numpy()[:, :A.nrows_local].T
                    assert np.array_equal(got, want), f"{tag} {name} k={k}: column-major A*B differs"
                if T == np.float64 and k == 16:              # (a rank-uniform condition: the exchange below is collective)
                    # round 5: the same product through the RUN TILES on the column-major blocks (hpcla_spmm_runs_colmajor_k16_f64_*):
                    # descriptors of the plan's split column space, block lists over ITS 64-row blocks, interior blocks while
                    # the exchange is in flight (the first of them through the plan-time tuner), boundary blocks with the
                    # ghost segment.  The random pattern's blocks do not fit the tile and take the per-entry path.
                    rpb64 = capi.load().hpcla_spmm_rows_per_block()
                    nb64 = (A.nrows_local + rpb64 - 1) // rpb64
                    desc = torch.empty(capi.load().hpcla_spmm_runs_desc_bytes(A.nrows_local), dtype=torch.uint8, device="cuda")
                    n_fit = ctypes.c_int64(-1)
                    capi.call(f"hpcla_spmm_runs_build_{sfx}", dptr(rowptr), dptr(colval_split), A.nrows_local, A.nnz, 0, plan.n_own,
                              dptr(desc), ctypes.byref(n_fit), s)
                    if name == "poisson2d":
                        assert n_fit.value == nb64, f"{tag}: {n_fit.value} of {nb64} blocks fit"
                    flags64 = torch.zeros(max(nb64, 1), dtype=torch.int32, device="cuda")
                    capi.call(f"hpcla_classify_blocks_{sfx}", dptr(rowptr), dptr(colval_split), A.nrows_local, 0, plan.n_own, rpb64,
                              dptr(flags64), s)
                    int64_ = torch.nonzero(flags64[:nb64] == 0).flatten().to(torch.int32).contiguous()
                    bnd64 = torch.nonzero(flags64[:nb64] != 0).flatten().to(torch.int32).contiguous()
                    fnr = f"hpcla_spmm_runs_colmajor_k16_f64_{sfx}"
                    chosen = ctypes.c_int(-1)
                    ldp = max(nloc + (nloc & 1), 2)              # an EVEN leading dimension: one row of NaN padding where nloc is odd
                    for it, (Bg, (Bc0, _)) in enumerate(zip(Bgs, outs)):
                        Bc = torch.full((k, ldp), float("nan"), dtype=tT, device="cuda")
                        Bc[:, :nloc] = Bc0
                        assert Bc.data_ptr() % 16 == 0
                        C = torch.full((k, max(A.nrows_local, 1)), float("nan"), dtype=tT, device="cuda")
                        capi.call("hpcla_halo_begin_strided_f64", halo, dptr(Bc), 1, ldp, dptr(stage), s)
                        if int64_.numel() and it == 0:
                            capi.call(f"hpcla_spmm_runs_colmajor_tune_block_order_f64_{sfx}", dptr(rowptr), dptr(colval_split), dptr(A.nzval),
                                      dptr(Bc), ldp, None, k, plan.n_own, dptr(C), max(A.nrows_local, 1), A.nrows_local, A.nnz, 0, dptr(desc),
                                      dptr(int64_), int(int64_.numel()), s, ctypes.byref(chosen))
                        elif int64_.numel():
                            capi.call(fnr, dptr(rowptr), dptr(colval_split), dptr(A.nzval), dptr(Bc), ldp, None, k, plan.n_own, dptr(C),
                                      max(A.nrows_local, 1), A.nrows_local, A.nnz, 0, dptr(desc), dptr(int64_), int(int64_.numel()), s)
                        capi.call("hpcla_halo_end", halo, s)
                        if bnd64.numel():
                            capi.call(fnr, dptr(rowptr), dptr(colval_split), dptr(A.nzval), dptr(Bc), ldp, ghost, k, plan.n_own, dptr(C),
                                      max(A.nrows_local, 1), A.nrows_local, A.nnz, 0, dptr(desc), dptr(bnd64), int(bnd64.numel()), s)
                        torch.cuda.synchronize()
                        want = orc.spmm(rows.rowptr.astype(Ti), cv.astype(Ti), rows.vals.astype(T), np.ascontiguousarray(Bg[ci]))
                        got = C.cpu().numpy()[:, :A.nrows_local].T
                        assert np.array_equal(got, want), f"{tag} {name}: run tiles on column-major blocks differ (pass {it})"
                    capi.call("hpcla_spmm_block_order_hint", dptr(rowptr), 0)
                flag = ctypes.c_int(0)
                capi.call("hpcla_halo_status", halo, ctypes.byref(flag))
                assert flag.value == 0, f"{tag} {name}: an exchange timed out"
        hp.check_exchange_health(backend, always=True)
        print(f"{tag} ok", flush=True)
        hp.clear_spmm_cache()
        hp.clear_plan_cache()
        os.environ.pop("HPCLA_NARROW_INDICES", None)
    dist.barrier()
    return 0


if __name__ == "__main__":
    sys.exit(main())
