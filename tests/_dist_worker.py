"""Worker for tests/test_distributed_cpu.py: one process per rank, gloo backend, CPU only.
Exercises the real comm_* primitives and build_host_vector_plan across processes and checks the
result against the single-process oracle restatement.  Exit code 0 = all checks passed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    import hpcla_amd as hp
    from hpcla_amd import backends as B
    from oracle import oracle as orc

    dist.init_process_group("gloo")
    rank, nranks = dist.get_rank(), dist.get_world_size()
    comm = hp.CommTorch()
    assert hp.comm_rank(comm) == rank and hp.comm_size(comm) == nranks

    # comm primitives (src/backends.jl:225-327)
    ag = B.comm_allgather(comm, np.array([rank, 10 * rank]))
    assert ag.tolist() == [v for r in range(nranks) for v in (r, 10 * r)]
    a2a = B.comm_alltoall_counts(comm, np.array([100 * rank + q for q in range(nranks)]))
    assert a2a.tolist() == [100 * q + rank for q in range(nranks)]
    blob = B.comm_bcast_bytes(comm, bytes(range(128)) if rank == 0 else None, 128, root=0)
    assert blob == bytes(range(128))
    hs = B.comm_allgather_bytes(comm, bytes([rank]) * 32)
    assert hs == [bytes([r]) * 32 for r in range(nranks)]

    for kind in ("poisson", "sprand", "nonuniform", "config5"):
        if kind == "config5":
            # BASELINE config 5's shape: uniformly random columns, ~30 entries per row, so a rank touches ~98 % of EVERY
            # other rank's slice -- at 8 ranks each rank has 7 send and 7 recv neighbours (src/sparse.jl:1875-1984)
            n = 40 * nranks
            gen = lambda lo, hi: orc.sprand_rows(n, 30.0 / n, lo, hi)
            xp = orc.uniform_partition(n, nranks)
        elif kind == "poisson":
            nx, ny = 16, 4 * nranks + 1
            n = nx * ny
            gen = lambda lo, hi: orc.poisson2d_rows(nx, ny, lo, hi)
            xp = orc.uniform_partition(n, nranks)
        elif kind == "sprand":
            n = 500
            gen = lambda lo, hi: orc.sprand_rows(n, 0.02, lo, hi)
            xp = orc.uniform_partition(n, nranks)
        else:
            n = 300
            gen = lambda lo, hi: orc.sprand_rows(n, 0.03, lo, hi)
            xp = np.array([0] + [min(n, 40 + (n * r) // nranks) for r in range(1, nranks)] + [n])
        rp = orc.uniform_partition(n, nranks)
        cis = [orc.compress_columns(gen(int(rp[r]), int(rp[r + 1])))[0] for r in range(nranks)]
        want = orc.vector_plans(cis, xp)[rank]
        got = hp.build_host_vector_plan(cis[rank], xp, comm)
        assert got.send_rank_ids == want.send_rank_ids, (kind, got.send_rank_ids, want.send_rank_ids)
        assert got.recv_rank_ids == want.recv_rank_ids
        for a, b in zip(got.send_indices, want.send_indices):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(got.recv_perm, want.recv_perm):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(got.local_src_indices, want.local_src_indices)
        np.testing.assert_array_equal(got.local_dst_indices, want.local_dst_indices)
        others = [r for r in range(nranks) if r != rank]
        if kind == "config5":
            # the target's shape: every rank exchanges with every other rank (7 neighbours each way at 8 ranks)
            assert got.send_rank_ids == others and got.recv_rank_ids == others, (got.send_rank_ids, got.recv_rank_ids)
        if kind == "poisson":
            # row slabs of a 2-D grid: the ranks next door and nobody else, ghost = one grid line per neighbour
            nb = [r for r in (rank - 1, rank + 1) if 0 <= r < nranks]
            assert got.send_rank_ids == nb and got.recv_rank_ids == nb
            assert all(len(p) == nx for p in got.recv_perm) and all(len(i) == nx for i in got.send_indices)

        # tag-21 value exchange over gloo, driven by the plan lists, must reproduce x[col_indices]
        import torch
        x = orc.fill_uniform(0, n, 11)
        xl = x[xp[rank]:xp[rank + 1]]
        ops, bufs, keep = [], [], []
        for r, perm in zip(got.recv_rank_ids, got.recv_perm):
            t = torch.empty(len(perm), dtype=torch.float64)
            bufs.append(t)
            ops.append(dist.P2POp(dist.irecv, t, r))
        for r, idx in zip(got.send_rank_ids, got.send_indices):
            t = torch.from_numpy(xl[idx].copy())
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, r))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        ghost = np.concatenate([b.numpy() for b in bufs]) if bufs else np.empty(0)
        ext = np.concatenate([xl, ghost])
        np.testing.assert_array_equal(ext[hp.split_column_map(got)], x[cis[rank]])

        if kind == "config5":
            # SpMM ghost rows (dense._spmm_plan): wishes -> Alltoall -> whole-slice lists, then the width-k exchange over
            # gloo; every needed row of B must sit where the split column map of the SpMM says
            from hpcla_amd.sparse import whole_slice_lists, whole_slice_wishes
            wish = whole_slice_wishes(got, xp, nranks)
            assert wish.tolist() == [0 if r == rank else 1 for r in range(nranks)], wish
            granted = B.comm_alltoall_counts(comm, wish)
            assert granted.tolist() == wish.tolist()
            send_idx, recv_counts, cmap = whole_slice_lists(got, cis[rank], xp, wish, granted)
            k = 3
            Bg = orc.fill_uniform(0, n * k, 17).reshape(n, k)
            Bl = Bg[xp[rank]:xp[rank + 1]]
            for i in send_idx:                                       # whole slices leave B in place: one contiguous run
                np.testing.assert_array_equal(i, np.arange(len(Bl)))
            got_rows = B.comm_exchange_arrays(comm, got.send_rank_ids, [Bl[i].ravel() for i in send_idx],
                                              got.recv_rank_ids, [c * k for c in recv_counts], np.float64)
            ext = np.concatenate([Bl] + [g.reshape(-1, k) for g in got_rows])
            np.testing.assert_array_equal(ext[cmap], Bg[cis[rank]])

        # structural hash agrees on every rank
        rows = gen(int(rp[rank]), int(rp[rank + 1]))
        _, cv = orc.compress_columns(rows)
        hsh = hp.compute_structural_hash(rp, cis[rank], rows.rowptr, cv, comm)
        allh = B.comm_allgather_bytes(comm, hsh)
        assert all(h == hsh for h in allh)

    # TransposePlan redistribution (src/sparse.jl:1519-1865): rows of A^T land on the owner of the
    # column in col_partition, ascending (row, col); compare with scipy's global transpose
    import scipy.sparse as sp
    m, ncol = 230, 170
    glob = orc.sprand_rows(ncol, 0.03, 0, m)
    Ag = sp.csr_matrix((glob.vals, glob.colidx, glob.rowptr), shape=(m, ncol))
    rp, cp = orc.uniform_partition(m, nranks), orc.uniform_partition(ncol, nranks)
    loc = Ag[rp[rank]:rp[rank + 1], :]
    st = hp.HostTransposeStructure(loc.indptr, loc.indices, rp, cp, comm)
    # emulate TransposePlan.execute on the host: gather into send order, range exchange, gather into result order
    send = loc.data[st.send_order]
    got = B.comm_exchange_arrays(comm, st.peers_out, [send[st.bounds[r]:st.bounds[r + 1]] for r in st.peers_out],
                                 st.peers_in, st.recv_counts, np.float64)
    recv = np.concatenate([send[st.bounds[rank]:st.bounds[rank + 1]]] + got)
    assert len(recv) == st.n_total and st.n_own_segment == st.bounds[rank + 1] - st.bounds[rank]
    trp, tci, tv = st.rowptr_t, st.col_t, recv[st.final_order]
    AT = Ag.T.tocsr()
    AT.sort_indices()
    want = AT[cp[rank]:cp[rank + 1], :]
    np.testing.assert_array_equal(trp, want.indptr)
    np.testing.assert_array_equal(tci, want.indices)
    np.testing.assert_array_equal(tv, want.data)

    # MatrixPlan row gather (src/sparse.jl:554-978): structure of the gathered rows of B == B[needed,:]
    from hpcla_amd.matmat import build_row_gather_plan
    kk, nb = 170, 90
    Bg = orc.sprand_rows(nb, 0.05, 0, kk)
    Bm = sp.csr_matrix((Bg.vals, Bg.colidx, Bg.rowptr), shape=(kk, nb))
    bp = orc.uniform_partition(kk, nranks)
    Bloc = Bm[bp[rank]:bp[rank + 1], :]
    needed = np.unique(np.random.default_rng(100 + rank).integers(0, kk, 60))
    pl = build_row_gather_plan(needed, bp, Bloc.indptr, lambda pos: Bloc.indices[pos].astype(np.int64), comm)
    want_g = Bm[needed, :]
    np.testing.assert_array_equal(pl.g_rowptr, want_g.indptr)
    np.testing.assert_array_equal(pl.g_col, want_g.indices)
    # value movement lists reproduce the values: emulate the exchange over gloo
    ops, bufs, keep = [], [], []
    for r, cnt in zip(pl.recv_rank_ids, pl.recv_counts):
        t = torch.empty(cnt, dtype=torch.float64); bufs.append(t); ops.append(dist.P2POp(dist.irecv, t, r))
    for r, pos in zip(pl.send_rank_ids, pl.send_pos):
        t = torch.from_numpy(Bloc.data[pos].copy()); keep.append(t); ops.append(dist.P2POp(dist.isend, t, r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    g_val = np.full(len(pl.g_col), np.nan)
    g_val[pl.local_dst_start:pl.local_dst_start + len(pl.local_src)] = Bloc.data[pl.local_src]
    for b, dst in zip(bufs, pl.recv_dst_start):
        g_val[dst:dst + len(b)] = b.numpy()
    np.testing.assert_array_equal(g_val, want_g.data)

    # repartition plans (src/vectors.jl:470-722, src/sparse.jl:4069-4600): range lists drive a gloo
    # exchange that must reproduce the global object sliced by the target partition
    from types import SimpleNamespace
    from hpcla_amd.repartition import RangePlan, SparseRepartitionPlan

    def run_ranges(src, dst, send_ranks, send_off, send_cnt, recv_ranks, recv_off, recv_cnt, ls, ld, lc):
        dst[ld:ld + lc] = src[ls:ls + lc]
        got_bufs = B.comm_exchange_arrays(comm, send_ranks, [src[o:o + c] for o, c in zip(send_off, send_cnt)],
                                          recv_ranks, recv_cnt, np.float64)
        for o, b in zip(recv_off, got_bufs):
            dst[o:o + len(b)] = b

    nvec = 97
    xg = orc.fill_uniform(0, nvec, 5)
    src_p = orc.uniform_partition(nvec, nranks)
    for tgt in (np.array([0] + [3 * r for r in range(1, nranks)] + [nvec]),          # almost all on the last rank
                np.array([0] + [nvec] * nranks),                                      # all on rank 0, others empty
                src_p):
        pl = RangePlan(src_p, tgt.astype(np.int64), rank)
        out = np.full(pl.result_local_size, np.nan)
        run_ranges(xg[src_p[rank]:src_p[rank + 1]], out, pl.send_rank_ids, [s for s, _ in pl.send_ranges],
                   [c for _, c in pl.send_ranges], pl.recv_rank_ids, pl.recv_offsets, pl.recv_counts,
                   pl.local_src_start, pl.local_dst_offset, pl.local_count)
        np.testing.assert_array_equal(out, xg[tgt[rank]:tgt[rank + 1]])

    tgt = np.array([0] + [min(m, 7 + (m * r) // nranks + 11 * r) for r in range(1, nranks)] + [m], dtype=np.int64)
    ci = np.unique(loc.indices).astype(np.int64)
    cv = np.searchsorted(ci, loc.indices).astype(np.int32)
    fakeA = SimpleNamespace(row_partition=rp, col_partition=cp, rowptr=loc.indptr.astype(np.int32), colval=cv,
                            col_indices=ci, backend=SimpleNamespace(comm=comm, Ti=np.dtype(np.int32)))
    sp_plan = SparseRepartitionPlan(fakeA, tgt)
    want_r = Ag[tgt[rank]:tgt[rank + 1], :]
    np.testing.assert_array_equal(sp_plan.result_rowptr, want_r.indptr)
    np.testing.assert_array_equal(sp_plan.result_col_indices[sp_plan.result_colval.astype(np.int64)], want_r.indices)
    np.testing.assert_array_equal(sp_plan.result_col_indices, np.unique(want_r.indices))
    vals_out = np.full(len(want_r.data), np.nan)
    run_ranges(loc.data, vals_out, sp_plan.rows.send_rank_ids, sp_plan.send_value_offsets, sp_plan.send_nnz_counts,
               sp_plan.rows.recv_rank_ids, sp_plan.recv_value_offsets, sp_plan.recv_nnz_counts,
               sp_plan.local_value_src, sp_plan.local_value_offset, sp_plan.local_nnz)
    np.testing.assert_array_equal(vals_out, want_r.data)

    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: OK")


if __name__ == "__main__":
    main()
