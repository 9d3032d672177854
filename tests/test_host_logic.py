"""CPU tests of the host layer (no GPU): partitions, column compression, VectorPlan lists, the
split-column map, and that libhpcla_rocm.so loads and exports every symbol of include/hpcla_rocm.h."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol(hp):
    hdr = open(os.path.join(ROOT, "include", "hpcla_rocm.h")).read()
    declared = set(re.findall(r"\b(hpcla_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hpcla_comm", "hpcla_halo_plan"}
    lib = hp._capi.load()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in the header but not exported"
    assert declared == set(hp._capi.EXPORTED_SYMBOLS), declared ^ set(hp._capi.EXPORTED_SYMBOLS)
    assert lib.hpcla_version() == 100
    assert lib.hpcla_spmv_rows_per_block() == 256


def test_capi_argument_errors_without_gpu(hp):
    """Argument validation happens on the host before any launch (error convention: negative
    status + hpcla_last_error text)."""
    lib = hp._capi.load()
    assert lib.hpcla_spmv_csr_f64_i32(None, None, None, None, None, -1, 0, 0, None) == -1
    assert "negative" in hp._capi.last_error()
    assert lib.hpcla_spmv_csr_f64_i32(None, None, None, None, None, 4, 0, 2, None) == -1
    assert lib.hpcla_spmv_csr_f64_i32(None, None, None, None, None, 0, 0, 0, None) == 0   # empty matrix
    with pytest.raises(hp._capi.HPCLAError):
        hp._capi.call("hpcla_remap_i32", None, None, None, 5, 0, None)


@pytest.mark.parametrize("n,nranks", [(10, 4), (8, 2), (7, 7), (3, 5), (0, 2), (16777216, 8)])
def test_uniform_partition(hp, orc, n, nranks):
    p = hp.uniform_partition(n, nranks)
    np.testing.assert_array_equal(p, orc.uniform_partition(n, nranks))
    assert p[0] == 0 and p[-1] == n and np.all(np.diff(p) >= 0)
    assert np.diff(p).max() - np.diff(p).min() <= 1


def test_owner_of_clamps(hp, orc):
    part = np.array([0, 3, 3, 8])
    g = np.arange(0, 9)
    np.testing.assert_array_equal(hp.owner_of(part, g), orc.owner_of(part, g))
    assert hp.owner_of(part, np.array([8]))[0] == 2      # index == last boundary -> clamped


def test_compress_columns_matches_reference_semantics(hp, orc):
    from hpcla_amd.sparse import _compress_columns
    rows = orc.sprand_rows(5000, 0.002, 100, 900)
    ci_ref, cv_ref = orc.compress_columns(rows)          # unique!(sort) + searchsortedfirst
    ci, cv = _compress_columns(rows.colidx, 5000, np.int32)
    np.testing.assert_array_equal(ci, ci_ref)
    np.testing.assert_array_equal(cv, cv_ref)
    ci0, cv0 = _compress_columns(np.empty(0, dtype=np.int64), 10, np.int32)
    assert len(ci0) == 0 and len(cv0) == 0
    with pytest.raises(ValueError):
        _compress_columns(np.array([11]), 10, np.int32)


def _plans_serial_sim(hp, cis, xp):
    """Run build_host_vector_plan for every rank in ONE process with a fake comm that replays the
    exchanges from the other ranks' inputs."""
    from hpcla_amd import backends as B
    nranks = len(xp) - 1
    owners = [hp.owner_of(xp, ci) for ci in cis]
    counts = np.array([[int(np.sum(o == q)) for q in range(nranks)] for o in owners])

    class FakeComm(B.AbstractComm):
        def __init__(self, r):
            self.r = r

    plans = []
    orig = (B.comm_rank, B.comm_size)
    import hpcla_amd.sparse as S
    saved = (S.comm_rank, S.comm_size, S.comm_alltoall_counts, S.comm_exchange_indices)
    try:
        S.comm_rank = lambda c: c.r
        S.comm_size = lambda c: nranks
        S.comm_alltoall_counts = lambda c, sc: counts[:, c.r].copy()

        def exch(c, send_to, send_arrays, recv_from, recv_counts):
            out = []
            for q, cnt in zip(recv_from, recv_counts):
                req = cis[q][owners[q] == c.r]
                assert len(req) == cnt
                out.append(req.copy())
            return out
        S.comm_exchange_indices = exch
        for r in range(nranks):
            plans.append(S.build_host_vector_plan(cis[r], xp, FakeComm(r)))
    finally:
        S.comm_rank, S.comm_size, S.comm_alltoall_counts, S.comm_exchange_indices = saved
    return plans


@pytest.mark.parametrize("nranks", [1, 2, 3, 5])
@pytest.mark.parametrize("kind", ["poisson", "sprand"])
def test_host_vector_plan_equals_oracle(hp, orc, nranks, kind):
    if kind == "poisson":
        nx, ny = 12, 15
        n = nx * ny
        gen = lambda lo, hi: orc.poisson2d_rows(nx, ny, lo, hi)
    else:
        n = 600
        gen = lambda lo, hi: orc.sprand_rows(n, 0.01, lo, hi)
    rp = orc.uniform_partition(n, nranks)
    xp = orc.uniform_partition(n, nranks) if nranks < 3 else np.sort(
        np.concatenate([[0, n], (np.arange(1, nranks) * n) // nranks + 3]))   # x partition != row partition
    cis = [orc.compress_columns(gen(int(rp[r]), int(rp[r + 1])))[0] for r in range(nranks)]
    want = orc.vector_plans(cis, xp)
    got = _plans_serial_sim(hp, cis, xp)
    x = orc.fill_uniform(0, n, 7)
    xl = [x[xp[r]:xp[r + 1]] for r in range(nranks)]
    gathered = orc.execute_plans(want, xl)
    for r in range(nranks):
        g, w = got[r], want[r]
        assert g.send_rank_ids == w.send_rank_ids and g.recv_rank_ids == w.recv_rank_ids
        for a, b in zip(g.send_indices, w.send_indices):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(g.recv_perm, w.recv_perm):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(g.local_src_indices, w.local_src_indices)
        np.testing.assert_array_equal(g.local_dst_indices, w.local_dst_indices)
        # split-column map: [x.v | ghost buffer] indexed by the map reproduces `gathered`
        m = hp.split_column_map(g)
        ghost = np.concatenate([gathered[r][p] for p in g.recv_perm]) if g.recv_perm else np.empty(0)
        ext = np.concatenate([xl[r], ghost])
        np.testing.assert_array_equal(ext[m], gathered[r])
        np.testing.assert_array_equal(gathered[r], x[cis[r]])


def test_structural_hash_is_stable_and_discriminates(hp):
    c = hp.CommSerial()
    a = hp.compute_structural_hash([0, 4], [0, 1, 2], np.array([0, 1, 3], np.int32), np.array([0, 1, 2], np.int32), c)
    b = hp.compute_structural_hash([0, 4], [0, 1, 2], np.array([0, 1, 3], np.int32), np.array([0, 1, 2], np.int32), c)
    d = hp.compute_structural_hash([0, 4], [0, 1, 3], np.array([0, 1, 3], np.int32), np.array([0, 1, 2], np.int32), c)
    assert a == b and a != d and len(a) == 32
    assert hp.compute_partition_hash(np.array([0, 4])) != hp.compute_partition_hash(np.array([0, 5]))


def test_product_path_has_no_oracle_or_cpu_fallback():
    """The package must never import oracle/ and must fail loudly without its HIP library."""
    pkg = os.path.join(ROOT, "linearalgebrampi.jl_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\.*oracle|import_module\(.*oracle|oracle\.(lib|spmv|build)\(",
                                 src, flags=re.M), fn
    import hpcla_amd as hp2
    saved = hp2._capi.LIB_PATH, hp2._capi._lib
    try:
        hp2._capi.LIB_PATH, hp2._capi._lib = "/nonexistent/libhpcla_rocm.so", None
        with pytest.raises(ImportError):
            hp2._capi.load()
    finally:
        hp2._capi.LIB_PATH, hp2._capi._lib = saved


def test_backend_requires_gpu(hp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        hp.backend_rocm_serial()


# ---- repartition range plans (src/vectors.jl:511-620) ---------------------------------------------
def _random_partition(rng, n, nranks):
    cuts = np.sort(rng.integers(0, n + 1, nranks - 1))
    return np.concatenate([[0], cuts, [n]]).astype(np.int64)


@pytest.mark.parametrize("nranks", [1, 2, 5, 8])
def test_range_plans_are_consistent_and_reproduce_the_target_slices(nranks):
    from hpcla_amd.repartition import RangePlan, check_partition
    rng = np.random.default_rng(nranks)
    n = 211
    x = rng.random(n)
    for _ in range(20):
        src, tgt = _random_partition(rng, n, nranks), _random_partition(rng, n, nranks)
        check_partition(tgt, n, nranks)
        plans = [RangePlan(src, tgt, r) for r in range(nranks)]
        outs = [np.full(p.result_local_size, np.nan) for p in plans]
        for r, p in enumerate(plans):
            loc = x[src[r]:src[r + 1]]
            assert p.result_local_size == tgt[r + 1] - tgt[r]
            assert r not in p.send_rank_ids and r not in p.recv_rank_ids
            assert p.send_rank_ids == sorted(p.send_rank_ids) and p.recv_rank_ids == sorted(p.recv_rank_ids)
            outs[r][p.local_dst_offset:p.local_dst_offset + p.local_count] = \
                loc[p.local_src_start:p.local_src_start + p.local_count]
            for dest, (s, c) in zip(p.send_rank_ids, p.send_ranges):
                q = plans[dest]
                i = q.recv_rank_ids.index(r)                      # every send has its matching receive
                assert q.recv_counts[i] == c
                outs[dest][q.recv_offsets[i]:q.recv_offsets[i] + c] = loc[s:s + c]
        for r, p in enumerate(plans):                             # and every receive a matching send
            for srcr, c in zip(p.recv_rank_ids, p.recv_counts):
                j = plans[srcr].send_rank_ids.index(r)
                assert plans[srcr].send_ranges[j][1] == c
            np.testing.assert_array_equal(outs[r], x[tgt[r]:tgt[r + 1]])


def test_check_partition_rejects_malformed_targets():
    from hpcla_amd.repartition import check_partition
    for bad in ([0, 5], [1, 3, 10], [0, 7, 5, 10], [0, 3, 9]):
        with pytest.raises(ValueError):
            check_partition(np.array(bad), 10, 2 if len(bad) == 3 else len(bad) - 1 if bad != [0, 5] else 2)


# ---- whole-slice exchange lists for SpMM ("switch to all-gather of B") -----------------
@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_whole_slice_lists_reproduce_the_gathered_rows(hp, orc, nranks):
    """All ranks simulated in one process: wishes -> Alltoall -> lists; the emulated exchange must put
    every needed row of B where the SpMM's split column map expects it, whole-slice and requested-rows
    neighbours mixed; message sizes agree on both ends; whole-slice sends are contiguous (no pack)."""
    from hpcla_amd.sparse import whole_slice_lists, whole_slice_wishes
    rng = np.random.default_rng(nranks)
    n = 120 * nranks
    xp = orc.uniform_partition(n, nranks)
    # rank r touches ~90 % of rank (r+1)'s columns (-> whole slice), ~10 % of the others (-> requested rows)
    cis = []
    for r in range(nranks):
        sel = []
        for o in range(nranks):
            frac = 1.0 if o == r else (0.9 if o == (r + 1) % nranks else 0.1)
            cols = np.arange(xp[o], xp[o + 1])
            sel.append(cols[rng.random(len(cols)) < frac])
        cis.append(np.sort(np.concatenate(sel)).astype(np.int64))
    oplans = orc.vector_plans(cis, xp)
    plans = [hp.HostVectorPlan(p.send_rank_ids, p.send_indices, p.recv_rank_ids, p.recv_perm, p.local_src_indices,
                               p.local_dst_indices, p.n_gathered, int(xp[r + 1] - xp[r])) for r, p in enumerate(oplans)]
    wishes = [whole_slice_wishes(p, xp, nranks) for p in plans]
    assert all(w[(r + 1) % nranks] == 1 and w.sum() == 1 for r, w in enumerate(wishes)) or nranks == 2
    granted = [np.array([wishes[q][r] for q in range(nranks)]) for r in range(nranks)]      # the Alltoall
    lists = [whole_slice_lists(plans[r], cis[r], xp, wishes[r], granted[r]) for r in range(nranks)]
    Bg = rng.random((n, 3))
    for r in range(nranks):
        send_idx_r, recv_counts, cmap = lists[r]
        segs = []
        for src, cnt in zip(plans[r].recv_rank_ids, recv_counts):
            j = plans[src].send_rank_ids.index(r)
            idx = lists[src][0][j]
            assert len(idx) == cnt                                   # both ends agree on the size
            if wishes[r][src]:
                np.testing.assert_array_equal(idx, np.arange(xp[src + 1] - xp[src]))   # contiguous: sent in place
            segs.append(Bg[xp[src]:xp[src + 1]][idx])
        ext = np.concatenate([Bg[xp[r]:xp[r + 1]]] + segs)
        np.testing.assert_array_equal(ext[cmap], Bg[cis[r]])


@pytest.mark.parametrize("nranks,n_chunks", [(2, 4), (3, 3), (4, 5), (4, 1), (8, 4)])
def test_panel_chunk_lists_agree_on_both_ends_and_reproduce_the_exchange(hp, orc, nranks, n_chunks):
    """Panel-ordered SpMM (HPCLA_SPMM_ORDER=panel): every link's list is cut into chunk-sets on both ends by the same
    formula.  All ranks simulated: per chunk-set the sender's piece has the length the receiver expects, the emulated
    chunk-set exchanges put every ghost row exactly where (chunk_of, newpos) say, the mapping is a bijection onto the
    chunk-sets' ghost buffers, and chunk-sets of a link follow each other in list order (arrival order = column order
    inside a neighbour's segment)."""
    from hpcla_amd.sparse import panel_chunk_lists, whole_slice_lists, whole_slice_wishes
    rng = np.random.default_rng(7 * nranks + n_chunks)
    n = 97 * nranks + 5
    xp = orc.uniform_partition(n, nranks)
    cis = []
    for r in range(nranks):
        sel = []
        for o in range(nranks):
            frac = 1.0 if o == r else (0.95 if o == (r + 1) % nranks else 0.07)    # whole-slice and requested-rows links
            cols = np.arange(xp[o], xp[o + 1])
            sel.append(cols[rng.random(len(cols)) < frac])
        cis.append(np.sort(np.concatenate(sel)).astype(np.int64))
    oplans = orc.vector_plans(cis, xp)
    plans = [hp.HostVectorPlan(p.send_rank_ids, p.send_indices, p.recv_rank_ids, p.recv_perm, p.local_src_indices,
                               p.local_dst_indices, p.n_gathered, int(xp[r + 1] - xp[r])) for r, p in enumerate(oplans)]
    wishes = [whole_slice_wishes(p, xp, nranks) for p in plans]
    granted = [np.array([wishes[q][r] for q in range(nranks)]) for r in range(nranks)]
    lists = [whole_slice_lists(plans[r], cis[r], xp, wishes[r], granted[r]) for r in range(nranks)]
    cut = [panel_chunk_lists(lists[r][0], lists[r][1], n_chunks) for r in range(nranks)]
    Bg = rng.random((n, 2))
    for r in range(nranks):
        send_idx_r, recv_counts, cmap = lists[r]
        _, recv_chunk_counts, chunk_of, newpos = cut[r]
        n_ghost = int(sum(recv_counts))
        # what the UNCUT exchange delivers, in ghost order
        whole = np.concatenate([Bg[xp[src]:xp[src + 1]][lists[src][0][plans[src].send_rank_ids.index(r)]]
                                for src in plans[r].recv_rank_ids]) if n_ghost else np.empty((0, 2))
        # emulate the chunk-set exchanges
        bufs = []
        for c in range(n_chunks):
            parts = []
            for i, src in enumerate(plans[r].recv_rank_ids):
                piece = cut[src][0][c][plans[src].send_rank_ids.index(r)]
                assert len(piece) == recv_chunk_counts[c][i], "the two ends of a link disagree on a chunk's size"
                parts.append(Bg[xp[src]:xp[src + 1]][piece])
            bufs.append(np.concatenate(parts) if parts else np.empty((0, 2)))
        assert sum(len(b) for b in bufs) == n_ghost
        seen = [np.zeros(len(b), dtype=bool) for b in bufs]
        for p in range(n_ghost):
            c, q = int(chunk_of[p]), int(newpos[p])
            np.testing.assert_array_equal(bufs[c][q], whole[p])
            assert not seen[c][q]
            seen[c][q] = True
        assert all(s.all() for s in seen)
        # chunks of one neighbour's segment are contiguous pieces in order
        off = 0
        for cnt in recv_counts:
            seg = chunk_of[off:off + cnt]
            assert np.all(np.diff(seg) >= 0)
            off += cnt


def test_rccl_init_guard_reports_failure_and_deadline(hp, monkeypatch):
    """backends._init_rccl_guarded: ncclCommInitRank runs on a helper thread with a deadline; a failing or a
    hanging initialisation becomes a REASON (all-gathered by the caller, who then continues on the peer
    windows alone) instead of an exception on some ranks or a hang."""
    import ctypes
    import time
    from hpcla_amd import backends

    class FakeLib:
        def __init__(self, status, delay=0.0):
            self.status, self.delay, self.device = status, delay, None

        def hpcla_set_device(self, d):
            self.device = d
            return 0

        def hpcla_comm_init_rank_ex(self, handle_ref, idbuf, nranks, rank, flags):
            time.sleep(self.delay)
            return self.status

    h = ctypes.c_void_p()
    ok = FakeLib(0)
    assert backends._init_rccl_guarded(ok, h, None, 2, 0, 0, 3) == "" and ok.device == 3
    why = backends._init_rccl_guarded(FakeLib(-3), h, None, 2, 0, 0, 0)
    assert why.startswith("status -3")
    monkeypatch.setenv("HPCLA_RCCL_INIT_TIMEOUT_S", "0.2")
    t0 = time.perf_counter()
    why = backends._init_rccl_guarded(FakeLib(0, delay=5.0), h, None, 2, 0, 0, 0)
    assert "did not return" in why and time.perf_counter() - t0 < 3.0


def test_xcd_group_order_is_a_bijection():
    """Restatement of csrc/spmv.hip xcd_group_index (the GPU suite checks the kernel's own through bit-equal results)."""
    def index(b, n, l):
        span = 1 << (3 + l)
        if b >= n - (n & (span - 1)):
            return b
        xcd, q = b & 7, b >> 3
        return ((((q >> l) << 3) + xcd) << l) + (q & ((1 << l) - 1))
    for n in (1, 7, 8, 64, 1238, 4096, 65536 + 13):
        for l in (1, 2, 6, 10):
            img = sorted(index(b, n, l) for b in range(n))
            assert img == list(range(n)), (n, l)
    # blocks of one group land on one XCD: workgroup ids b with equal b % 8 and q // G
    G, l = 64, 6
    got = [index(b, 8 * G * 4, l) for b in range(0, 8 * G, 8)]          # XCD 0's first G workgroups
    assert got == list(range(G))


def test_narrowing_decision_of_int64_plans(hp, monkeypatch):
    """sparse.can_narrow_indices: an Int64 matrix's plan streams Int32 indices iff everything an index array of the
    plan can hold fits -- row pointers (nnz), split columns (own offsets, then positions in the ghost segment) and send
    indices.  Every BASELINE configuration fits per GPU; a synthetic ghost segment past 2^31 must be REFUSED (and so
    must nnz >= 2^31), 1-based callers lose one value."""
    from hpcla_amd.sparse import INT32_MAX, can_narrow_indices, narrowing_enabled
    assert INT32_MAX == 2**31 - 1
    # BASELINE shapes per GPU: config 2 (4096^2), config 3 whole (8192^2), config 4 share, config 5 share
    assert can_narrow_indices(83_869_696, 16_777_216, 16_777_216, 0)
    assert can_narrow_indices(335_511_552, 67_108_864, 67_108_864, 0)
    assert can_narrow_indices(117_178_368, 16_777_216, 16_777_216, 2 * 262_144)
    assert can_narrow_indices(62_500_000, 2_097_152, 2_097_152, 14_680_064)
    # refusals
    assert not can_narrow_indices(1000, 10, 10, 2**31)                    # ghost positions past Int32
    assert not can_narrow_indices(1000, 10, 2**30, 2**30)                 # n_own + n_ghost == 2^31
    assert can_narrow_indices(1000, 10, 2**30, 2**30 - 1)                 # largest split column = 2^31 - 2
    assert not can_narrow_indices(2**31, 10, 10, 0)                       # rowptr[end] = nnz does not fit
    assert can_narrow_indices(2**31 - 1, 10, 10, 0)
    assert not can_narrow_indices(2**31 - 1, 10, 10, 0, index_base=1)     # Julia's 1-based rowptr[end] = nnz + 1
    assert not can_narrow_indices(10, 2**31, 10, 0)
    # the switch
    monkeypatch.setenv("HPCLA_NARROW_INDICES", "0")
    assert not narrowing_enabled()
    monkeypatch.setenv("HPCLA_NARROW_INDICES", "1")
    assert narrowing_enabled()
    monkeypatch.delenv("HPCLA_NARROW_INDICES")
    assert narrowing_enabled()


def test_element_type_of_a_backend_and_what_float32_covers(hp):
    """HPCBackend carries T like the reference's HPCBackend{T,...}: Float64 (graded) and Float32 (the reference's other GPU
    configuration); the C entry-point suffix, the rounding of reduction results and the Float64-only guard follow from it."""
    from hpcla_amd.backends import CommSerial, DeviceCPU, SolverNone
    from hpcla_amd.vectors import _round_to, f64_only, sfx_of
    b64 = hp.HPCBackend(np.float64, np.int32, DeviceCPU(), CommSerial(), SolverNone())
    b32 = hp.HPCBackend(np.float32, np.int64, DeviceCPU(), CommSerial(), SolverNone())
    assert sfx_of(b64) == "f64" and sfx_of(b32) == "f32" and hp.cpu_version(b32).T == np.dtype(np.float32)
    v = 0.1 + 2.0 ** -40
    assert _round_to(b64, v) == v and _round_to(b32, v) == float(np.float32(v)) != v
    f64_only(b64, "anything")
    with pytest.raises(TypeError, match="Float64 backends only"):
        f64_only(b32, "transpose(A) (materialised)")
    for bad in (np.float16, np.complex128, np.int32):
        with pytest.raises(TypeError, match="float64 or float32"):
            hp.HPCBackend(bad, np.int32, DeviceCPU(), CommSerial(), SolverNone())
    # every _f32 entry of the header is bound, and each has its _f64 twin's argument classes (pointers / sizes) up to the scalar type
    sig = hp._capi._SIGNATURES
    for name, args in sig.items():
        if "_f32" in name and name.replace("_f32", "_f64") in sig:
            twin = sig[name.replace("_f32", "_f64")]
            norm = lambda seq: [("s" if a in (hp._capi._f32, hp._capi._f64) else a) for a in seq]
            mine = norm(args)
            if name.startswith("hpcla_spmv_dist_f32"):      # + the staging vector of the widened exchange, before the stream
                assert mine[-2] is hp._capi._vp
                mine = mine[:-2] + mine[-1:]
            assert mine == norm(twin), name


def test_spmm_pitch_and_rows_on_pitch(hp, monkeypatch):
    """Round 6: odd k runs on the even row pitch k + 1 (dense.spmm_pitch, dense._rows_on_pitch) -- host logic, CPU tensors:
    the pitch is a function of (element type, k, order) only (every rank computes the same exchange width); a block that
    already sits on the pitch is taken as it is, any other is copied once; the padding is never part of the block."""
    import types
    import torch
    dense = hp.dense
    A64 = types.SimpleNamespace(T=np.dtype(np.float64))
    A32 = types.SimpleNamespace(T=np.dtype(np.float32))
    monkeypatch.delenv("HPCLA_SPMM_ORDER", raising=False)
    assert [dense.spmm_pitch(A64, k) for k in (0, 1, 2, 3, 4, 5, 15, 16, 17)] == [0, 1, 2, 4, 4, 6, 16, 16, 18]
    assert [dense.spmm_pitch(A32, k) for k in (1, 3, 15)] == [1, 3, 15]            # Float32 kernels keep the pitch k
    monkeypatch.setenv("HPCLA_SPMM_ORDER", "panel")
    assert dense.spmm_pitch(A64, 15) == 15                                        # the panel order keeps the pitch k
    monkeypatch.delenv("HPCLA_SPMM_ORDER")
    M = torch.arange(35, dtype=torch.float64).reshape(7, 5)
    P = dense._rows_on_pitch(M, 6)
    assert tuple(P.shape) == (7, 6) and P.stride(0) == 6 and torch.equal(P[:, :5], M)
    V = P[:, :5]                                                                   # a result as spmm() hands it out
    assert dense._rows_on_pitch(V, 6).data_ptr() == V.data_ptr()                   # already on the pitch: no copy
    assert dense._rows_on_pitch(V, 5).is_contiguous() and torch.equal(dense._rows_on_pitch(V, 5), M)
    E = dense._rows_on_pitch(torch.empty((0, 5), dtype=torch.float64), 6)
    assert tuple(E.shape) == (0, 6)

