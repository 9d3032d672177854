"""Conjugate-gradient harness for BASELINE config 4 (3-D Poisson, 100 iterations).

The reference has no Krylov solver; a caller composes one from ``A*p``
(src/sparse.jl:2096-2128), ``dot`` (src/vectors.jl:798-812) and broadcast updates
(src/vectors.jl:1203-1226).  This harness does exactly that with the DeviceROCm operators, keeping
every scalar on the device: alpha = rr/pAp and beta = rr_new/rr are consumed by the update kernels
as (numerator, denominator) device pointers, so an iteration never synchronises the host.

Forms, identical arithmetic per element (only the reduction trees differ):

* ``fused=False``  one library call per reference operator: SpMV, dot, axpy, axpy, norm, xpay
  (8 kernel launches, SpMV + 96 B/row of vector traffic -- the "textbook unfused" count of SURVEY 8d);
* ``fused=True``   SpMV with the p.Ap partials in its epilogue, one kernel for r -= a Ap / sum r^2, one for
  x += a p / p = r + b p (the x update is deferred to where p is read anyway): SpMV + 64 B/row.
  ``native_loop=True`` (default) hands ALL iterations to the library in one call
  (``hpcla_cg_iterations_f64_*``: the same three launches per iteration, same arguments, same bits), so
  the host language is not in the loop at all; ``native_loop=False`` issues the three calls per iteration
  from Python.

The solve is split into ``cg_setup`` (x = 0, r = p = b, sum r0^2 -- allocation lives in a reusable
``CGWorkspace``) and ``cg_iterate`` (exactly k iterations, enqueue only), so a benchmark can time the
iterations and nothing else; ``cg_fixed_iterations`` is the two together plus the history read-back.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np

from . import _capi
from .sparse import get_vector_plan, mul_, mul_dot_
from .vectors import HPCVector, _Scratch, cg_direction_, cg_residual_, current_stream_ptr, dot, dptr, norm


def _torch():
    import torch
    return torch


def _cg_iteration(A, x, r, p, Ap, rr_cur, rr_nxt, pAp, fused: bool) -> None:
    if fused:
        mul_dot_(Ap, A, p, pAp)                                 # Ap = A*p, pAp = p.Ap
        cg_residual_(r, Ap, 1.0, rr_cur, pAp, rr_nxt)           # r -= a Ap; rr_new        (a = rr/pAp)
        cg_direction_(x, p, r, 1.0, rr_cur, pAp, 1.0, rr_nxt, rr_cur)   # x += a p; p = r + (rr_new/rr) p
        return
    mul_(Ap, A, p)                                              # Ap = A*p
    dot(p, Ap, out=pAp)                                         # pAp
    x.axpy_(1.0, p, num=rr_cur, den=pAp)                        # x += (rr/pAp) p
    r.axpy_(-1.0, Ap, num=rr_cur, den=pAp)                      # r -= (rr/pAp) Ap
    norm(r, 2, out=rr_nxt)                                      # rr_new = sum(r^2)
    p.xpay_(r, 1.0, num=rr_nxt, den=rr_cur)                     # p = r + (rr_new/rr) p


class CGWorkspace:
    """Everything a solve allocates: x, r, p, Ap (partitioned like b), the history array (sum r_k^2 lives in
    hist[k] -- it IS the rr storage of the iteration, so recording costs no copies), pAp.  Reusable across
    solves of the same size; ``cg_fixed_iterations`` makes one when none is given."""

    def __init__(self, b: HPCVector, max_iters: int):
        torch = _torch()
        dev = b.v.device
        self.x = HPCVector.zeros(b.partition, b.backend)
        self.r = b.similar()
        self.p = b.similar()
        self.Ap = b.similar()
        self.max_iters = int(max_iters)
        self.hist = torch.zeros(self.max_iters + 2, dtype=torch.float64, device=dev)
        self.pAp = torch.zeros(1, dtype=torch.float64, device=dev)
        self.done = 0                  # iterations since the last cg_setup

    def fits(self, b: HPCVector, iters: int) -> bool:
        return self.max_iters >= iters and self.x.structural_hash == b.structural_hash and self.x.v.device == b.v.device


def cg_setup(A, b: HPCVector, ws: CGWorkspace, fused: bool = True):
    """x0 = 0, r0 = p0 = b, hist[0] = sum r0^2; builds / looks up the plan.  Returns the state ``cg_iterate``
    continues from.  Not an iteration: benchmarks keep it outside the timed region."""
    from .vectors import f64_only
    f64_only(A.backend, "the CG building blocks")
    plan = get_vector_plan(A, ws.p)
    fused = bool(fused and plan.result_partition_hash == ws.p.structural_hash)
    ws.x.v.zero_()
    ws.r.v.copy_(b.v)
    ws.p.v.copy_(b.v)
    ws.hist.zero_()
    norm(ws.r, 2, out=ws.hist[0:1])            # out form leaves sum(r^2) on the device (8 B/elt)
    ws.done = 0
    if fused:                                  # the SpMV+dot scratch hangs off the plan
        torch = _torch()
        if getattr(plan, "_dot_work", None) is None:
            nbytes = _capi.load().hpcla_spmv_dot_work_bytes(A.nrows_local)
            plan._dot_work = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=ws.p.v.device)
    return plan, fused


def cg_iterate(A, ws: CGWorkspace, plan, fused: bool, iters: int, native_loop: bool = True) -> None:
    """Exactly ``iters`` more iterations, enqueued on the current stream (no allocation, no host sync)."""
    first = ws.done
    if first + iters > ws.max_iters:
        raise ValueError("cg_iterate: workspace history too short")
    hist = ws.hist
    if fused and native_loop and A._packed_for(plan) is None:
        sfx = "i64" if plan.is_i64 else "i32"
        work, _ = _Scratch.get(ws.r.v.device)
        _capi.call(f"hpcla_cg_iterations_f64_{sfx}", plan.halo if plan.has_halo else None, A.backend.rccl,
                   dptr(plan.rowptr_of(A)), dptr(plan.colval_split), dptr(A.nzval), A.nrows_local, A.nnz, 0,
                   dptr(plan.interior), plan.n_interior, dptr(plan.boundary), plan.n_boundary,
                   dptr(ws.x.v), dptr(ws.r.v), dptr(ws.p.v), dptr(ws.Ap.v), dptr(hist[first:]), dptr(ws.pAp),
                   dptr(plan._dot_work), dptr(work), int(iters), current_stream_ptr())
    else:
        for k in range(first, first + iters):
            _cg_iteration(A, ws.x, ws.r, ws.p, ws.Ap, hist[k:k + 1], hist[k + 1:k + 2], ws.pAp, fused)
    ws.done = first + iters


def cg_fixed_iterations(A, b: HPCVector, iters: int, record_history: bool = True,
                        fused: bool = True, graph: bool = False, native_loop: bool = True,
                        workspace: Optional[CGWorkspace] = None) -> Tuple[HPCVector, List[float]]:
    """Textbook CG from x0 = 0, exactly ``iters`` iterations, no convergence exit.
    Returns (x, [||r_0||, ..., ||r_iters||]) (history read back once at the end; ``record_history=False``
    skips the read-back and returns []).

    ``graph=True`` captures one PAIR of iterations (their scalars ping-pong through two fixed slots and return
    to the start after two) into a HIP graph and replays it: every library entry point only enqueues work on
    the stream it is given (no allocation, no host sync) and the exchange epoch lives in device memory, so
    whole distributed iterations are capturable.  Same kernels, same arguments, hence the same bits as the
    eager loop.  It pays off where an iteration is shorter than the host time to issue its launches (small
    systems); profiles/MEASUREMENTS_r03.md has the measurement for large ones, where eager stays the default."""
    if A.backend.T == np.dtype(np.float32):
        return _cg_composed_f32(A, b, iters, record_history)
    torch = _torch()
    ws = workspace if workspace is not None and workspace.fits(b, iters) else CGWorkspace(b, iters)
    plan, fused = cg_setup(A, b, ws, fused)
    if not graph or iters < 4:
        cg_iterate(A, ws, plan, fused, iters, native_loop)
    else:
        _cg_graph_replay(A, ws, plan, fused, iters)
    h = ws.hist[:iters + 1].sqrt().cpu().tolist() if record_history else []
    if h and h[-1] != h[-1]:                   # NaN: breakdown, or the poison of an expired exchange wait -- ask
        from .sparse import check_exchange_health
        check_exchange_health(b.backend)
    return ws.x, h


def _cg_composed_f32(A, b: HPCVector, iters: int, record_history: bool):
    """The same iteration on a Float32 backend, composed from the Float32 operators the way a caller of the reference composes
    it (`A*p`, `dot`, the vector updates; there is no reference solver): mul!, dot (formed in double, rounded to
    Float32), u + a*v.  Not fused, one host read-back per scalar -- the fused entries (hpcla_cg_iterations_*) are Float64's."""
    from .sparse import mul_
    from .vectors import dot
    x = HPCVector.zeros(b.partition, b.backend)
    r, p = b.copy(), b.copy()
    Ap = b.similar()
    rr = dot(r, r)
    hist = [math.sqrt(rr)]
    for _ in range(iters):
        mul_(Ap, A, p)
        pAp = dot(p, Ap)
        alpha = float(np.float32(rr) / np.float32(pAp))
        x = x._axpby(1.0, p, alpha)
        r = r._axpby(1.0, Ap, -alpha)
        rr_new = dot(r, r)
        beta = float(np.float32(rr_new) / np.float32(rr))
        p = r._axpby(1.0, p, beta)
        rr = rr_new
        hist.append(math.sqrt(rr))
    if hist[-1] != hist[-1]:
        from .sparse import check_exchange_health
        check_exchange_health(b.backend)
    return x, (hist if record_history else [])


class CGGraphPair:
    """One PAIR of iterations captured into a HIP graph (state must have run >= 2 eager iterations: they size
    the scratch buffers and probe the plan).  ``replay(n)`` continues the solve by 2n iterations."""

    def __init__(self, A, ws: CGWorkspace, plan, fused: bool, native_loop: bool = False):
        torch = _torch()
        dev = ws.hist.device
        self.ws = ws
        self.rr = torch.zeros(2, dtype=torch.float64, device=dev)
        self.pair_hist = torch.zeros(2, dtype=torch.float64, device=dev)
        rr = self.rr
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):                           # capture only: nothing executes here
            _cg_iteration(A, ws.x, ws.r, ws.p, ws.Ap, rr[0:1], rr[1:2], ws.pAp, fused)
            self.pair_hist[0:1].copy_(rr[1:2])
            _cg_iteration(A, ws.x, ws.r, ws.p, ws.Ap, rr[1:2], rr[0:1], ws.pAp, fused)
            self.pair_hist[1:2].copy_(rr[0:1])

    def replay(self, n_pairs: int, record_history: bool = True) -> None:
        ws = self.ws
        self.rr[0:1].copy_(ws.hist[ws.done:ws.done + 1])
        for _ in range(n_pairs):
            self.graph.replay()
            if record_history:
                ws.hist[ws.done + 1:ws.done + 3].copy_(self.pair_hist)
            ws.done += 2
        if not record_history:
            ws.hist[ws.done:ws.done + 1].copy_(self.rr[0:1])


def _cg_graph_replay(A, ws: CGWorkspace, plan, fused: bool, iters: int) -> None:
    cg_iterate(A, ws, plan, fused, 2, native_loop=False)             # eager: allocations, plan probes
    g = CGGraphPair(A, ws, plan, fused)
    g.replay((iters - 2) // 2)
    if ws.done < iters:
        cg_iterate(A, ws, plan, fused, iters - ws.done, native_loop=False)
