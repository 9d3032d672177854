"""Conjugate-gradient harness for BASELINE config 4 (3-D Poisson, 100 iterations).

The reference has no Krylov solver; a caller composes one from ``A*p``
(src/sparse.jl:2096-2128), ``dot`` (src/vectors.jl:798-812) and broadcast updates
(src/vectors.jl:1203-1226).  This harness does exactly that with the DeviceROCm operators, keeping
every scalar on the device: alpha = rr/pAp and beta = rr_new/rr are consumed by the update kernels
as (numerator, denominator) device pointers, so an iteration never synchronises the host.

Two forms, identical arithmetic per element (only the reduction trees differ):

* ``fused=False``  one library call per reference operator: SpMV, dot, axpy, axpy, norm, xpay
  (8 kernel launches, SpMV + 96 B/row of vector traffic -- the "textbook unfused" count of SURVEY 8d);
* ``fused=True``   SpMV with the p.Ap partials in its epilogue, one kernel for r -= a Ap / sum r^2, one for
  x += a p / p = r + b p (the x update is deferred to where p is read anyway): 5 launches,
  SpMV + 64 B/row.
"""
from __future__ import annotations

from typing import List, Tuple

from .sparse import get_vector_plan, mul_, mul_dot_
from .vectors import HPCVector, cg_direction_, cg_residual_, dot, norm


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


def cg_fixed_iterations(A, b: HPCVector, iters: int, record_history: bool = True,
                        fused: bool = True, graph: bool = False) -> Tuple[HPCVector, List[float]]:
    """Textbook CG from x0 = 0, exactly ``iters`` iterations, no convergence exit.
    Returns (x, [||r_0||, ..., ||r_iters||]) (history read back once at the end).

    ``graph=True`` captures one PAIR of iterations (the rr ping-pong returns to its start after two)
    into a HIP graph and replays it: every library entry point only enqueues work on the stream it is
    given (no allocation, no host sync), so the whole iteration is capturable.  Same kernels, same
    arguments, hence the same bits as the eager loop; it pays off where an iteration is shorter than
    the host time to issue its launches (small systems).  The first two iterations always run
    eagerly (they also size the scratch buffers and probe the plan)."""
    torch = _torch()
    dev = b.v.device
    x = HPCVector.zeros(b.partition, b.backend)
    r = b.copy()
    p = b.copy()
    Ap = b.similar()
    plan = get_vector_plan(A, p)               # build/cached plan outside the loop
    fused = fused and plan.result_partition_hash == p.structural_hash
    # device scalars: rr[2] ping-pong, pAp
    rr = torch.zeros(2, dtype=torch.float64, device=dev)
    pAp = torch.zeros(1, dtype=torch.float64, device=dev)
    hist = torch.zeros(iters + 2, dtype=torch.float64, device=dev)
    norm(r, 2, out=rr[0:1])                    # out form leaves sum(r^2) on the device (8 B/elt)
    if record_history:
        hist[0:1].copy_(rr[0:1])

    def pair(first: int, count: int) -> None:
        cur = first & 1
        for k in range(count):
            nxt = 1 - cur
            _cg_iteration(A, x, r, p, Ap, rr[cur:cur + 1], rr[nxt:nxt + 1], pAp, fused)
            if record_history:
                hist[first + k + 1:first + k + 2].copy_(rr[nxt:nxt + 1])
            cur = nxt

    if (not graph or iters < 4) and record_history:
        # eager loop: the history array IS the rr storage (sum r_k^2 lives in hist[k]), so recording costs no copies
        for k in range(iters):
            _cg_iteration(A, x, r, p, Ap, hist[k:k + 1], hist[k + 1:k + 2], pAp, fused)
    elif not graph or iters < 4:
        pair(0, iters)
    else:
        pair(0, 2)                                                   # eager: allocations, plan probes
        pair_hist = torch.zeros(2, dtype=torch.float64, device=dev)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):                                    # capture only: nothing executes here
            _cg_iteration(A, x, r, p, Ap, rr[0:1], rr[1:2], pAp, fused)
            pair_hist[0:1].copy_(rr[1:2])
            _cg_iteration(A, x, r, p, Ap, rr[1:2], rr[0:1], pAp, fused)
            pair_hist[1:2].copy_(rr[0:1])
        done = 2
        while done + 2 <= iters:
            g.replay()
            if record_history:
                hist[done + 1:done + 3].copy_(pair_hist)
            done += 2
        if done < iters:
            pair(done, iters - done)
    h = hist[:iters + 1].sqrt().cpu().tolist() if record_history else []
    return x, h
