"""Conjugate-gradient harness for BASELINE config 4 (3-D Poisson, 100 iterations).

The reference has no Krylov solver; a caller composes one from ``A*p``
(src/sparse.jl:2096-2128), ``dot`` (src/vectors.jl:798-812) and broadcast updates
(src/vectors.jl:1203-1226).  This harness does exactly that with the DeviceROCm operators, keeping
every scalar on the device: alpha = rr/pAp and beta = rr_new/rr are consumed by the update kernels
as (numerator, denominator) device pointers, so an iteration never synchronises the host.

Two forms, identical arithmetic per element (only the reduction trees differ):

* ``fused=False``  one library call per reference operator: SpMV, dot, axpy, axpy, norm, xpay
  (8 kernel launches, SpMV + 96 B/row of vector traffic -- the "textbook unfused" count of SURVEY 8d);
* ``fused=True``   SpMV with the p.Ap partials in its epilogue, one update kernel for
  x += a p / r -= a Ap / sum r^2, then xpay: 5 launches, SpMV + ~72 B/row.
"""
from __future__ import annotations

from typing import List, Tuple

from .sparse import get_vector_plan, mul_, mul_dot_
from .vectors import HPCVector, cg_update_, dot, norm


def _torch():
    import torch
    return torch


def cg_fixed_iterations(A, b: HPCVector, iters: int, record_history: bool = True,
                        fused: bool = True) -> Tuple[HPCVector, List[float]]:
    """Textbook CG from x0 = 0, exactly ``iters`` iterations, no convergence exit.
    Returns (x, [||r_0||, ..., ||r_iters||]) (history read back once at the end)."""
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
    hist = torch.zeros(iters + 1, dtype=torch.float64, device=dev)
    norm(r, 2, out=rr[0:1])                    # out form leaves sum(r^2) on the device (8 B/elt)
    if record_history:
        hist[0:1].copy_(rr[0:1])
    cur = 0
    for it in range(iters):
        nxt = 1 - cur
        rr_cur, rr_nxt = rr[cur:cur + 1], rr[nxt:nxt + 1]
        if fused:
            mul_dot_(Ap, A, p, pAp)                                 # Ap = A*p, pAp = p.Ap
            cg_update_(x, r, p, Ap, 1.0, rr_cur, pAp, rr_nxt)       # x += a p; r -= a Ap; rr_new
        else:
            mul_(Ap, A, p)                                          # Ap = A*p
            dot(p, Ap, out=pAp)                                     # pAp
            x.axpy_(1.0, p, num=rr_cur, den=pAp)                    # x += (rr/pAp) p
            r.axpy_(-1.0, Ap, num=rr_cur, den=pAp)                  # r -= (rr/pAp) Ap
            norm(r, 2, out=rr_nxt)                                  # rr_new = sum(r^2)
        p.xpay_(r, 1.0, num=rr_nxt, den=rr_cur)                     # p = r + (rr_new/rr) p
        if record_history:
            hist[it + 1:it + 2].copy_(rr_nxt)
        cur = nxt
    h = hist.sqrt().cpu().tolist() if record_history else []
    return x, h
