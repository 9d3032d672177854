#!/usr/bin/env python3
"""Per-kernel rates of the CG building blocks (dot, nrm2sq, axpy, xpay, scale, axpby) at the
config-4 per-GPU vector length (16 777 216) -- HIP-event timing, interleaved rounds."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import hpcla_amd as hp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_777_216
    b = hp.backend_rocm_serial(np.float64, np.int32)
    part = np.array([0, n])
    mk = lambda seed: hp.HPCVector.from_global(np.random.default_rng(seed).random(n), b, partition=part)
    x, y, z = mk(1), mk(2), mk(3)
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    one = torch.ones(1, dtype=torch.float64, device="cuda")
    ops = {
        "dot (16 B/elt)": (16, lambda: hp.dot(x, y, out=out)),
        "nrm2sq (8 B/elt)": (8, lambda: hp.norm(x, 2, out=out)),
        "axpy y+=a*x (24 B/elt)": (24, lambda: y.axpy_(1e-9, x, num=one, den=one)),
        "xpay y=x+a*y (24 B/elt)": (24, lambda: y.xpay_(x, 0.999, num=one, den=one)),
        "axpby z=a*x+b*y (24 B/elt)": (24, lambda: x._axpby(1.0, y, 1.0)),
        "scale y=a*x (16 B/elt)": (16, lambda: x * 2.0),
    }
    res = {k: [] for k in ops}
    for rnd in range(6):
        for k, (_, fn) in ops.items():
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            a.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(e) / 20)
    print(f"# n = {n}")
    for k, (bpe, _) in ops.items():
        med = float(np.median(res[k][1:]))
        print(f"{k:32s} {med*1e3:9.1f} us   {bpe*n/med/1e6:8.1f} GB/s   {bpe*n/med/1e6/8000:6.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
