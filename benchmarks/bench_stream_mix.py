#!/usr/bin/env python3
"""Streaming ceilings of this box for the byte mixes of the SpMM workload (poisson2d_spmm: A 0.54 GB read,
B 1.07 GB read, C 1.07 GB written): a device-to-device copy (50 % writes), a read-only pass, and a
write-only fill, each timed with HIP events over 20 repetitions.  torch ops only -- a yardstick, not product code."""
import torch

def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

n = 8_388_608 * 16                       # doubles in B (and C)
B = torch.rand(n, dtype=torch.float64, device="cuda")
C = torch.empty_like(B)
A = torch.rand(67_000_000, dtype=torch.float64, device="cuda")   # 0.54 GB
t_copy = timed(lambda: C.copy_(B))
t_fill = timed(lambda: C.fill_(1.0))
t_readB = timed(lambda: B.sum())
t_readA = timed(lambda: A.sum())
gb = n * 8 / 1e9
print(f"copy  B->C ({2*gb:.2f} GB moved): {t_copy:.4f} ms = {2*gb/t_copy:.2f} TB/s")
print(f"fill  C    ({gb:.2f} GB written): {t_fill:.4f} ms = {gb/t_fill:.2f} TB/s")
print(f"sum   B    ({gb:.2f} GB read):    {t_readB:.4f} ms = {gb/t_readB:.2f} TB/s")
print(f"sum   A    ({A.numel()*8/1e9:.2f} GB read):    {t_readA:.4f} ms = {A.numel()*8/1e9/t_readA:.2f} TB/s")
print(f"SpMM byte mix as separate streams: copy + read A = {t_copy + t_readA:.4f} ms  (the kernel must also gather B rows 5x through L2)")
