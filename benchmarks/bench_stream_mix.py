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
# the same byte mix moved by ONE hand-written streaming kernel (benchmarks/tune/stream_mix.hip)
import ctypes, os
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune", "libstream_mix.so")
if os.path.exists(so):
    lib = ctypes.CDLL(so)
    lib.stream_mix.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    total = (A.numel() + 2 * n) * 8 / 1e9
    for variant, name in ((0, "grid-stride"), (1, "per-workgroup slabs")):
        for blocks in (256 * 8, 256 * 16, 256 * 64, 131072):
            t = timed(lambda: lib.stream_mix(variant, A.data_ptr(), A.numel() * 8, B.data_ptr(), C.data_ptr(), n * 8, blocks, st))
            print(f"one kernel, {name:20s} {blocks:7d} workgroups: {t:.4f} ms = {total / t:.2f} TB/s of the {total:.2f} GB mix")
