"""Per-launch device time of the FIRST launches after setup (4096^2 headline problem): is the first-25-launch
penalty seen by `bench.py --steps 20 --warmup 5` a few very slow launches (TLB / first touch) or a ramp (clocks)?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hpcla_amd as hp
from hpcla_amd import workloads as wl
import bench


class A:  # args
    index = "i32"; host_setup = False


job = bench.Job(torch, None, 1, 0)
backend = hp.backend_rocm_serial(np.float64, np.int32)
idle_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
run = bench.PoissonRun(hp, wl, job, backend, A, 4096, False, 1, 0)
time.sleep(idle_ms * 1e-3)
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(80)]
for a, b in evs:
    a.record(); hp.mul_(run.y, run.A, run.x); b.record()
torch.cuda.synchronize()
t = [a.elapsed_time(b) for a, b in evs]
print(f"idle {idle_ms} ms before the first launch; per-launch ms:")
for i in range(0, 80, 10):
    print("  ", " ".join(f"{v:.4f}" for v in t[i:i + 10]))
print("mean first 5", np.mean(t[:5]), "launches 5..24", np.mean(t[5:25]), "launches 40..79", np.mean(t[40:]))
