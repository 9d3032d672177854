#!/usr/bin/env python3
"""One-off check: the 2-rank bench CG record (512 x 512 x 128 global, 100 iterations) printed
residual_last = 4887.44 > residual_first = 3344.3.  Is that the matrix (CG's residual norm is not monotone)
or an exchange defect?  Run the SAME global problem on ONE GPU and compare."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpcla_amd as hp
from hpcla_amd import workloads as wl
from benchmarks.extra_workloads import device_stencil
b = hp.backend_rocm_serial(np.float64, np.int32)
N, nz = 512, 128
n = N * N * nz
A = device_stencil(hp, torch, b, (N, N, nz), 0, n)
rhs = hp.HPCVector.zeros(A.row_partition, b)
hp._capi.call("hpcla_fill_uniform_f64", rhs.v.data_ptr(), 0, n, wl.SEED_RHS, torch.cuda.current_stream().cuda_stream)
x, hist = hp.cg_fixed_iterations(A, rhs, 100)
print("one GPU, 512x512x128: residual_first %.9f residual_last %.9f  min %.3f at it %d" % (hist[0], hist[-1], min(hist), int(np.argmin(hist))))
print("every 10th:", [round(h, 2) for h in hist[::10]])
