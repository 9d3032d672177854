"""HPCVector on DeviceROCm (reference: src/vectors.jl).

``HPCVector{T,B}`` (src/vectors.jl:21-30): ``structural_hash`` (hash of the partition),
``partition`` (host), ``v`` (local slice, here a device buffer held as a torch CUDA tensor of the backend's element
type -- float64, or float32 for the reference's Float32 configurations (test/test_utils.jl:62-80), csrc/f32.hip -- torch
is only the allocator/stream provider), ``backend``.

Every arithmetic method launches kernels of libhpcla_rocm through the C ABI on torch's current
stream; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import numpy as np

from . import _capi
from .backends import (HPCBackend, assert_backends_compatible, comm_allgather, comm_rank,
                       comm_size)
from .partition import compute_partition_hash, uniform_partition


def _torch():
    import torch
    return torch


def current_stream_ptr() -> ctypes.c_void_p:
    return ctypes.c_void_p(_torch().cuda.current_stream().cuda_stream)


def dptr(t) -> ctypes.c_void_p:
    """Raw device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if t.device.type != "cuda":
        # every operator computes in the HIP library; a CPU-resident operand (the result of
        # to_backend(x, cpu_version(b))) is an error, never a silent host fallback
        raise TypeError("operand lives in host memory; this build has no CPU compute path "
                        "(move it with to_backend(x, rocm_backend), or use the reference package on the CPU)")
    return ctypes.c_void_p(t.data_ptr())


def sfx_of(backend: HPCBackend) -> str:
    """Suffix of the C entry points for the backend's element type (``_f64`` kernels, or csrc/f32.hip's ``_f32``)."""
    return "f32" if backend.T == np.dtype(np.float32) else "f64"


def torch_dtype_of(backend: HPCBackend):
    torch = _torch()
    return torch.float32 if backend.T == np.dtype(np.float32) else torch.float64


def f64_only(backend: HPCBackend, what: str) -> None:
    """The Float32 element type covers the reference's hot path (A*x, mul!, A*HPCMatrix, dot, norm, u+v, a*v); the
    widened rows and the fused CG pieces are Float64 entries."""
    if backend.T != np.dtype(np.float64):
        raise TypeError(f"{what}: offered for Float64 backends only (Float32 covers A*x, mul!, A*B with dense B, dot, norm, "
                        "sum, maximum / minimum, u+v, u-v, a*v, v/a, and CG composed from them)")


def _round_to(backend: HPCBackend, v: float) -> float:
    """A reduction's double result, rounded once to the backend's element type (dot / norm return T in the reference)."""
    return float(np.float32(v)) if backend.T == np.dtype(np.float32) else v


class _Scratch:
    """Per-device reduction workspace + result scalar (allocated once; the launch functions never
    allocate, cdna_hip_programming.md Guideline 9)."""
    _by_device = {}

    @classmethod
    def get(cls, device):
        torch = _torch()
        key = (device.type, device.index)
        if key not in cls._by_device:
            nbytes = _capi.load().hpcla_reduce_work_bytes()
            cls._by_device[key] = (torch.empty(nbytes // 8, dtype=torch.float64, device=device),
                                   torch.zeros(8, dtype=torch.float64, device=device))
        return cls._by_device[key]


class HPCVector:
    """Row-partitioned distributed vector.  Construct with :func:`HPCVector.from_global`
    (``HPCVector(v_global, backend)``, src/vectors.jl:119-129) or :func:`HPCVector_local`."""

    def __init__(self, structural_hash: bytes, partition: np.ndarray, v, backend: HPCBackend):
        self.structural_hash = structural_hash
        self.partition = np.asarray(partition, dtype=np.int64)
        self.v = v
        self.backend = backend

    # -- constructors ------------------------------------------------------------------------------
    @classmethod
    def from_global(cls, v_global, backend: HPCBackend, partition: Optional[np.ndarray] = None):
        torch = _torch()
        v_global = np.asarray(v_global, dtype=backend.T)
        nranks, rank = comm_size(backend.comm), comm_rank(backend.comm)
        if partition is None:
            partition = uniform_partition(len(v_global), nranks)
        lo, hi = int(partition[rank]), int(partition[rank + 1])
        local = torch.from_numpy(np.ascontiguousarray(v_global[lo:hi])).to(backend.torch_device)
        return cls(compute_partition_hash(partition), partition, local, backend)

    @classmethod
    def zeros(cls, partition: np.ndarray, backend: HPCBackend):
        torch = _torch()
        rank = comm_rank(backend.comm)
        n = int(partition[rank + 1] - partition[rank])
        return cls(compute_partition_hash(partition), partition,
                   torch.zeros(n, dtype=torch_dtype_of(backend), device=backend.torch_device), backend)

    def similar(self):
        torch = _torch()
        return HPCVector(self.structural_hash, self.partition, torch.empty_like(self.v), self.backend)

    def copy(self):
        return HPCVector(self.structural_hash, self.partition, self.v.clone(), self.backend)

    # -- shape ------------------------------------------------------------------------------------
    def __len__(self):
        return int(self.partition[-1])

    @property
    def local_length(self) -> int:
        return int(self.v.numel())

    # -- host views (parity checks only) -------------------------------------------------------------
    def local_values(self) -> np.ndarray:
        """test/test_utils.jl:235-243 ``local_values``: Array(v.v)."""
        return self.v.detach().cpu().numpy()

    def gather(self) -> np.ndarray:
        """``Vector(v)`` (src/HPCLinearAlgebra.jl:817-830): allgatherv of the whole vector to every
        rank's host -- used by parity checks, never on the hot path."""
        from .backends import CommSerial, _dist, _host_device
        loc = self.local_values()
        if isinstance(self.backend.comm, CommSerial):
            return loc
        torch = _torch()
        dist = _dist()
        comm = self.backend.comm
        dev = _host_device(comm)
        sizes = np.diff(self.partition)
        nmax = int(sizes.max()) if len(sizes) else 0
        pad = torch.zeros(nmax, dtype=torch_dtype_of(self.backend), device=dev)
        pad[:len(loc)] = torch.from_numpy(loc).to(dev)
        outs = [torch.empty_like(pad) for _ in range(comm_size(comm))]
        dist.all_gather(outs, pad, group=comm.group)
        return np.concatenate([o[:int(s)].cpu().numpy() for o, s in zip(outs, sizes)])

    # -- checks -------------------------------------------------------------------------------------
    def _same_partition(self, other: "HPCVector") -> None:
        """In-place / fused updates write into their operands and so need equal partitions."""
        assert_backends_compatible(self.backend, other.backend)
        if self.structural_hash != other.structural_hash:
            raise ValueError("HPCVector operands of an in-place update have different partitions; "
                             "repartition(v, u.partition) first")

    def _aligned(self, other: "HPCVector") -> "HPCVector":
        """``other`` on MY partition: the reference repartitions the second operand of ``u+v``,
        ``u-v`` and ``dot(x,y)`` when the partition hashes differ (src/vectors.jl:803-811, 870-876,
        887-893); here over RCCL, device to device (repartition.py)."""
        assert_backends_compatible(self.backend, other.backend)
        if self.structural_hash == other.structural_hash:
            return other
        if len(self) != len(other):
            raise ValueError(f"HPCVector length mismatch: {len(self)} vs {len(other)}")
        from .repartition import repartition_vector
        return repartition_vector(other, self.partition)

    # -- elementwise: u+v, u-v, -v, a*v, v/a (src/vectors.jl:868-903, 944-964) -----------------------
    def _axpby(self, a: float, other: "HPCVector", b: float) -> "HPCVector":
        other = self._aligned(other)
        out = self.similar()
        _capi.call(f"hpcla_axpby_{sfx_of(self.backend)}", float(a), dptr(self.v), float(b), dptr(other.v), dptr(out.v),
                   self.local_length, current_stream_ptr())
        return out

    def __add__(self, other):
        return self._axpby(1.0, other, 1.0)

    def __sub__(self, other):
        return self._axpby(1.0, other, -1.0)

    def __neg__(self):
        return self.__mul__(-1.0)

    def __mul__(self, a):
        if isinstance(a, HPCVector):
            return NotImplemented
        out = self.similar()
        _capi.call(f"hpcla_scale_{sfx_of(self.backend)}", float(a), dptr(self.v), dptr(out.v), self.local_length,
                   current_stream_ptr())
        return out

    __rmul__ = __mul__

    def __truediv__(self, a):
        out = self.similar()
        _capi.call(f"hpcla_divide_{sfx_of(self.backend)}", dptr(self.v), float(a), dptr(out.v), self.local_length,
                   current_stream_ptr())
        return out

    # -- fused updates (broadcast `dest .= x .+ a .* p`, src/vectors.jl:1203-1226) --------------------
    def axpy_(self, a: float, x: "HPCVector", num=None, den=None) -> "HPCVector":
        """self .= self .+ (a*num/den) .* x ; num/den optional device scalars (1-element tensors)."""
        self._same_partition(x)
        f64_only(self.backend, "axpy_")
        _capi.call("hpcla_axpy_f64", float(a), dptr(num), dptr(den), dptr(x.v), dptr(self.v),
                   self.local_length, current_stream_ptr())
        return self

    def xpay_(self, x: "HPCVector", a: float, num=None, den=None) -> "HPCVector":
        """self .= x .+ (a*num/den) .* self."""
        self._same_partition(x)
        f64_only(self.backend, "xpay_")
        _capi.call("hpcla_xpay_f64", dptr(x.v), float(a), dptr(num), dptr(den), dptr(self.v),
                   self.local_length, current_stream_ptr())
        return self


def cg_update_(x: HPCVector, r: HPCVector, p: HPCVector, Ap: HPCVector, a: float, num, den, rr_out):
    """Fused CG update: ``x .+= s .* p ; r .-= s .* Ap ; rr_out = sum(r.^2)`` with
    ``s = a*num/den`` read from device scalars -- one pass (48 B/elt) instead of two broadcasts and a
    norm (56 B/elt, src/vectors.jl:1203-1226, 758-765)."""
    x._same_partition(r), x._same_partition(p), x._same_partition(Ap)
    f64_only(x.backend, "cg_update_")
    work, _ = _Scratch.get(x.v.device)
    _capi.call("hpcla_cg_update_f64", x.backend.rccl, float(a), dptr(num), dptr(den), dptr(p.v), dptr(Ap.v),
               dptr(x.v), dptr(r.v), x.local_length, dptr(rr_out), dptr(work), current_stream_ptr())
    return rr_out


def cg_residual_(r: HPCVector, Ap: HPCVector, a: float, num, den, rr_out):
    """``r .-= s .* Ap ; rr_out = sum(r.^2)``, ``s = a*num/den`` from device scalars (24 B/elt)."""
    r._same_partition(Ap)
    f64_only(r.backend, "cg_residual_")
    work, _ = _Scratch.get(r.v.device)
    _capi.call("hpcla_cg_residual_f64", r.backend.rccl, float(a), dptr(num), dptr(den), dptr(Ap.v), dptr(r.v),
               r.local_length, dptr(rr_out), dptr(work), current_stream_ptr())
    return rr_out


def cg_direction_(x: HPCVector, p: HPCVector, r: HPCVector, a: float, a_num, a_den, b: float, b_num, b_den):
    """``x .+= s .* p ; p .= r .+ t .* p`` with ``s = a*a_num/a_den``, ``t = b*b_num/b_den`` (40 B/elt): the
    deferred x update of a CG iteration rides on the direction update, which reads the same p."""
    x._same_partition(p), x._same_partition(r)
    f64_only(x.backend, "cg_direction_")
    _capi.call("hpcla_cg_direction_f64", float(a), dptr(a_num), dptr(a_den), float(b), dptr(b_num), dptr(b_den),
               dptr(r.v), dptr(x.v), dptr(p.v), x.local_length, current_stream_ptr())


def HPCVector_local(v_local, backend: HPCBackend) -> HPCVector:
    """src/vectors.jl:76-94: partition inferred by an Allgather of the local sizes."""
    torch = _torch()
    if isinstance(v_local, np.ndarray):
        v_local = torch.from_numpy(np.ascontiguousarray(v_local, dtype=backend.T))
    v_local = v_local.to(device=backend.torch_device, dtype=torch_dtype_of(backend)).contiguous()
    sizes = comm_allgather(backend.comm, np.array([v_local.numel()], dtype=np.int64))
    partition = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return HPCVector(compute_partition_hash(partition), partition, v_local, backend)


def _host_scalar(t, backend) -> float:
    """Read a device scalar; a NaN may be the poison of an expired exchange wait -- ask before returning it."""
    v = float(t.item())
    if v != v:
        from .sparse import check_exchange_health
        check_exchange_health(backend)
    return v


# ---- reductions -----------------------------------------------------------------------------------
def _reduce(kind: str, x: HPCVector, y: Optional[HPCVector], out=None):
    """Launch the local reduction + RCCL all-reduce; returns the 1-element device tensor."""
    work, scal = _Scratch.get(x.v.device)
    if out is None:
        out = scal[:1]
    comm = x.backend.rccl
    s = current_stream_ptr()
    t = sfx_of(x.backend)                  # Float32 operands: the scalar is still formed and all-reduced in double (f32.hip)
    if kind == "dot":
        _capi.call(f"hpcla_dot_{t}", comm, dptr(x.v), dptr(y.v), x.local_length, dptr(out), dptr(work), s)
    elif kind == "nrm2sq":
        _capi.call(f"hpcla_nrm2sq_{t}", comm, dptr(x.v), x.local_length, dptr(out), dptr(work), s)
    elif kind == "asum":
        _capi.call(f"hpcla_asum_{t}", comm, dptr(x.v), x.local_length, dptr(out), dptr(work), s)
    elif kind == "amax":
        _capi.call(f"hpcla_amax_{t}", comm, dptr(x.v), x.local_length, dptr(out), dptr(work), s)
    else:  # pragma: no cover
        raise ValueError(kind)
    return out


def vsum(v: HPCVector, out=None):
    """``sum(v)`` (src/vectors.jl:838-845)."""
    work, scal = _Scratch.get(v.v.device)
    r = out if out is not None else scal[:1]
    _capi.call(f"hpcla_sum_{sfx_of(v.backend)}", v.backend.rccl, dptr(v.v), v.local_length, dptr(r), dptr(work),
               current_stream_ptr())
    return r if out is not None else _round_to(v.backend, _host_scalar(r, v.backend))


def prod(v: HPCVector) -> float:
    """``prod(v)`` (src/vectors.jl:853-858): local product (1 for an empty part), then an all-reduce with ``*``."""
    f64_only(v.backend, "prod")
    work, scal = _Scratch.get(v.v.device)
    _capi.call("hpcla_prod_f64", v.backend.rccl, dptr(v.v), v.local_length, dptr(scal[:1]), dptr(work), current_stream_ptr())
    return _host_scalar(scal[:1], v.backend)


def _maxval(v: HPCVector, negate: int) -> float:
    work, scal = _Scratch.get(v.v.device)
    _capi.call(f"hpcla_maxval_{sfx_of(v.backend)}", v.backend.rccl, dptr(v.v), v.local_length, negate, dptr(scal[:1]), dptr(work),
               current_stream_ptr())
    return _host_scalar(scal[:1], v.backend)


def maximum(v: HPCVector) -> float:
    """``maximum(v)`` (src/vectors.jl:815-824)."""
    return _maxval(v, 0)


def minimum(v: HPCVector) -> float:
    """``minimum(v)`` (src/vectors.jl:826-836)."""
    return -_maxval(v, 1)


def dot(x: HPCVector, y: HPCVector, out=None):
    """``dot(x, y)`` (src/vectors.jl:798-812).  Returns a Python float (host sync), or, when ``out``
    (1-element device tensor) is given, leaves the result on the device and returns ``out``."""
    y = x._aligned(y)
    r = _reduce("dot", x, y, out)
    return r if out is not None else _round_to(x.backend, _host_scalar(r, x.backend))


def norm(v: HPCVector, p: float = 2, out=None):
    """``norm(v, p)`` (src/vectors.jl:758-780): p=2 sums squares then sqrt, p=1 asum, p=Inf max.
    With ``out`` (1-element device tensor) nothing is read back: ``out`` receives the all-reduced
    quantity the reference feeds to its final step -- for p=2 that is the SUM OF SQUARES (the sqrt is
    the caller's), for p=1 / Inf the norm itself, for any other p the sum of |x|^p."""
    if p == 2:
        r = _reduce("nrm2sq", v, None, out)
        return r if out is not None else _round_to(v.backend, math.sqrt(_host_scalar(r, v.backend)))
    if p == 1:
        r = _reduce("asum", v, None, out)
        return r if out is not None else _round_to(v.backend, _host_scalar(r, v.backend))
    if p == math.inf:
        r = _reduce("amax", v, None, out)
        return r if out is not None else _host_scalar(r, v.backend)
    if not (p > 0):
        raise ValueError("norm: p must be positive")
    f64_only(v.backend, "norm(v, p) for p other than 1, 2, Inf")
    work, scal = _Scratch.get(v.v.device)                                        # general p (:774-779)
    r = out if out is not None else scal[:1]
    _capi.call("hpcla_powsum_f64", v.backend.rccl, dptr(v.v), v.local_length, float(p), dptr(r), dptr(work),
               current_stream_ptr())
    return r if out is not None else _host_scalar(r, v.backend) ** (1.0 / p)
