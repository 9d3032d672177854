"""``to_backend`` and the array-conversion hooks (reference: src/HPCLinearAlgebra.jl:316-378,
src/sparse.jl:869; hooks added by a device extension: ext/HPCLinearAlgebraCUDAExt.jl:129-187).

``to_backend(x, backend)`` moves the DEVICE-resident arrays of a vector / dense matrix / sparse matrix
to ``backend.device`` and re-labels the object; partitions, hashes and the host structure arrays are
shared, exactly like the reference (``cached_transpose`` is dropped, :369).  Device -> CPU is what the
reference's GPU tests compare on (``to_backend(x, cpu_version(backend))``, test/test_utils.jl:203-207);
CPU -> device is the upload path of the constructors."""
from __future__ import annotations

import numpy as np

from .backends import DeviceCPU, DeviceROCm, HPCBackend
from .dense import HPCMatrix
from .sparse import HPCSparseMatrix
from .vectors import HPCVector


def _torch():
    import torch
    return torch


def _convert_array(a, backend: HPCBackend):
    """``_convert_array(v, device)`` (src/HPCLinearAlgebra.jl:316-320 + the extension's methods): identity
    when the array already lives on the target device, else a copy there."""
    torch = _torch()
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    dev = backend.torch_device
    if a.device == dev:
        return a
    return a.to(dev)


def _to_target_device(idx, backend: HPCBackend):
    """``_to_target_device(v::Vector{Ti}, device)`` (src/sparse.jl:869): structure arrays for the kernels."""
    return _convert_array(idx, backend)


def to_backend(obj, backend: HPCBackend):
    if not isinstance(backend.device, (DeviceCPU, DeviceROCm)):
        raise TypeError(f"to_backend: unknown device {backend.device!r}")
    if getattr(obj, "backend", None) is not None and obj.backend.T != backend.T:
        raise TypeError(f"to_backend: element type {obj.backend.T} -> {backend.T} (to_backend moves arrays, it does not convert them)")
    if isinstance(obj, HPCVector):                                   # :337-340
        return HPCVector(obj.structural_hash, obj.partition, _convert_array(obj.v, backend), backend)
    if isinstance(obj, HPCMatrix):                                   # :347-350
        out = HPCMatrix(obj.row_partition, obj.col_partition, _convert_array(obj.A, backend), backend)
        out.structural_hash = obj.structural_hash
        return out
    if isinstance(obj, HPCSparseMatrix):                             # :358-378
        if backend.Ti != obj.Ti:
            raise TypeError(f"to_backend: index type {obj.Ti} -> {backend.Ti} (the reference keeps Ti)")
        rowptr_t = _to_target_device(obj.rowptr_target if obj._rowptr is None else obj.rowptr, backend)
        out = HPCSparseMatrix(obj.row_partition, obj.col_partition, obj.col_indices, obj._rowptr, obj._colval,
                              _convert_array(obj.nzval, backend), rowptr_t, backend)
        if obj._colval_target is not None:
            out._colval_target = _to_target_device(obj._colval_target, backend)
        out.structural_hash = obj.structural_hash
        out.cached_transpose = None                                  # invalidated like the reference (:369)
        return out
    raise TypeError(f"to_backend: unsupported object {type(obj).__name__}")
