"""linearalgebrampi.jl_amd -- MI355X-native DeviceROCm backend for the HPCLinearAlgebra.jl
(sloisel/LinearAlgebraMPI.jl) distributed SpMV / SpMM / CG hot path.

Layout
  csrc/              hand-written HIP kernels + the C ABI (include/hpcla_rocm.h) -> libhpcla_rocm.so
  _capi.py           ctypes binding of the C ABI (no fallback)
  backends.py        HPCBackend{T,Ti,Device,Comm,Solver}, DeviceROCm, comm_* primitives
  partition.py       uniform_partition, structural hashes
  vectors.py         HPCVector, dot, norm, fused updates
  sparse.py          HPCSparseMatrix, VectorPlan (host lists + device plan), A*x, mul!
  dense.py           HPCMatrix, A*B (SpMM)
  cg.py              fixed-iteration CG harness
  transpose.py matmat.py addition.py repartition.py   the SURVEY 8f "next" rows and their plans

The directory name contains a dot, so it is imported through the top-level alias module
``hpcla_amd`` (``import hpcla_amd as hp``).
"""
from . import _capi
from .backends import (AbstractComm, AbstractDevice, CommSerial, CommTorch, DeviceCPU, DeviceROCm, HPCBackend,
                       SolverNone, assert_backends_compatible, backend_rocm_mpi,
                       backend_rocm_serial, backends_compatible, comm_exchange_arrays, comm_rank, comm_size,
                       cpu_version, eltype_backend, indextype_backend)
from .partition import (compute_partition_hash, compute_structural_hash, owner_of,
                        uniform_partition)
from .vectors import HPCVector, HPCVector_local, cg_direction_, cg_residual_, cg_update_, dot, maximum, minimum, norm, prod, vsum
from .sparse import (HPCSparseMatrix, HPCSparseMatrix_from_global, HPCSparseMatrix_local,
                     HPCSparseMatrix_local_device,
                     HostVectorPlan, VectorPlan, build_host_vector_plan, cache_sizes,
                     ExchangeTimeout, check_exchange_health,
                     clear_plan_cache, execute_plan, get_vector_plan, mul_, mul_dot_, split_column_map)
from .dense import (HPCMatrix, HPCMatrix_local, TransposedHPCMatrix, clear_dense_plan_cache, clear_spmm_cache,
                    dense_matvec, dense_matvec_t, spmm, spmm_block_order_of, spmm_exchange_bytes, spmm_runs_fit_of)
from .matmat import clear_matrix_plan_cache, get_matrix_plan, spgemm
from .cg import CGGraphPair, CGWorkspace, cg_fixed_iterations, cg_iterate, cg_setup
from .convert import to_backend
from .transpose import (HostTransposeStructure, TransposedHPCSparseMatrix, TransposedHPCVector, TransposePlan,
                        adjoint, clear_transpose_plan_cache, get_transpose_plan, transpose)
from .addition import add_scaled_identity, sparse_add
from .repartition import (RangePlan, SparseRepartitionPlan, clear_repartition_cache, exchange_ranges,
                          get_sparse_repartition_plan, get_vector_repartition_plan, repartition)

__all__ = [n for n in dir() if not n.startswith("_")] + ["_capi"]
