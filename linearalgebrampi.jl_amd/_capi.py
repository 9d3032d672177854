"""ctypes binding of libhpcla_rocm.so (include/hpcla_rocm.h).

This is the Python twin of the ``@ccall`` stubs in INTEGRATION.md: same entry points, same
status convention (`status == 0 || error(...)`, cf. ext/HPCLinearAlgebraCUDAExt.jl:248-251).
There is no fallback: if the shared library is missing or fails to load, importing the package's
device layer raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhpcla_rocm.so")

OK = 0
LAYOUT_ROW = 0
LAYOUT_COL = 1
UNIQUE_ID_BYTES = 128
WINDOW_DESC_BYTES = 128
WINDOW_TABLE_ROWS = 4
COMM_NO_RCCL = 1
HALO_SINGLE_BUFFER = 1


class HPCLAError(RuntimeError):
    def __init__(self, fn: str, status: int, text: str):
        super().__init__(f"{fn} failed with status {status}: {text}")
        self.status = status


_lib = None

_i32 = ctypes.c_int
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64
_f64 = ctypes.c_double
_f32 = ctypes.c_float
_vp = ctypes.c_void_p

# name -> argtypes ; every function returns int status unless listed in _RESTYPES
_SIGNATURES = {
    "hpcla_version": [],
    "hpcla_last_error": [],
    "hpcla_device_count": [_vp],
    "hpcla_set_device": [_i32],
    "hpcla_device_info": [_i32, _vp, _vp, _i32],
    "hpcla_spmv_csr_f64_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "hpcla_spmv_csr_f64_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "hpcla_spmv_split_f64_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp],
    "hpcla_spmv_split_f64_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp],
    "hpcla_spmv_rows_per_block": [],
    "hpcla_spmv_longrows_work_bytes": [_i64],
    "hpcla_spmv_longrows_f64_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp, _vp],
    "hpcla_spmv_longrows_f64_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp, _vp],
    "hpcla_spmv_block_order_hint": [_vp, _i32],
    "hpcla_spmv_tune_block_order_f64_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp],
    "hpcla_spmv_tune_block_order_f64_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp],
    "hpcla_spmm_rows_per_block": [],
    "hpcla_spmm_runs_desc_bytes": [_i64],
    "hpcla_spmm_runs_build_i32": [_vp, _vp, _i64, _i64, _i32, _i64, _vp, _vp, _vp],
    "hpcla_spmm_runs_build_i64": [_vp, _vp, _i64, _i64, _i32, _i64, _vp, _vp, _vp],
    "hpcla_spmm_banded_blocks_i32": [_vp, _vp, _i64, _i64, _i32, _i64, _i32, _vp, _vp],
    "hpcla_spmm_banded_blocks_i64": [_vp, _vp, _i64, _i64, _i32, _i64, _i32, _vp, _vp],
    "hpcla_spmm_runs_k16_f64_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp],
    "hpcla_spmm_runs_k16_f64_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp],
    "hpcla_spmm_runs_colmajor_k16_f64_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _i64, _vp],
    "hpcla_spmm_runs_colmajor_k16_f64_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _i64, _vp],
    "hpcla_spmm_runs_colmajor_tune_block_order_f64_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _vp],
    "hpcla_spmm_runs_colmajor_tune_block_order_f64_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _vp],
    "hpcla_spmm_block_order_hint": [_vp, _i32],
    "hpcla_spmm_tune_block_order_f64_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32,
                                            _vp, _i64, _vp, _vp],
    "hpcla_spmm_tune_block_order_f64_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32,
                                            _vp, _i64, _vp, _vp],
    "hpcla_remap_i32": [_vp, _vp, _vp, _i64, _i32, _vp],
    "hpcla_remap_i64": [_vp, _vp, _vp, _i64, _i32, _vp],
    "hpcla_remap_i64_to_i32": [_vp, _vp, _vp, _i64, _i32, _vp],
    "hpcla_narrow_i64_to_i32": [_vp, _vp, _i64, _vp, _vp],
    "hpcla_classify_blocks_i32": [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp],
    "hpcla_classify_blocks_i64": [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp],
    "hpcla_spmm_csr_f64_i32": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _i32, _vp],
    "hpcla_spmm_csr_f64_i64": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _i32, _vp],
    "hpcla_spmm_split_f64_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_spmm_split_f64_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_spmm_split_ccol_f64_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_spmm_split_ccol_f64_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_spmm_split_colmajor_f64_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_spmm_split_colmajor_f64_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_halo_begin_strided_f64": [_vp, _vp, _i64, _i64, _vp, _vp],
    "hpcla_spmm_panel_f64_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp],
    "hpcla_spmm_panel_f64_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp],
    "hpcla_transpose_f64": [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _vp],
    "hpcla_gather_f64_i32": [_vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "hpcla_gather_f64_i64": [_vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "hpcla_comm_get_unique_id": [_vp],
    "hpcla_comm_init_rank": [_vp, _vp, _i32, _i32],
    "hpcla_comm_init_rank_ex": [_vp, _vp, _i32, _i32, _i32],
    "hpcla_comm_window_export": [_vp, _vp],
    "hpcla_comm_window_attach": [_vp, _vp],
    "hpcla_comm_status": [_vp, _vp],
    "hpcla_comm_window_selftest": [_vp, _f64, _vp],
    "hpcla_comm_window_detach": [_vp],
    "hpcla_halo_plan_detach": [_vp],
    "hpcla_device_identity": [_i32, _vp],
    "hpcla_halo_plan_export": [_vp, _vp, _vp],
    "hpcla_halo_plan_attach": [_vp, _vp, _vp],
    "hpcla_halo_status": [_vp, _vp],
    "hpcla_halo_plan_probe": [_vp, _i64, _vp, _vp, _i64, _vp, _vp],
    "hpcla_set_halo_mode": [_i32],
    "hpcla_comm_rank": [_vp, _vp],
    "hpcla_comm_size": [_vp, _vp],
    "hpcla_comm_destroy": [_vp],
    "hpcla_allreduce_f64": [_vp, _vp, _i64, _i32, _vp],
    "hpcla_exchange_ranges_f64": [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "hpcla_halo_plan_create": [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32],
    "hpcla_halo_plan_create_ex": [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32],
    "hpcla_halo_plan_chain": [_vp, _vp],
    "hpcla_halo_plan_destroy": [_vp],
    "hpcla_halo_ghost_ptr": [_vp, _vp, _vp],
    "hpcla_halo_begin": [_vp, _vp, _vp],
    "hpcla_halo_end": [_vp, _vp],
    "hpcla_spmv_dist_f64_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp],
    "hpcla_spmv_dist_f64_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp],
    "hpcla_spmv_dot_work_bytes": [_i64],
    "hpcla_spmv_dist_dot_f64_i32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp],
    "hpcla_spmv_dist_dot_f64_i64": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp],
    "hpcla_cg_update_f64": [_vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_cg_residual_f64": [_vp, _f64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_cg_direction_f64": [_f64, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "hpcla_cg_iterations_f64_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _i32, _vp],
    "hpcla_cg_iterations_f64_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _i32, _vp],
    "hpcla_colspace_work_bytes": [_i64],
    "hpcla_compress_columns_i32": [_vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp],
    "hpcla_compress_columns_i64": [_vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp],
    "hpcla_digest_i32": [_vp, _i64, _vp, _vp],
    "hpcla_digest_i64": [_vp, _i64, _vp, _vp],
    "hpcla_poisson2d_nnz": [_i64, _i64, _i64, _i64],
    "hpcla_gen_poisson2d": [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "hpcla_poisson3d_nnz": [_i64, _i64, _i64, _i64, _i64],
    "hpcla_gen_poisson3d": [_i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "hpcla_gemv_rowmajor_f64": [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp],
    "hpcla_gemv_t_work_bytes": [_i64, _i64],
    "hpcla_gemv_t_rowmajor_f64": [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "hpcla_spgemm_bin_cap": [_i32],
    "hpcla_spgemm_ub_i32": [_vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "hpcla_spgemm_ub_i64": [_vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "hpcla_spgemm_numeric_i32": [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "hpcla_spgemm_numeric_i64": [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "hpcla_spgemm_compact": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "hpcla_spgemm_numeric_mapped_f64": [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp],
    "hpcla_packed_create_i32": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _vp],
    "hpcla_packed_destroy": [_vp],
    "hpcla_packed_info": [_vp, _vp, _vp],
    "hpcla_spmv_packed_f64_i32": [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp],
    "hpcla_spmv_dist_packed_f64_i32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64,
                                       _vp, _i64, _vp, _vp, _vp],
    "hpcla_reduce_work_bytes": [],
    "hpcla_dot_f64": [_vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_nrm2sq_f64": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_asum_f64": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_amax_f64": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_maxval_f64": [_vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "hpcla_sum_f64": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_prod_f64": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_powsum_f64": [_vp, _vp, _i64, _f64, _vp, _vp, _vp],
    "hpcla_axpy_f64": [_f64, _vp, _vp, _vp, _vp, _i64, _vp],
    "hpcla_xpay_f64": [_vp, _f64, _vp, _vp, _vp, _i64, _vp],
    "hpcla_scale_f64": [_f64, _vp, _vp, _i64, _vp],
    "hpcla_divide_f64": [_vp, _f64, _vp, _i64, _vp],
    "hpcla_axpby_f64": [_f64, _vp, _f64, _vp, _vp, _i64, _vp],
    "hpcla_merge_combine_f64_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "hpcla_merge_combine_f64_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "hpcla_fill_uniform_f64": [_vp, _i64, _i64, _u64, _vp],
    # Float32 element type (csrc/f32.hip)
    "hpcla_spmv_csr_f32_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "hpcla_spmv_csr_f32_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "hpcla_spmv_split_f32_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp],
    "hpcla_spmv_split_f32_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp],
    "hpcla_spmm_csr_f32_i32": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _i32, _vp],
    "hpcla_spmm_csr_f32_i64": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _i32, _vp],
    "hpcla_spmm_split_f32_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_spmm_split_f32_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_spmv_dist_f32_i32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp],
    "hpcla_spmv_dist_f32_i64": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp],
    "hpcla_halo_begin_f32": [_vp, _vp, _vp, _vp],
    "hpcla_halo_begin_strided_f32": [_vp, _vp, _i64, _i64, _vp, _vp],
    "hpcla_spmm_split_colmajor_f32_i32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_spmm_split_colmajor_f32_i64": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp],
    "hpcla_transpose_f32": [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _vp],
    "hpcla_dot_f32": [_vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_nrm2sq_f32": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_asum_f32": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_amax_f32": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_sum_f32": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hpcla_maxval_f32": [_vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "hpcla_axpby_f32": [_f32, _vp, _f32, _vp, _vp, _i64, _vp],
    "hpcla_scale_f32": [_f32, _vp, _vp, _i64, _vp],
    "hpcla_divide_f32": [_vp, _f32, _vp, _i64, _vp],
}
_RESTYPES = {
    "hpcla_last_error": ctypes.c_char_p,
    "hpcla_reduce_work_bytes": _i64,
    "hpcla_spmv_dot_work_bytes": _i64,
    "hpcla_colspace_work_bytes": _i64,
    "hpcla_poisson2d_nnz": _i64,
    "hpcla_poisson3d_nnz": _i64,
    "hpcla_spgemm_bin_cap": _i64,
    "hpcla_gemv_t_work_bytes": _i64,
    "hpcla_spmm_runs_desc_bytes": _i64,
    "hpcla_spmv_longrows_work_bytes": _i64,
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load() -> ctypes.CDLL:
    """Load libhpcla_rocm.so.  Inside a PyTorch process torch must be imported first so that the
    HIP runtime (libamdhip64.so.7) and librccl.so.1 already mapped by torch are the ones the
    library binds to (one HIP runtime per process: torch tensors' device pointers and streams are
    then directly usable)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C linearalgebrampi.jl_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (maps torch's libamdhip64 / librccl first)
    except Exception:  # pragma: no cover - torch is plumbing, the library works without it
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


def last_error() -> str:
    s = load().hpcla_last_error()
    return s.decode("utf-8", "replace") if s else ""


def check(fn_name: str, status: int) -> None:
    if status != OK:
        raise HPCLAError(fn_name, status, last_error())


def call(fn_name: str, *args) -> None:
    """Call a status-returning entry point and raise HPCLAError on failure."""
    check(fn_name, getattr(load(), fn_name)(*args))
