// colmajor.hip -- Float64 A * B on the CALLER's column-major blocks, and the exchange that goes with it (gfx950).
//
// The reference's dense block is a Julia Matrix: column-major (src/dense.jl:63).  The tuned SpMM kernels of spmm.hip work on
// row-major rows (one 128-byte line per B row at k = 16), so a column-major caller pays two layout conversions around the
// product: on the 5-point matrix x 16 (4096 x 2048 rows) 0.43 ms + 0.57 ms around a 0.52 ms product -- 1.52 ms, 0.22 of the
// HBM peak by the product's own bytes (benchmarks/bench_colmajor.py, profiles/r04_colmajor.log).  With LANES = ROWS
// (rowgather_t.h) a column-major operand needs no conversion: the 64 rows of a wave read ONE contiguous run of a column per
// gather instruction wherever the matrix is banded (every stencil), the result leaves as 512-byte runs, A is staged once per
// 16 columns.  Each C(r, c) is one lane's sequential sum in stored order with separate multiply and add: the bits of the
// reference's column loop (src/sparse.jl:2391-2413) and of the row-major kernels.
//
// For unstructured rows (config 5's random pattern) column-major is the wrong layout whatever the kernel -- every (entry,
// column) pair touches its own line, 16 line fetches per stored entry where a row-major row costs one -- so callers keep the
// conversions there and use these entries where the structure is banded (the run-descriptor fit of hpcla_spmm_runs_build_*
// is the test the Julia extension applies).
//
// Distributed form: B_own and C stay column-major; the ghost rows are the halo plan's ordinary row-major segment (k doubles
// per ghost row).  hpcla_halo_begin_strided_f64 stages the rows a plan SENDS from the strided operand into a row-major
// staging block (a few thousand rows on a slab) and posts the ordinary exchange from there.
#include <stdlib.h>

#include "comm_internal.h"
#include "common.h"
#include "rowgather_t.h"

namespace hpcla {

template <typename I>
static int colmajor_launch(const I *rowptr, const I *colval, const double *nzval, const DenseOperand<double> &b, double *C,
                           int64_t c_rs, int64_t c_cs, int64_t nrows, int64_t nnz, int k, int index_base,
                           const int32_t *block_list, int64_t n_blocks, void *stream, const char *who)
{
    if (nrows < 0 || nnz < 0 || k < 0) return set_error(HPCLA_ERR_INVALID, "%s: negative size", who);
    if (index_base != 0 && index_base != 1) return set_error(HPCLA_ERR_INVALID, "%s: index_base must be 0 or 1", who);
    if (nrows == 0 || k == 0) return HPCLA_OK;
    if (!rowptr || !C) return set_error(HPCLA_ERR_INVALID, "%s: null rowptr / C", who);
    if (nnz > 0 && (!colval || !nzval || !b.own)) return set_error(HPCLA_ERR_INVALID, "%s: null colval / nzval / B with nnz > 0", who);
    const int64_t all_blocks = (nrows + F_RPB - 1) / F_RPB;
    int64_t launch_blocks = all_blocks;
    if (block_list) {
        if (n_blocks < 0 || n_blocks > all_blocks) return set_error(HPCLA_ERR_INVALID, "%s: n_blocks out of range", who);
        launch_blocks = n_blocks;
    }
    if (launch_blocks == 0) return HPCLA_OK;
    HPCLA_CHECK_GRID(launch_blocks, who);
    const int vec_ok = stage_vec_ok<double, I>(colval, nzval);
    const bool split = b.ghost != nullptr;
    hipStream_t s = as_stream(stream);
    const int nt = c_rs == 1;                   // contiguous result runs: stored non-temporally (written once, read by nothing;
                                                // measured neutral here, 0.6346 / 0.6352 ms with / without)
#define HPCLA_COLMAJOR(KC, UR)                                                                                              \
    do {                                                                                                                    \
        const int groups = (k + KC - 1) / KC;                                                                               \
        if (groups > 65535) return set_error(HPCLA_ERR_UNSUPPORTED, "%s: more than %d columns", who, 65535 * KC);           \
        dim3 grid((uint32_t)launch_blocks, (uint32_t)groups), block(F_RPB);                                                 \
        if (split)                                                                                                          \
            rowgather_kernel<double, I, true, KC, UR><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, c_rs, c_cs, nrows, \
                                                                            nnz, index_base, k, block_list, vec_ok, nt);    \
        else                                                                                                                \
            rowgather_kernel<double, I, false, KC, UR><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, c_rs, c_cs,      \
                                                                             nrows, nnz, index_base, k, block_list, vec_ok, \
                                                                             nt);                                           \
    } while (0)
    // columns per workgroup (A is staged once per group): HPCLA_COLMAJOR_KC = 4 | 8 | 16 overrides the measured default
    static const int kc_env = [] {
        const char *e = getenv("HPCLA_COLMAJOR_KC");
        return e ? atoi(e) : 0;
    }();
    static const int ur_env = [] {
        const char *e = getenv("HPCLA_COLMAJOR_UR");
        return e ? atoi(e) : 0;
    }();
    const int kc = kc_env == 4 || kc_env == 8 || kc_env == 16 ? kc_env : (k <= 4 ? 4 : (k <= 8 ? 8 : 16));
    if (kc == 4) HPCLA_COLMAJOR(4, 4);
    else if (kc == 8 && ur_env == 2) HPCLA_COLMAJOR(8, 2);
    else if (kc == 8) HPCLA_COLMAJOR(8, 4);
    else if (ur_env == 1) HPCLA_COLMAJOR(16, 1);
    else if (ur_env == 3) HPCLA_COLMAJOR(16, 3);
    else if (ur_env == 4) HPCLA_COLMAJOR(16, 4);
    else HPCLA_COLMAJOR(16, 2);
#undef HPCLA_COLMAJOR
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

// C = A * B with B and C in column-major blocks (spmm.hip routes hpcla_spmm_csr_f64_* here when both layouts are COL)
int spmm_colmajor_i32(const int32_t *rowptr, const int32_t *colval, const double *nzval, const double *B, int64_t ldb, double *C,
                      int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base, void *stream)
{
    const DenseOperand<double> b{B, 1, ldb, nullptr, 0, 0, 0};
    return colmajor_launch<int32_t>(rowptr, colval, nzval, b, C, 1, ldc, nrows, nnz, k, index_base, nullptr, 0, stream, "spmm_csr");
}
int spmm_colmajor_i64(const int64_t *rowptr, const int64_t *colval, const double *nzval, const double *B, int64_t ldb, double *C,
                      int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base, void *stream)
{
    const DenseOperand<double> b{B, 1, ldb, nullptr, 0, 0, 0};
    return colmajor_launch<int64_t>(rowptr, colval, nzval, b, C, 1, ldc, nrows, nnz, k, index_base, nullptr, 0, stream, "spmm_csr");
}

// stage[idx * w + c] = (double) x[idx * rs + c * cs] at the plan's send positions
template <typename T, typename I>
__global__ __launch_bounds__(256) void stage_at_kernel(const T *__restrict__ x, int64_t rs, int64_t cs, const I *__restrict__ idx,
                                                       double *__restrict__ stage, int64_t n_idx, int w)
{
    const int64_t total = n_idx * w;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = (int64_t)idx[e / w];
        const int c = (int)(e % w);
        stage[i * w + c] = (double)x[i * rs + c * cs];
    }
}

template <typename T>
int halo_begin_strided(hpcla_halo_plan_t *plan, const T *x, int64_t x_rs, int64_t x_cs, double *stage, void *stream,
                       const char *who)
{
    if (!plan) return set_error(HPCLA_ERR_INVALID, "%s: null plan", who);
    if (plan->send_ranks.empty() && plan->recv_ranks.empty()) return HPCLA_OK;
    if (plan->n_send_total > 0) {
        if (!x || !stage) return set_error(HPCLA_ERR_INVALID, "%s: null x / stage", who);
        if (x_rs < 1 || (plan->width > 1 && x_cs < 1)) return set_error(HPCLA_ERR_INVALID, "%s: strides must be positive", who);
        const int64_t total = plan->n_send_total * plan->width;
        int64_t g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        if (plan->idx_is_i64)
            stage_at_kernel<T, int64_t><<<(uint32_t)g, 256, 0, as_stream(stream)>>>(x, x_rs, x_cs, (const int64_t *)plan->send_idx, stage,
                                                                                  plan->n_send_total, plan->width);
        else
            stage_at_kernel<T, int32_t><<<(uint32_t)g, 256, 0, as_stream(stream)>>>(x, x_rs, x_cs, (const int32_t *)plan->send_idx, stage,
                                                                                  plan->n_send_total, plan->width);
        HPCLA_CHECK_LAUNCH();
    }
    return hpcla_halo_begin(plan, stage, stream);
}
template int halo_begin_strided<float>(hpcla_halo_plan_t *, const float *, int64_t, int64_t, double *, void *, const char *);
template int halo_begin_strided<double>(hpcla_halo_plan_t *, const double *, int64_t, int64_t, double *, void *, const char *);

}  // namespace hpcla

using namespace hpcla;

#define HPCLA_COLMAJOR_SPLIT(SFX, ITYPE)                                                                                    \
    HPCLA_API int hpcla_spmm_split_colmajor_f64_##SFX(const ITYPE *rowptr, const ITYPE *colval_split, const double *nzval,  \
                                                      const double *B_own, int64_t ldb_own, const double *B_ghost,          \
                                                      int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc,             \
                                                      int64_t nrows, int64_t nnz, int k, int index_base,                    \
                                                      const int32_t *block_list, int64_t n_blocks, void *stream)            \
    {                                                                                                                       \
        if (n_own < 0) return set_error(HPCLA_ERR_INVALID, "spmm_split_colmajor: negative n_own");                          \
        if (k > 1 && (ldb_own < n_own || ldc < nrows))                                                                      \
            return set_error(HPCLA_ERR_INVALID, "spmm_split_colmajor: leading dimension smaller than the row count");       \
        if (B_ghost && ldb_ghost < k) return set_error(HPCLA_ERR_INVALID, "spmm_split_colmajor: ldb_ghost < k");            \
        const DenseOperand<double> b{B_own, 1, ldb_own, B_ghost, ldb_ghost, 1, n_own};                                      \
        return colmajor_launch<ITYPE>(rowptr, colval_split, nzval, b, C, 1, ldc, nrows, nnz, k, index_base, block_list,     \
                                      n_blocks, stream, "spmm_split_colmajor");                                             \
    }
HPCLA_COLMAJOR_SPLIT(i32, int32_t)
HPCLA_COLMAJOR_SPLIT(i64, int64_t)

HPCLA_API int hpcla_halo_begin_strided_f64(hpcla_halo_plan_t *plan, const double *x, int64_t x_rs, int64_t x_cs, double *stage,
                                           void *stream)
{
    return halo_begin_strided<double>(plan, x, x_rs, x_cs, stage, stream, "halo_begin_strided_f64");
}
