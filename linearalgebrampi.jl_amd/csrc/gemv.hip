// gemv.hip -- dense row-partitioned mat-vec y = A_local * x_full (SURVEY.md 8f "next" rank 4).
//
// Reference: Base.:*(A::HPCMatrix, x::HPCVector) / mul! (src/dense.jl:614-658): a
// DenseMatrixVectorPlan gathers the WHOLE vector x on every rank (CPU-staged MPI), then
// `LinearAlgebra.mul!(y.v, A.A, plan.gathered)` -- a BLAS gemv whose summation order is
// implementation-defined.  Here x arrives through the same GPU-resident RCCL halo plan as the SpMV
// (every rank sends its slice to every other rank); the kernel reads x in place as three segments
// [slices of lower ranks | own slice | slices of higher ranks] -- no assembled copy.
//
// Pure HBM stream of A (8 B per entry, 2 flop): one wavefront per row, 16-byte loads, shuffle tree.
// Local block is ROW-major (device-native layout of HPCMatrix, see dense.py).  Parity: tolerance
// (tree order vs BLAS order), 1e-12 relative to |A||x|.
#include "common.h"

namespace hpcla {

__device__ __forceinline__ double seg_dot(const double *__restrict__ a, const double *__restrict__ x,
                                          int64_t n, int lane)
{
    double acc = 0.0;
    // 16-byte loads when both pointers are 16-byte aligned, scalar otherwise
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(x)) & 15) == 0) {
        const double2 *a2 = reinterpret_cast<const double2 *>(a);
        const double2 *x2 = reinterpret_cast<const double2 *>(x);
        const int64_t n2 = n / 2;
        for (int64_t i = lane; i < n2; i += 64) {
            const double2 av = a2[i], xv = x2[i];
            acc += av.x * xv.x;
            acc += av.y * xv.y;
        }
        if ((n & 1) && lane == 0) acc += a[n - 1] * x[n - 1];
    } else {
        for (int64_t i = lane; i < n; i += 64) acc += a[i] * x[i];
    }
    return acc;
}

__global__ __launch_bounds__(256) void gemv_rowmajor_kernel(const double *__restrict__ A, int64_t lda,
                                                            int64_t nrows, const double *__restrict__ x_lo,
                                                            int64_t n_lo, const double *__restrict__ x_own,
                                                            int64_t n_own, const double *__restrict__ x_hi,
                                                            int64_t n_hi, double *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const double *a = A + row * lda;
    double acc = 0.0;
    if (n_lo > 0) acc += seg_dot(a, x_lo, n_lo, lane);
    if (n_own > 0) acc += seg_dot(a + n_lo, x_own, n_own, lane);
    if (n_hi > 0) acc += seg_dot(a + n_lo + n_own, x_hi, n_hi, lane);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) y[row] = acc;
}

// ---- y_partial = A_local^T * x_local (transpose(A) * x, src/dense.jl:1210-1261) -------------------
// The reference multiplies the local block transposed by the local slice of x and all-reduces the
// ncols partial sums on the host.  Row-major A: consecutive columns are consecutive addresses, so a
// wavefront covers W = min(64, pow2 >= ncols) columns x (64/W) rows per pass (a tall-skinny block
// with ncols == W is then read as one contiguous stream); 4 wavefronts per workgroup interleave rows.
// Deterministic: fixed-order LDS reduction per workgroup into partial[chunk][col], then stage 2 sums
// the chunks in ascending order.  HBM-bound: 8 B per matrix entry, x re-read once per column tile.
constexpr int GEMVT_THREADS = 256;

__global__ __launch_bounds__(GEMVT_THREADS) void gemv_t_stage1(const double *__restrict__ A, int64_t lda,
                                                               int64_t nrows, int64_t ncols,
                                                               const double *__restrict__ x, int W,
                                                               int64_t rows_per_chunk,
                                                               double *__restrict__ partial)
{
    __shared__ double red[GEMVT_THREADS];
    const int tid = threadIdx.x;
    const int c = tid % W;               // column within the tile
    const int rl = tid / W;              // row phase, 0 .. 256/W - 1
    const int rstep = GEMVT_THREADS / W;
    const int64_t col = (int64_t)blockIdx.x * W + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = min(nrows, r0 + rows_per_chunk);
    double acc = 0.0;
    if (col < ncols) {
        int64_t r = r0 + rl;
        for (; r + 3 * rstep < r1; r += 4 * rstep) {      // 4 independent loads in flight
            const double a0 = A[r * lda + col], a1 = A[(r + rstep) * lda + col];
            const double a2 = A[(r + 2 * rstep) * lda + col], a3 = A[(r + 3 * rstep) * lda + col];
            const double x0 = x[r], x1 = x[r + rstep], x2 = x[r + 2 * rstep], x3 = x[r + 3 * rstep];
            acc += a0 * x0;
            acc += a1 * x1;
            acc += a2 * x2;
            acc += a3 * x3;
        }
        for (; r < r1; r += rstep) acc += A[r * lda + col] * x[r];
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < W && col < ncols) {
        double s = red[tid];
        for (int k = 1; k < rstep; ++k) s += red[k * W + tid];   // ascending row phase
        partial[(int64_t)blockIdx.y * ncols + col] = s;
    }
}

__global__ __launch_bounds__(256) void gemv_t_stage2(const double *__restrict__ partial, int64_t nchunks,
                                                     int64_t ncols, double *__restrict__ y)
{
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    double s = 0.0;
    for (int64_t k = 0; k < nchunks; ++k) s += partial[k * ncols + col];
    y[col] = s;
}

static int64_t gemv_t_rows_per_chunk(int64_t nrows, int64_t ncols)
{
    // enough workgroups to fill 256 CUs several times over, but chunks of at least 64 rows
    int W = 1;
    while (W < 64 && W < ncols) W <<= 1;
    const int64_t tiles = (ncols + W - 1) / W;
    int64_t want_chunks = (4096 + tiles - 1) / (tiles > 0 ? tiles : 1);
    if (want_chunks < 1) want_chunks = 1;
    int64_t rpc = (nrows + want_chunks - 1) / want_chunks;
    if (rpc < 64) rpc = 64;
    return rpc;
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int64_t hpcla_gemv_t_work_bytes(int64_t nrows, int64_t ncols)
{
    if (nrows <= 0 || ncols <= 0) return 8;
    const int64_t rpc = gemv_t_rows_per_chunk(nrows, ncols);
    const int64_t nchunks = (nrows + rpc - 1) / rpc;
    return nchunks * ncols * (int64_t)sizeof(double);
}

HPCLA_API int hpcla_gemv_t_rowmajor_f64(const double *A, int64_t lda, int64_t nrows, int64_t ncols,
                                        const double *x, double *y_full, void *work, void *stream)
{
    if (nrows < 0 || ncols < 0 || lda < ncols) return set_error(HPCLA_ERR_INVALID, "gemv_t: bad sizes");
    if (ncols == 0) return HPCLA_OK;
    if (!y_full) return set_error(HPCLA_ERR_INVALID, "gemv_t: null output");
    hipStream_t s = as_stream(stream);
    if (nrows == 0) {                                     // empty local block: the partial sum is zero
        HPCLA_CHECK_HIP(hipMemsetAsync(y_full, 0, (size_t)ncols * sizeof(double), s));
        return HPCLA_OK;
    }
    if (!A || !x || !work) return set_error(HPCLA_ERR_INVALID, "gemv_t: null pointer");
    int W = 1;
    while (W < 64 && W < ncols) W <<= 1;
    const int64_t rpc = gemv_t_rows_per_chunk(nrows, ncols);
    const int64_t nchunks = (nrows + rpc - 1) / rpc;
    const int64_t tiles = (ncols + W - 1) / W;
    if (nchunks > 65535 || tiles > 0x7fffffff) return set_error(HPCLA_ERR_UNSUPPORTED, "gemv_t: grid too large");
    gemv_t_stage1<<<dim3((uint32_t)tiles, (uint32_t)nchunks), GEMVT_THREADS, 0, s>>>(
        A, lda, nrows, ncols, x, W, rpc, static_cast<double *>(work));
    HPCLA_CHECK_LAUNCH();
    gemv_t_stage2<<<(uint32_t)((ncols + 255) / 256), 256, 0, s>>>(static_cast<const double *>(work), nchunks,
                                                                ncols, y_full);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}



HPCLA_API int hpcla_gemv_rowmajor_f64(const double *A, int64_t lda, int64_t nrows, const double *x_lo,
                                      int64_t n_lo, const double *x_own, int64_t n_own,
                                      const double *x_hi, int64_t n_hi, double *y, void *stream)
{
    if (nrows < 0 || n_lo < 0 || n_own < 0 || n_hi < 0 || lda < n_lo + n_own + n_hi)
        return set_error(HPCLA_ERR_INVALID, "gemv: bad sizes");
    if (nrows == 0) return HPCLA_OK;
    if (!y || (n_lo + n_own + n_hi > 0 && !A)) return set_error(HPCLA_ERR_INVALID, "gemv: null pointer");
    if ((n_lo > 0 && !x_lo) || (n_own > 0 && !x_own) || (n_hi > 0 && !x_hi))
        return set_error(HPCLA_ERR_INVALID, "gemv: null x segment");
    HPCLA_CHECK_GRID((nrows + 3) / 4, "gemv");
    gemv_rowmajor_kernel<<<(uint32_t)((nrows + 3) / 4), 256, 0, as_stream(stream)>>>(
        A, lda, nrows, x_lo, n_lo, x_own, n_own, x_hi, n_hi, y);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}
