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

}  // namespace hpcla

using namespace hpcla;

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
    gemv_rowmajor_kernel<<<(uint32_t)((nrows + 3) / 4), 256, 0, as_stream(stream)>>>(
        A, lda, nrows, x_lo, n_lo, x_own, n_own, x_hi, n_hi, y);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}
