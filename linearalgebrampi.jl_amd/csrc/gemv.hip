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

// Tall-skinny blocks (ncols <= 128: multi-vectors, the usual HPCMatrix shape): one wavefront per row
// would leave most lanes idle and pay a wave launch per 8..512 bytes.  Here L = pow2 >= ncols/2 lanes
// share a row (each owns two adjacent columns, whose x values stay in registers for the whole kernel),
// a wavefront covers 64/L rows per pass and loops over ROWS_PER_WAVE_PASSES passes with 4 loads in
// flight; a contiguous block (lda == ncols == 2L) is then read as one sequential stream.
constexpr int SKINNY_PASSES = 16;

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double seg_x(const double *__restrict__ x_lo, int64_t n_lo,
                                        const double *__restrict__ x_own, int64_t n_own,
                                        const double *__restrict__ x_hi, int64_t n_hi, int64_t e)
{
    if (e < n_lo) return x_lo[e];
    if (e < n_lo + n_own) return x_own[e - n_lo];
    if (e < n_lo + n_own + n_hi) return x_hi[e - n_lo - n_own];
    return 0.0;
}

__global__ __launch_bounds__(256) void gemv_skinny_kernel(const double *__restrict__ A, int64_t lda,
                                                          int64_t nrows, int64_t ncols, int L,
                                                          const double *__restrict__ x_lo, int64_t n_lo,
                                                          const double *__restrict__ x_own, int64_t n_own,
                                                          const double *__restrict__ x_hi, int64_t n_hi,
                                                          double *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int sub = lane % L;                 // column pair within the row
    const int rsub = lane / L;                // row within the pass
    const int rpp = 64 / L;                   // rows per pass
    const int64_t c0 = 2 * (int64_t)sub;
    const bool has0 = c0 < ncols, has1 = c0 + 1 < ncols;
    const double xa = has0 ? seg_x(x_lo, n_lo, x_own, n_own, x_hi, n_hi, c0) : 0.0;
    const double xb = has1 ? seg_x(x_lo, n_lo, x_own, n_own, x_hi, n_hi, c0 + 1) : 0.0;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t row0 = wave * (int64_t)(rpp * SKINNY_PASSES);
    const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && has1;
#pragma unroll 4
    for (int p = 0; p < SKINNY_PASSES; ++p) {
        const int64_t row = row0 + (int64_t)p * rpp + rsub;
        double acc = 0.0;
        if (row < nrows) {
            const double *a = A + row * lda + c0;
            if (vec) {
                const double2 av = *reinterpret_cast<const double2 *>(a);
                acc = av.x * xa;
                acc += av.y * xb;
            } else {
                if (has0) acc = a[0] * xa;
                if (has1) acc += a[1] * xb;
            }
        }
        // the L lanes of a row: within 16 lanes by DPP (quad permutes, then the half-row and row mirrors: every lane ends
        // with the sum of its 4, 8, 16) -- VALU moves instead of the two ds_bpermute per step that made this kernel
        // LDS-pipe-bound (ten of them per KiB loaded at L = 32); only the steps across 16-lane rows go through the crossbar
        if (L > 1) acc += dpp_f64<0xB1>(acc);          // quad_perm [1,0,3,2]
        if (L > 2) acc += dpp_f64<0x4E>(acc);          // quad_perm [2,3,0,1]
        if (L > 4) acc += dpp_f64<0x141>(acc);         // row_half_mirror
        if (L > 8) acc += dpp_f64<0x140>(acc);         // row_mirror
        if (L > 16) acc += __shfl_xor(acc, 16, 64);
        if (L > 32) acc += __shfl_xor(acc, 32, 64);
        if (sub == 0 && row < nrows) y[row] = acc;
    }
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
        for (; r + 7 * rstep < r1; r += 8 * rstep) {      // 8 independent loads in flight
            double av[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = A[(r + (int64_t)u * rstep) * lda + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = x[r + (int64_t)u * rstep];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += av[u] * xv[u];
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

// The same with 16-byte loads (ncols and lda even, A 16-byte aligned): a lane owns TWO adjacent columns, a wavefront
// covers up to 128 columns x (128/W) rows per pass -- 8-byte loads run at 0.54-0.70 of the 16-byte rate
// (MI355X_MICROARCH.md), which is what the scalar form measured (0.53-0.66 of peak against 0.66-0.74 for A*x).
__global__ __launch_bounds__(GEMVT_THREADS) void gemv_t_stage1_vec2(const double *__restrict__ A, int64_t lda,
                                                                    int64_t nrows, int64_t ncols,
                                                                    const double *__restrict__ x, int W,
                                                                    int64_t rows_per_chunk,
                                                                    double *__restrict__ partial)
{
    typedef double vd2 __attribute__((ext_vector_type(2)));
    __shared__ vd2 red[GEMVT_THREADS];
    const int tid = threadIdx.x;
    const int WL = W / 2;                // lanes per row
    const int c = tid % WL;              // column PAIR within the tile
    const int rl = tid / WL;
    const int rstep = GEMVT_THREADS / WL;
    const int64_t col = (int64_t)blockIdx.x * W + 2 * c;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = min(nrows, r0 + rows_per_chunk);
    vd2 acc = (vd2)(0.0);
    if (col < ncols) {
        int64_t r = r0 + rl;
        for (; r + 7 * rstep < r1; r += 8 * rstep) {      // 8 independent 16-byte loads in flight
            vd2 av[8];
            double xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = *reinterpret_cast<const vd2 *>(A + (r + (int64_t)u * rstep) * lda + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = x[r + (int64_t)u * rstep];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc.x += av[u].x * xv[u];
                acc.y += av[u].y * xv[u];
            }
        }
        for (; r < r1; r += rstep) {
            const vd2 a = *reinterpret_cast<const vd2 *>(A + r * lda + col);
            const double xr = x[r];
            acc.x += a.x * xr;
            acc.y += a.y * xr;
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < WL && col < ncols) {
        vd2 s = red[tid];
        for (int k = 1; k < rstep; ++k) {                 // ascending row phase
            s.x += red[k * WL + tid].x;
            s.y += red[k * WL + tid].y;
        }
        *reinterpret_cast<vd2 *>(partial + (int64_t)blockIdx.y * ncols + col) = s;
    }
}

// stage 2: column sums of partial[nchunks][ncols].  A workgroup owns CT = min(64, W) columns; its
// 1024/CT row phases each sum every (1024/CT)-th chunk in ascending order (8 loads in flight), then
// the phases are added in ascending order through LDS -- a fixed tree, so the result is deterministic.
constexpr int GEMVT2_THREADS = 1024;

__global__ __launch_bounds__(GEMVT2_THREADS) void gemv_t_stage2(const double *__restrict__ partial,
                                                                int64_t nchunks, int64_t ncols, int CT,
                                                                double *__restrict__ y)
{
    __shared__ double red[GEMVT2_THREADS];
    const int tid = threadIdx.x;
    const int c = tid % CT, ph = tid / CT, nph = GEMVT2_THREADS / CT;
    const int64_t col = (int64_t)blockIdx.x * CT + c;
    double s = 0.0;
    if (col < ncols) {
        int64_t k = ph;
        for (; k + 7 * nph < nchunks; k += 8 * (int64_t)nph) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(k + (int64_t)u * nph) * ncols + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nchunks; k += nph) s += partial[k * ncols + col];
    }
    red[tid] = s;
    __syncthreads();
    if (ph == 0 && col < ncols) {
        double t = red[c];
        for (int q = 1; q < nph; ++q) t += red[q * CT + c];
        y[col] = t;
    }
}

static int64_t gemv_t_rows_per_chunk(int64_t nrows, int64_t ncols)
{
    // enough workgroups to fill 256 CUs several times over, but chunks of at least 64 rows
    int W = 1;
    while (W < 64 && W < ncols) W <<= 1;
    const int64_t tiles = (ncols + W - 1) / W;
    // ~8 workgroups per CU in stage 1; half that for very narrow blocks, whose chunks would shrink to a few dozen KB
    // (1 000 000 x 16: 0.030 ms with 1024 chunks, 0.034 with 2048)
    const int64_t target = ncols <= 32 ? 1024 : 2048;
    int64_t want_chunks = (target + tiles - 1) / (tiles > 0 ? tiles : 1);
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
    const bool vec2 = ncols % 2 == 0 && lda % 2 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(work) & 15) == 0;
    if (vec2) {
        int W2 = 2;
        while (W2 < 128 && W2 < ncols) W2 <<= 1;
        const int64_t tiles2 = (ncols + W2 - 1) / W2;
        gemv_t_stage1_vec2<<<dim3((uint32_t)tiles2, (uint32_t)nchunks), GEMVT_THREADS, 0, s>>>(
            A, lda, nrows, ncols, x, W2, rpc, static_cast<double *>(work));
    } else {
        gemv_t_stage1<<<dim3((uint32_t)tiles, (uint32_t)nchunks), GEMVT_THREADS, 0, s>>>(
            A, lda, nrows, ncols, x, W, rpc, static_cast<double *>(work));
    }
    HPCLA_CHECK_LAUNCH();
    const int CT = W < 64 ? W : 64;
    gemv_t_stage2<<<(uint32_t)((ncols + CT - 1) / CT), GEMVT2_THREADS, 0, s>>>(
        static_cast<const double *>(work), nchunks, ncols, CT, y_full);
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
    const int64_t ncols = n_lo + n_own + n_hi;
    if (ncols > 0 && ncols <= 128) {
        int L = 1;
        while (2 * L < ncols) L <<= 1;
        const int64_t rows_per_wave = (int64_t)(64 / L) * SKINNY_PASSES;
        const int64_t waves = (nrows + rows_per_wave - 1) / rows_per_wave;
        HPCLA_CHECK_GRID((waves + 3) / 4, "gemv");
        gemv_skinny_kernel<<<(uint32_t)((waves + 3) / 4), 256, 0, as_stream(stream)>>>(
            A, lda, nrows, ncols, L, x_lo, n_lo, x_own, n_own, x_hi, n_hi, y);
        HPCLA_CHECK_LAUNCH();
        return HPCLA_OK;
    }
    HPCLA_CHECK_GRID((nrows + 3) / 4, "gemv");
    gemv_rowmajor_kernel<<<(uint32_t)((nrows + 3) / 4), 256, 0, as_stream(stream)>>>(
        A, lda, nrows, x_lo, n_lo, x_own, n_own, x_hi, n_hi, y);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}
