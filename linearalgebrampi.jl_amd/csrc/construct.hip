// construct.hip -- device-side construction of the compressed local column space (SURVEY.md 8f
// "next" rank 2) and on-device generators of the BASELINE stencil workloads.
//
// Reference (host, per nonzero): `col_indices = unique!(sort(copy(rowval)))` then a binary search per
// nonzero (`compress_AT`, src/sparse.jl:501-509, 137-144) -- O(nnz log nnz), seconds at 10^8 nnz.
// Here: a presence bitmap over the column window, one exclusive scan, one emit pass -- O(nnz + window),
// milliseconds, same result (sorted unique global columns, local = rank in that list).
#include "common.h"

namespace hpcla {

constexpr int SCAN_T = 256;
constexpr int SCAN_E = 4;                       // elements per thread
constexpr int SCAN_B = SCAN_T * SCAN_E;         // 1024 elements per block

__global__ __launch_bounds__(256) void mark_present_kernel(const int64_t *__restrict__ col,
                                                           int64_t nnz, int64_t lo, int64_t window,
                                                           unsigned char *__restrict__ present,
                                                           int *__restrict__ err)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < nnz; i += stride) {
        const int64_t c = col[i] - lo;
        if (c < 0 || c >= window) { atomicOr(err, 1); continue; }
        present[c] = 1;
    }
}

// block-local inclusive scan of SCAN_B flags; writes the block total
__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t *s_warp, int64_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_warp[w] = inc;
    __syncthreads();
    int64_t base = 0;
    for (int k = 0; k < w; ++k) base += s_warp[k];
    if (total) {
        int64_t tot = 0;
        for (int k = 0; k < SCAN_T / 64; ++k) tot += s_warp[k];
        *total = tot;
    }
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_T) void scan_phase1_kernel(const unsigned char *__restrict__ flags,
                                                             int64_t n, int64_t *__restrict__ block_sums)
{
    __shared__ int64_t s_warp[SCAN_T / 64];
    const int64_t b0 = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_E;
    int64_t v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_E; ++k)
        if (b0 + k < n) v += flags[b0 + k] ? 1 : 0;
    int64_t tot;
    (void)block_exclusive_scan(v, s_warp, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// one block scans all block sums in place (exclusive), returns the grand total in total[0]
__global__ __launch_bounds__(SCAN_T) void scan_phase2_kernel(int64_t *__restrict__ block_sums, int64_t nb,
                                                             int64_t *__restrict__ total)
{
    __shared__ int64_t s_warp[SCAN_T / 64];
    __shared__ int64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t c = 0; c < nb; c += SCAN_T) {
        const int64_t i = c + threadIdx.x;
        const int64_t v = i < nb ? block_sums[i] : 0;
        int64_t tot;
        const int64_t ex = block_exclusive_scan(v, s_warp, &tot);
        const int64_t carry = s_carry;
        if (i < nb) block_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = s_carry;
}

// rank[i] = number of present flags before i (exclusive), for every i in the window
__global__ __launch_bounds__(SCAN_T) void scan_phase3_kernel(const unsigned char *__restrict__ flags,
                                                             int64_t n, const int64_t *__restrict__ block_offs,
                                                             int64_t lo, int64_t *__restrict__ rank,
                                                             int64_t *__restrict__ col_indices)
{
    __shared__ int64_t s_warp[SCAN_T / 64];
    const int64_t b0 = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_E;
    int f[SCAN_E];
    int64_t v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_E; ++k) {
        f[k] = (b0 + k < n && flags[b0 + k]) ? 1 : 0;
        v += f[k];
    }
    int64_t pos = block_offs[blockIdx.x] + block_exclusive_scan(v, s_warp, nullptr);
#pragma unroll
    for (int k = 0; k < SCAN_E; ++k) {
        if (b0 + k < n) {
            rank[b0 + k] = pos;
            if (f[k]) { if (col_indices) col_indices[pos] = lo + b0 + k; ++pos; }
        }
    }
}

template <typename I>
__global__ __launch_bounds__(256) void emit_colval_kernel(const int64_t *__restrict__ col, int64_t nnz,
                                                          int64_t lo, const int64_t *__restrict__ rank,
                                                          I *__restrict__ colval, int base)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < nnz; i += stride) colval[i] = (I)(rank[col[i] - lo] + base);
}

// ---- stencil generators (create_2d_laplacian, test/test_factorization.jl:60-102, and its 7-point
// analogue): rows [row_start,row_end) with GLOBAL 0-based columns, ascending within a row ----------
// number of stored entries in rows 0..idx-1: 5*idx minus the missing west/east/south/north neighbours
__host__ __device__ __forceinline__ int64_t poisson2d_prefix(int64_t idx, int64_t nx, int64_t ny)
{
    const int64_t j = idx / nx, i = idx % nx;
    const int64_t west = j + (i > 0 ? 1 : 0);                 // rows with i == 0 before idx
    const int64_t east = j;                                   // rows with i == nx-1 before idx
    const int64_t south = idx < nx ? idx : nx;                // rows of the first grid line before idx
    const int64_t north = idx > (ny - 1) * nx ? idx - (ny - 1) * nx : 0;   // rows of the last line before idx
    return 5 * idx - west - east - south - north;
}

// 7-point analogue, idx = (k*ny + j)*nx + i: 7*idx minus the missing neighbours of the rows before idx
__host__ __device__ __forceinline__ int64_t poisson3d_prefix(int64_t idx, int64_t nx, int64_t ny, int64_t nz)
{
    const int64_t nxy = nx * ny;
    const int64_t kk = idx / nxy, rem = idx % nxy;
    const int64_t lines = idx / nx, i = idx % nx;
    const int64_t west = lines + (i > 0 ? 1 : 0);
    const int64_t east = lines;
    const int64_t south = kk * nx + (rem < nx ? rem : nx);                                   // rows with j == 0
    const int64_t north = kk * nx + (rem > (ny - 1) * nx ? rem - (ny - 1) * nx : 0);           // rows with j == ny-1
    const int64_t down = idx < nxy ? idx : nxy;                                               // rows with k == 0
    const int64_t up = idx > (nz - 1) * nxy ? idx - (nz - 1) * nxy : 0;                        // rows with k == nz-1
    return 7 * idx - west - east - south - north - down - up;
}

__global__ __launch_bounds__(256) void gen_poisson3d_kernel(int64_t nx, int64_t ny, int64_t nz, int64_t row_start,
                                                            int64_t nloc, int64_t *__restrict__ rowptr,
                                                            int64_t *__restrict__ colidx,
                                                            double *__restrict__ vals)
{
    const int64_t nxy = nx * ny;
    const int64_t base = poisson3d_prefix(row_start, nx, ny, nz);
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; r <= nloc; r += stride) {
        const int64_t idx = row_start + r;
        int64_t p = poisson3d_prefix(idx, nx, ny, nz) - base;
        rowptr[r] = p;
        if (r == nloc) break;
        const int64_t i = idx % nx, j = (idx / nx) % ny, k = idx / nxy;
        if (k > 0)      { colidx[p] = idx - nxy; vals[p++] = -1.0; }
        if (j > 0)      { colidx[p] = idx - nx;  vals[p++] = -1.0; }
        if (i > 0)      { colidx[p] = idx - 1;   vals[p++] = -1.0; }
        colidx[p] = idx; vals[p++] = 6.0;
        if (i < nx - 1) { colidx[p] = idx + 1;   vals[p++] = -1.0; }
        if (j < ny - 1) { colidx[p] = idx + nx;  vals[p++] = -1.0; }
        if (k < nz - 1) { colidx[p] = idx + nxy; vals[p++] = -1.0; }
    }
}

__global__ __launch_bounds__(256) void gen_poisson2d_kernel(int64_t nx, int64_t ny, int64_t row_start,
                                                            int64_t nloc, int64_t *__restrict__ rowptr,
                                                            int64_t *__restrict__ colidx,
                                                            double *__restrict__ vals)
{
    const int64_t base = poisson2d_prefix(row_start, nx, ny);
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; r <= nloc; r += stride) {
        const int64_t idx = row_start + r;
        int64_t p = poisson2d_prefix(idx, nx, ny) - base;
        rowptr[r] = p;
        if (r == nloc) break;
        const int64_t i = idx % nx, j = idx / nx;
        if (j > 0)      { colidx[p] = idx - nx; vals[p++] = -1.0; }
        if (i > 0)      { colidx[p] = idx - 1;  vals[p++] = -1.0; }
        colidx[p] = idx; vals[p++] = 4.0;
        if (i < nx - 1) { colidx[p] = idx + 1;  vals[p++] = -1.0; }
        if (j < ny - 1) { colidx[p] = idx + nx; vals[p++] = -1.0; }
    }
}

static inline uint32_t grid_for(int64_t n)
{
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (uint32_t)g;
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int64_t hpcla_colspace_work_bytes(int64_t window)
{
    const int64_t nb = (window + SCAN_B - 1) / SCAN_B;
    // present flags (window bytes, padded) + rank (window int64) + block sums (nb int64) + total + err
    return ((window + 15) / 16) * 16 + 8 * window + 8 * nb + 64;
}

// Compress the global column ids of this rank's nonzeros (device int64) to the local column space.
// `window` columns starting at `col_lo` must cover every id (use the local row range widened by the
// stencil reach, or [0, ncols_global)).  On return *ncols_compressed_host = number of distinct
// columns; col_indices_out (capacity `window`) holds them ascending; colval_out holds the local
// indices (+ index_base).  Synchronises the stream once (to return the count).
template <typename I>
static int colspace_impl(const int64_t *colidx_global, int64_t nnz, int64_t col_lo, int64_t window,
                         I *colval_out, int index_base, int64_t *col_indices_out,
                         int64_t *ncols_compressed_host, void *work, void *stream)
{
    if (nnz < 0 || window <= 0) return set_error(HPCLA_ERR_INVALID, "colspace: bad sizes");
    if (!work || !ncols_compressed_host || !col_indices_out)
        return set_error(HPCLA_ERR_INVALID, "colspace: null work/output");
    if (nnz > 0 && (!colidx_global || !colval_out)) return set_error(HPCLA_ERR_INVALID, "colspace: null arrays");
    hipStream_t s = as_stream(stream);
    const int64_t nb = (window + SCAN_B - 1) / SCAN_B;
    unsigned char *present = reinterpret_cast<unsigned char *>(work);
    int64_t *rank = reinterpret_cast<int64_t *>(present + ((window + 15) / 16) * 16);
    int64_t *block_sums = rank + window;
    int64_t *total = block_sums + nb;
    int *err = reinterpret_cast<int *>(total + 1);
    HPCLA_CHECK_HIP(hipMemsetAsync(present, 0, window, s));
    HPCLA_CHECK_HIP(hipMemsetAsync(total, 0, 16, s));
    if (nnz > 0) {
        mark_present_kernel<<<grid_for(nnz), 256, 0, s>>>(colidx_global, nnz, col_lo, window, present, err);
        HPCLA_CHECK_LAUNCH();
    }
    scan_phase1_kernel<<<(uint32_t)nb, SCAN_T, 0, s>>>(present, window, block_sums);
    HPCLA_CHECK_LAUNCH();
    scan_phase2_kernel<<<1, SCAN_T, 0, s>>>(block_sums, nb, total);
    HPCLA_CHECK_LAUNCH();
    scan_phase3_kernel<<<(uint32_t)nb, SCAN_T, 0, s>>>(present, window, block_sums, col_lo, rank, col_indices_out);
    HPCLA_CHECK_LAUNCH();
    if (nnz > 0) {
        emit_colval_kernel<I><<<grid_for(nnz), 256, 0, s>>>(colidx_global, nnz, col_lo, rank, colval_out, index_base);
        HPCLA_CHECK_LAUNCH();
    }
    int64_t h[2] = {0, 0};
    HPCLA_CHECK_HIP(hipMemcpyAsync(h, total, 16, hipMemcpyDeviceToHost, s));
    HPCLA_CHECK_HIP(hipStreamSynchronize(s));
    if ((int)(h[1] & 0xffffffff)) return set_error(HPCLA_ERR_INVALID, "colspace: a column id lies outside the window");
    *ncols_compressed_host = h[0];
    return HPCLA_OK;
}

HPCLA_API int hpcla_compress_columns_i32(const int64_t *colidx_global, int64_t nnz, int64_t col_lo,
                                         int64_t window, int32_t *colval_out, int index_base,
                                         int64_t *col_indices_out, int64_t *ncols_compressed_host,
                                         void *work, void *stream)
{
    return colspace_impl<int32_t>(colidx_global, nnz, col_lo, window, colval_out, index_base,
                                  col_indices_out, ncols_compressed_host, work, stream);
}

HPCLA_API int hpcla_compress_columns_i64(const int64_t *colidx_global, int64_t nnz, int64_t col_lo,
                                         int64_t window, int64_t *colval_out, int index_base,
                                         int64_t *col_indices_out, int64_t *ncols_compressed_host,
                                         void *work, void *stream)
{
    return colspace_impl<int64_t>(colidx_global, nnz, col_lo, window, colval_out, index_base,
                                  col_indices_out, ncols_compressed_host, work, stream);
}

// nnz of rows [row_start,row_end) of the nx*ny 5-point Laplacian (closed form, host)
// ---- structure digest: the device form of compute_structural_hash's local pass ---------------------
// The reference hashes rowptr / colval / col_indices on the host with Blake3 (src/sparse.jl:97-121) purely
// as a memoization key: digests are compared for equality, never against constants.  For a device-resident
// structure that would mean a D2H copy of colval (250 MB at config 5) plus a host hash per matrix; here a
// kernel folds the array into four 64-bit words, word k = sum_i mix(a[i]*P1 + (i+1)*P2 + S_k) mod 2^64.
// The position enters every term, so the digest is order-sensitive; integer addition is exact and
// commutative, so atomics give the same words on every run.  The host twin (partition.py) applies the
// same formula with numpy -- host-built and device-built structures hash alike.
namespace hpcla {
__device__ __forceinline__ uint64_t digest_mix(uint64_t z)
{
    z ^= z >> 29;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 32;
    return z;
}
template <typename I>
__global__ __launch_bounds__(256) void digest_kernel(const I *__restrict__ a, int64_t n, uint64_t *__restrict__ out)
{
    const uint64_t P1 = 0x9E3779B97F4A7C15ull, P2 = 0xD1B54A32D192ED03ull;
    const uint64_t S[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
    uint64_t h[4] = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t t = (uint64_t)(int64_t)a[i] * P1 + (uint64_t)(i + 1) * P2;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] += digest_mix(t + S[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t v = h[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd((unsigned long long *)(out + k), (unsigned long long)v);
    }
}
template <typename I>
static int digest_impl(const I *a, int64_t n, uint64_t *out_host, void *stream)
{
    if (n < 0 || !out_host) return set_error(HPCLA_ERR_INVALID, "digest: bad arguments");
    for (int k = 0; k < 4; ++k) out_host[k] = 0;
    if (n == 0) return HPCLA_OK;
    if (!a) return set_error(HPCLA_ERR_INVALID, "digest: null array");
    uint64_t *d = nullptr;
    HPCLA_CHECK_HIP(hipMalloc((void **)&d, 4 * sizeof(uint64_t)));
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(d, 0, 4 * sizeof(uint64_t), s);
    if (e == hipSuccess) {
        int64_t g = (n + 255) / 256;
        if (g > 4096) g = 4096;
        digest_kernel<I><<<(uint32_t)g, 256, 0, s>>>(a, n, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(HPCLA_ERR_HIP, "digest: %s", hipGetErrorString(e));
    return HPCLA_OK;
}
}  // namespace hpcla

HPCLA_API int hpcla_digest_i32(const int32_t *a, int64_t n, uint64_t *out_host, void *stream)
{
    return digest_impl<int32_t>(a, n, out_host, stream);
}
HPCLA_API int hpcla_digest_i64(const int64_t *a, int64_t n, uint64_t *out_host, void *stream)
{
    return digest_impl<int64_t>(a, n, out_host, stream);
}

HPCLA_API int64_t hpcla_poisson2d_nnz(int64_t nx, int64_t ny, int64_t row_start, int64_t row_end)
{
    return poisson2d_prefix(row_end, nx, ny) - poisson2d_prefix(row_start, nx, ny);
}

HPCLA_API int64_t hpcla_poisson3d_nnz(int64_t nx, int64_t ny, int64_t nz, int64_t row_start, int64_t row_end)
{
    return poisson3d_prefix(row_end, nx, ny, nz) - poisson3d_prefix(row_start, nx, ny, nz);
}

HPCLA_API int hpcla_gen_poisson3d(int64_t nx, int64_t ny, int64_t nz, int64_t row_start, int64_t row_end,
                                  int64_t *rowptr_out, int64_t *colidx_out, double *vals_out, void *stream)
{
    if (nx < 1 || ny < 1 || nz < 1 || row_start < 0 || row_end < row_start || row_end > nx * ny * nz)
        return set_error(HPCLA_ERR_INVALID, "gen_poisson3d: bad range");
    if (!rowptr_out || !colidx_out || !vals_out) return set_error(HPCLA_ERR_INVALID, "gen_poisson3d: null output");
    const int64_t nloc = row_end - row_start;
    gen_poisson3d_kernel<<<grid_for(nloc + 1), 256, 0, as_stream(stream)>>>(nx, ny, nz, row_start, nloc, rowptr_out,
                                                                          colidx_out, vals_out);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_gen_poisson2d(int64_t nx, int64_t ny, int64_t row_start, int64_t row_end,
                                  int64_t *rowptr_out, int64_t *colidx_out, double *vals_out, void *stream)
{
    if (nx < 1 || ny < 1 || row_start < 0 || row_end < row_start || row_end > nx * ny)
        return set_error(HPCLA_ERR_INVALID, "gen_poisson2d: bad range");
    if (!rowptr_out || !colidx_out || !vals_out) return set_error(HPCLA_ERR_INVALID, "gen_poisson2d: null output");
    const int64_t nloc = row_end - row_start;
    gen_poisson2d_kernel<<<grid_for(nloc + 1), 256, 0, as_stream(stream)>>>(nx, ny, row_start, nloc, rowptr_out,
                                                                          colidx_out, vals_out);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}
