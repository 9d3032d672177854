// tiled.hip -- OPT-IN "tile stream" copy of a CSR matrix for the SpMV of UNSTRUCTURED matrices (round 5).
//
// What the counters say about the CSR kernel on uniformly random columns (config 5's matrix times one vector, the shape of the
// reference's own single-rank SpMV benchmark, tools/benchmark_single_rank.jl:48-71): 8.6 GB of HBM / fabric traffic per launch
// where the algorithm needs 0.9 GB -- every stored entry gathers ONE 8-byte x value and the memory system fetches a whole
// 128-byte line for it (profiles/r05_pmc_sprandv8_*.csv): 9.5 x the algorithmic bytes at 6.9 TB/s.  The kernel is at the
// ceiling of what it moves; the fix is to move less.
//
// CACHE BLOCKING BY COLUMN TILE, without giving up the reference's bits.  The column space is cut into tiles of `tile_cols`
// consecutive columns (1 MiB of x by default: a few of them fit an XCD's 4 MiB L2).  A wave owns R consecutive rows and walks
// THEIR entries tile by tile -- all its entries in tile 0, then tile 1, ... -- while every other wave of the launch does the
// same at the same pace (the launch is sized so that every wave is resident at once), so at any moment the whole GPU gathers
// from a few tiles of x that live in the L2s: x is read from HBM once per XCD instead of one line per stored entry.
// A row's entries are stored by ascending column, a tile index grows with the column, so "tile by tile" visits a row's entries
// IN STORED ORDER: the wave keeps one running sum per row (in LDS) and adds the separately rounded products to it one by one --
// exactly acc += nzval[j] * x[colval[j]] of src/sparse.jl:2059-2064, bit for bit.
//
// The copy (plan time, device-built): the entries of each R-row group re-ordered by (tile, row, column) as three coalesced
// streams -- column (u32, split column space), value (f64), row within the group (u16) -- 14 B per entry, each group padded to
// a multiple of 64.  It snapshots the VALUES, so it belongs to the matrix object, not to the structure-keyed plan.
#include <string.h>

#include <new>
#include <vector>

#include "common.h"

struct hpcla_tiled {
    int64_t nrows = 0, nnz = 0, ncols = 0;
    int rows_per_group = 0, tile_shift = 0, n_tiles = 0;
    int64_t n_groups = 0, total = 0, bytes = 0;
    int64_t *group_off = nullptr;        // device, n_groups + 1 (multiples of 64)
    uint32_t *t_col = nullptr;           // device, total
    double *t_val = nullptr;             // device, total
    uint16_t *t_row = nullptr;           // device, total (0xFFFF = padding)
};

namespace hpcla {

constexpr int TL_WAVES = 4;              // waves (= row groups) per workgroup
constexpr uint16_t TL_PAD = 0xFFFF;
constexpr int TL_U = 4;                  // steps of 64 entries in flight per wave
constexpr int TL_MAX_TILES = 2048;       // cursors per wave in the builder's LDS (8 KiB per wave)

template <bool SPLIT>
__device__ __forceinline__ double tl_gather(const double *__restrict__ x_own, const double *__restrict__ x_ghost,
                                            int64_t n_own, int64_t col)
{
    if (SPLIT) return col < n_own ? x_own[col] : x_ghost[col - n_own];
    return x_own[col];
}

// ---- builder: one WAVE per row group, no workgroup barrier (LDS operations of one wave complete in order) ---------------
template <typename I>
__global__ __launch_bounds__(64 * TL_WAVES) void tiled_build_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval, int64_t nrows, int base,
    int R, int tile_shift, int T, int64_t n_groups, const int64_t *__restrict__ group_off,
    uint32_t *__restrict__ t_col, double *__restrict__ t_val, uint16_t *__restrict__ t_row)
{
    extern __shared__ uint32_t tl_cursor_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * TL_WAVES + wave;
    if (g >= n_groups) return;
    uint32_t *cursor = tl_cursor_all + (size_t)wave * T;
    const int64_t r0 = g * R, r1 = (r0 + R < nrows) ? r0 + R : nrows;
    const int64_t p0 = (int64_t)rowptr[r0] - base, p1 = (int64_t)rowptr[r1] - base;
    // 1. entries per tile
    for (int t = lane; t < T; t += 64) cursor[t] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int64_t i = p0 + lane; i < p1; i += 64) {
        const uint32_t t = (uint32_t)(((int64_t)colval[i] - base) >> tile_shift);
        atomicAdd(&cursor[t], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 2. exclusive scan over the T tiles: lane l owns K consecutive bins
    {
        const int K = (T + 63) / 64, b0 = lane * K;
        uint32_t sum = 0;
        for (int k = 0; k < K; ++k)
            if (b0 + k < T) sum += cursor[b0 + k];
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        uint32_t run = incl - sum;
        for (int k = 0; k < K; ++k)
            if (b0 + k < T) {
                const uint32_t c = cursor[b0 + k];
                cursor[b0 + k] = run;
                run += c;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 3. rows in order; within a row the tiles ascend, so a tile's entries of a row are one run of consecutive lanes
    const int64_t out0 = group_off[g];
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t a = (int64_t)rowptr[r] - base, b = (int64_t)rowptr[r + 1] - base;
        for (int64_t j0 = a; j0 < b; j0 += 64) {
            const int64_t j = j0 + lane;
            const bool valid = j < b;
            const int64_t c = valid ? (int64_t)colval[j] - base : 0;
            const uint32_t t = valid ? (uint32_t)(c >> tile_shift) : 0xFFFFFFFFu;
            const uint32_t prev_t = __shfl_up(t, 1, 64), next_t = __shfl_down(t, 1, 64);
            const bool is_start = valid && (lane == 0 || prev_t != t);
            const bool is_end = valid && (lane == 63 || next_t != t);
            const uint64_t starts = __ballot(is_start);
            const uint64_t below = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
            const int pos = below ? lane - (63 - __clzll((long long)below)) : 0;   // distance to this run's first lane
            uint32_t cur = 0;
            if (valid) cur = cursor[t];
            if (valid) {
                const int64_t dest = out0 + cur + pos;
                t_col[dest] = (uint32_t)c;
                t_val[dest] = nzval[j];
                t_row[dest] = (uint16_t)(r - r0);
            }
            if (is_end) cursor[t] = cur + (uint32_t)pos + 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- y = A*x from the tile stream ------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(64 * TL_WAVES) void spmv_tiled_kernel(
    const int64_t *__restrict__ group_off, const uint32_t *__restrict__ t_col, const double *__restrict__ t_val,
    const uint16_t *__restrict__ t_row, const double *__restrict__ x_own, const double *__restrict__ x_ghost, int64_t n_own,
    double *__restrict__ y, int64_t nrows, int R, int64_t n_groups, int tile_shift)
{
    extern __shared__ double tl_acc_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * TL_WAVES + wave;
    if (g >= n_groups) return;
    double *acc = tl_acc_all + (size_t)wave * R;
    for (int i = lane; i < R; i += 64) acc[i] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t s = group_off[g], e = group_off[g + 1];
    // TL_U steps of 64 entries per iteration: all of their stream loads leave first, then all of their x gathers, then the
    // steps are added one after the other -- a wave that takes one step at a time waits a full memory round trip for the
    // stream and another for the gather at every step (0.99 ms on config 5's matrix; profiles/r05_tile_stream.log)
    for (int64_t p = s; p < e; p += 64 * TL_U) {
        uint32_t cc[TL_U];
        double vv[TL_U], prod[TL_U];
        uint16_t rr[TL_U];
#pragma unroll
        for (int u = 0; u < TL_U; ++u) {
            const int64_t q = p + 64 * u + lane;
            const bool in = p + 64 * u < e;                               // wave-uniform (the stream is padded to 64)
            // the stream is read once: non-temporal, so that it does not displace the x tiles from the L2s
            cc[u] = in ? __builtin_nontemporal_load(t_col + q) : 0u;
            vv[u] = in ? __builtin_nontemporal_load(t_val + q) : 0.0;
            rr[u] = in ? __builtin_nontemporal_load(t_row + q) : TL_PAD;
        }
#pragma unroll
        for (int u = 0; u < TL_U; ++u) {
            prod[u] = 0.0;
            if (rr[u] != TL_PAD) prod[u] = vv[u] * tl_gather<SPLIT>(x_own, x_ghost, n_own, (int64_t)cc[u]);
        }
#pragma unroll
        for (int u = 0; u < TL_U; ++u) {
            if (p + 64 * u >= e) break;                                   // wave-uniform
            const uint16_t r = rr[u];
            const bool valid = r != TL_PAD;
            // A step of 64 entries may span several column tiles, and a row may have entries in more than one of them: the
            // step is taken TILE SEGMENT by tile segment (ascending), so that two runs of one row never update its sum at
            // once.  Inside a segment every row has at most one run of consecutive lanes; its head adds the run's products
            // in order.
            const uint32_t t = valid ? (cc[u] >> tile_shift) : 0xFFFFFFFFu;
            const uint32_t prev_r = __shfl_up((uint32_t)r, 1, 64), prev_t = __shfl_up(t, 1, 64);
            const bool head = valid && (lane == 0 || prev_r != (uint32_t)r || prev_t != t);
            const uint64_t stops = __ballot(head) | __ballot(!valid);
            const uint64_t above = lane == 63 ? 0ull : (stops >> (lane + 1));
            const int len = above ? 1 + __builtin_ctzll(above) : 64 - lane;  // entries of this lane's run (heads only)
            uint64_t segs = __ballot(valid && (lane == 0 || prev_t != t));   // first lane of every tile segment
            while (segs) {                                                   // wave-uniform
                const int lo = __builtin_ctzll(segs);
                segs &= segs - 1;
                const int hi = segs ? __builtin_ctzll(segs) : 64;
                const bool mine = head && lane >= lo && lane < hi;
                double a = 0.0;
                if (mine) a = acc[r] + prod[u];                              // the row's running sum continues: stored order
                for (int k = 1; __ballot(mine && k < len); ++k) {
                    const double q = __shfl_down(prod[u], k, 64);
                    if (mine && k < len) a += q;
                }
                if (mine) acc[r] = a;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    const int64_t r0 = g * R;
    for (int i = lane; i < R; i += 64)
        if (r0 + i < nrows) __builtin_nontemporal_store(acc[i], y + r0 + i);
}

static void tiled_free(hpcla_tiled *t)
{
    if (!t) return;
    if (t->group_off) (void)hipFree(t->group_off);
    if (t->t_col) (void)hipFree(t->t_col);
    if (t->t_val) (void)hipFree(t->t_val);
    if (t->t_row) (void)hipFree(t->t_row);
    delete t;
}

template <typename I>
static int tiled_create(hpcla_tiled **out, const I *rowptr, const I *colval, const double *nzval, int64_t nrows, int64_t nnz,
                        int64_t ncols, int index_base, int64_t tile_cols, void *stream)
{
    if (!out) return set_error(HPCLA_ERR_INVALID, "tiled_create: null output");
    *out = nullptr;
    if (nrows <= 0 || nnz <= 0 || ncols <= 0 || !rowptr || !colval || !nzval)
        return set_error(HPCLA_ERR_INVALID, "tiled_create: empty matrix or null pointer");
    if (index_base != 0 && index_base != 1) return set_error(HPCLA_ERR_INVALID, "tiled_create: index_base must be 0 or 1");
    if (ncols > 0xFFFFFFFFLL) return set_error(HPCLA_ERR_UNSUPPORTED, "tiled_create: more than 2^32 columns");
    if (tile_cols <= 0) tile_cols = 1 << 17;                 // 1 MiB of x
    int shift = 0;
    while (((int64_t)1 << shift) < tile_cols) ++shift;
    while (((ncols + ((int64_t)1 << shift) - 1) >> shift) > TL_MAX_TILES) ++shift;     // at most TL_MAX_TILES tiles
    const int T = (int)((ncols + ((int64_t)1 << shift) - 1) >> shift);
    // rows per wave: every wave of the launch resident at once (256 CUs x 32 waves), 64 <= R <= 1024
    int64_t R = (nrows + 8191) / 8192;
    R = (R + 63) / 64 * 64;
    if (R < 64) R = 64;
    if (R > 1024) R = 1024;
    const int64_t G = (nrows + R - 1) / R;
    hipStream_t s = as_stream(stream);
    // group boundaries -> padded stream offsets (plan time: one small D2H copy)
    std::vector<I> bounds((size_t)G + 1);
    {
        // rowptr[g * R], g < G: one strided 2-D copy (G rows of one element, pitch R elements); then rowptr[nrows]
        HPCLA_CHECK_HIP(hipMemcpy2DAsync(bounds.data(), sizeof(I), rowptr, (size_t)R * sizeof(I), sizeof(I), (size_t)G,
                                         hipMemcpyDeviceToHost, s));
        HPCLA_CHECK_HIP(hipMemcpyAsync(&bounds[(size_t)G], rowptr + nrows, sizeof(I), hipMemcpyDeviceToHost, s));
        HPCLA_CHECK_HIP(hipStreamSynchronize(s));
    }
    std::vector<int64_t> off((size_t)G + 1);
    off[0] = 0;
    for (int64_t g = 0; g < G; ++g) {
        const int64_t len = (int64_t)bounds[(size_t)g + 1] - (int64_t)bounds[(size_t)g];
        if (len < 0) return set_error(HPCLA_ERR_INVALID, "tiled_create: rowptr is not monotone");
        off[(size_t)g + 1] = off[(size_t)g] + (len + 63) / 64 * 64;
    }
    if ((int64_t)bounds[(size_t)G] - index_base != nnz) return set_error(HPCLA_ERR_INVALID, "tiled_create: rowptr[nrows] != nnz");
    hpcla_tiled *t = new (std::nothrow) hpcla_tiled();
    if (!t) return set_error(HPCLA_ERR_HIP, "tiled_create: out of host memory");
    t->nrows = nrows; t->nnz = nnz; t->ncols = ncols; t->rows_per_group = (int)R; t->tile_shift = shift; t->n_tiles = T;
    t->n_groups = G; t->total = off[(size_t)G];
    const int64_t tot = t->total > 0 ? t->total : 64;
    hipError_t e = hipMalloc((void **)&t->group_off, (size_t)(G + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&t->t_col, (size_t)tot * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&t->t_val, (size_t)tot * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&t->t_row, (size_t)tot * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpyAsync(t->group_off, off.data(), (size_t)(G + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(t->t_col, 0, (size_t)tot * sizeof(uint32_t), s);      // padding gathers x[0] (never used)
    if (e == hipSuccess) e = hipMemsetAsync(t->t_val, 0, (size_t)tot * sizeof(double), s);
    if (e == hipSuccess) e = hipMemsetAsync(t->t_row, 0xFF, (size_t)tot * sizeof(uint16_t), s);   // 0xFFFF = padding
    if (e != hipSuccess) {
        tiled_free(t);
        (void)hipGetLastError();
        return set_error(HPCLA_ERR_HIP, "tiled_create: %s", hipGetErrorString(e));
    }
    t->bytes = tot * 14 + (G + 1) * 8;
    const int64_t nb = (G + TL_WAVES - 1) / TL_WAVES;
    if (nb > 0x7fffffffLL) { tiled_free(t); return set_error(HPCLA_ERR_UNSUPPORTED, "tiled_create: too many row groups"); }
    tiled_build_kernel<I><<<(uint32_t)nb, 64 * TL_WAVES, (size_t)TL_WAVES * T * sizeof(uint32_t), s>>>(
        rowptr, colval, nzval, nrows, index_base, (int)R, shift, T, G, t->group_off, t->t_col, t->t_val, t->t_row);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);       // `off` / `bounds` go out of scope; plan time
    if (e != hipSuccess) {
        tiled_free(t);
        return set_error(HPCLA_ERR_HIP, "tiled_create: build kernel: %s", hipGetErrorString(e));
    }
    *out = t;
    return HPCLA_OK;
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int hpcla_tiled_create_i32(hpcla_tiled_t **out, const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                     int64_t nrows, int64_t nnz, int64_t ncols, int index_base, int64_t tile_cols, void *stream)
{
    return tiled_create<int32_t>(out, rowptr, colval_split, nzval, nrows, nnz, ncols, index_base, tile_cols, stream);
}

HPCLA_API int hpcla_tiled_create_i64(hpcla_tiled_t **out, const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                     int64_t nrows, int64_t nnz, int64_t ncols, int index_base, int64_t tile_cols, void *stream)
{
    return tiled_create<int64_t>(out, rowptr, colval_split, nzval, nrows, nnz, ncols, index_base, tile_cols, stream);
}

HPCLA_API int hpcla_tiled_destroy(hpcla_tiled_t *t)
{
    tiled_free(t);
    return HPCLA_OK;
}

HPCLA_API int hpcla_tiled_info(const hpcla_tiled_t *t, int64_t *bytes, int *n_tiles, int *rows_per_group)
{
    if (!t) return set_error(HPCLA_ERR_INVALID, "tiled_info: null handle");
    if (bytes) *bytes = t->bytes;
    if (n_tiles) *n_tiles = t->n_tiles;
    if (rows_per_group) *rows_per_group = t->rows_per_group;
    return HPCLA_OK;
}

HPCLA_API int hpcla_spmv_tiled_f64(const hpcla_tiled_t *t, const double *x_own, const double *x_ghost, int64_t n_own, double *y,
                                   void *stream)
{
    if (!t || !x_own || !y) return set_error(HPCLA_ERR_INVALID, "spmv_tiled: null pointer");
    if (!x_ghost && n_own < t->ncols) return set_error(HPCLA_ERR_INVALID, "spmv_tiled: x_own shorter than the column space and no ghost segment");
    const int64_t nb = (t->n_groups + TL_WAVES - 1) / TL_WAVES;
    const size_t lds = (size_t)TL_WAVES * t->rows_per_group * sizeof(double);
    hipStream_t s = as_stream(stream);
    if (x_ghost)
        spmv_tiled_kernel<true><<<(uint32_t)nb, 64 * TL_WAVES, lds, s>>>(t->group_off, t->t_col, t->t_val, t->t_row, x_own, x_ghost,
                                                                       n_own, y, t->nrows, t->rows_per_group, t->n_groups, t->tile_shift);
    else
        spmv_tiled_kernel<false><<<(uint32_t)nb, 64 * TL_WAVES, lds, s>>>(t->group_off, t->t_col, t->t_val, t->t_row, x_own, nullptr,
                                                                        0, y, t->nrows, t->rows_per_group, t->n_groups, t->tile_shift);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}
