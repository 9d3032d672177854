// spmm.hip -- CSR x dense-columns (SpMM) for MI355X, fp64.
//
// Replaces the reference's column loop `columns[k] = A * B[:, k]` (src/sparse.jl:2391-2413), which
// streams A from memory k times and does k halo exchanges.  Here A is streamed ONCE:
//
//   * a 256-thread workgroup owns RPB_MM = 64 consecutive rows; their (colval, nzval) range is
//     streamed coalesced into LDS, CHUNK_MM entries at a time;
//   * device-native layout (row-major B and C, k % 4 == 0): FOUR lanes share a row, each owns four
//     adjacent output columns and reads its 32-byte slice of a B row with two 16-byte loads
//     (spmm_rowblock_vec_kernel); any other layout / k: 16 lanes per row, one column each, generic
//     strides (spmm_rowblock_kernel);
//   * each C(r,c) is accumulated sequentially in stored order with separate multiply and add, so
//     every output column is bit-identical to a reference SpMV of that column.
//
// Generic strides are accepted for B and C (column-major = Julia Matrix), the row-major form is
// the fast one.  Algorithmic bytes: 12 B/nnz + 4 B/row + 8k B/row (C) + 8k B per B row touched.
#include "common.h"

namespace hpcla {

constexpr int TPB_MM = 256;
constexpr int GROUP = 16;                   // lanes per row
constexpr int NGROUPS = TPB_MM / GROUP;     // 16 rows in flight per pass
constexpr int SLOTS = 4;                    // rows per lane-group
constexpr int RPB_MM = NGROUPS * SLOTS;     // 64 rows per block
constexpr int CHUNK_MM = 2048;              // entries staged per pass: 2048 * 12 B = 24 KiB
constexpr int KT = 16;                      // columns per tile (one per lane of a group)

// The four rows of a lane-group advance TOGETHER, two entries each per step, so a lane has up to 8
// independent B loads in flight.  (A first version walked the rows one after the other with 4 loads
// in flight: a 5-entry stencil row then costs two latency rounds per row, eight per lane-group; the
// interleaved form measured 1.255 -> 1.041 ms on the 5-point matrix x 16 columns and 1.707 -> 1.616 ms
// on config 5's random pattern, profiles/r01_spmm_variants.log.)  Per row the entries are still
// accumulated one after the other in stored order: same bits.
template <typename I, bool SPLIT>
__global__ __launch_bounds__(TPB_MM) void spmm_rowblock_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ B_own, int64_t b_rs, int64_t b_cs,
    const double *__restrict__ B_ghost, int64_t bg_rs, int64_t n_own, double *__restrict__ C,
    int64_t c_rs, int64_t c_cs, int64_t nrows, int k, int base,
    const int32_t *__restrict__ block_list, uint32_t nblocks)
{
    __shared__ double s_val[CHUNK_MM];
    __shared__ int64_t s_col[CHUNK_MM];   // element offset of the B row (col * row stride), 64-bit

    const int tid = threadIdx.x;
    const int g = tid / GROUP, l = tid % GROUP;
    const uint32_t b = blockIdx.x;
    const int64_t blk = block_list ? (int64_t)block_list[b] : (int64_t)b;
    const int64_t r0 = blk * RPB_MM;
    const int nr = (int)((nrows - r0) < RPB_MM ? (nrows - r0) : RPB_MM);
    const int64_t p0 = (int64_t)rowptr[r0] - base;
    const int64_t p1 = (int64_t)rowptr[r0 + nr] - base;
    const int64_t total = p1 - p0;

    int64_t lo[SLOTS], hi[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int r = g + s * NGROUPS;
        lo[s] = hi[s] = 0;
        if (r < nr) {
            lo[s] = (int64_t)rowptr[r0 + r] - base - p0;
            hi[s] = (int64_t)rowptr[r0 + r + 1] - base - p0;
        }
    }

    for (int kt = 0; kt < k; kt += KT) {
        const int c = kt + l;
        const bool col_ok = c < k;
        const int64_t c_off = (int64_t)c * b_cs;
        double acc[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) acc[s] = 0.0;

        for (int64_t ch = 0; ch < total; ch += CHUNK_MM) {
            const int n = (int)((total - ch) < CHUNK_MM ? (total - ch) : CHUNK_MM);
            __syncthreads();   // previous pass finished reading LDS
            for (int i = tid; i < n; i += TPB_MM) {
                const int64_t col = (int64_t)__builtin_nontemporal_load(colval + p0 + ch + i) - base;
                s_val[i] = __builtin_nontemporal_load(nzval + p0 + ch + i);
                if (SPLIT)
                    s_col[i] = col < n_own ? col * b_rs : -(1 + (col - n_own) * bg_rs);
                else
                    s_col[i] = col * b_rs;
            }
            __syncthreads();
            if (!col_ok) continue;
            int a[SLOTS], len[SLOTS];
            int maxlen = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int64_t a64 = lo[s] > ch ? lo[s] : ch;
                const int64_t e64 = hi[s] < ch + n ? hi[s] : ch + n;
                a[s] = (int)(a64 - ch);
                len[s] = e64 > a64 ? (int)(e64 - a64) : 0;
                maxlen = len[s] > maxlen ? len[s] : maxlen;
            }
            for (int i = 0; i < maxlen; i += 2) {
                double v[SLOTS][2], bv[SLOTS][2];
                int64_t off[SLOTS][2];
                bool ok[SLOTS][2];
                // LDS reads first (all unconditional, index clamped to a valid entry), then the B
                // loads back to back, then the sums in stored order
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        ok[s][u] = i + u < len[s];
                        const int idx = ok[s][u] ? a[s] + i + u : 0;
                        off[s][u] = s_col[idx];
                        v[s][u] = s_val[idx];
                    }
                }
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        bv[s][u] = 0.0;
                        if (ok[s][u]) {
                            if (SPLIT && off[s][u] < 0)
                                bv[s][u] = B_ghost[(-off[s][u] - 1) + c];
                            else
                                bv[s][u] = B_own[off[s][u] + c_off];
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (ok[s][u]) acc[s] += v[s][u] * bv[s][u];
                }
            }
        }
        // (col_ok is uniform per lane for the whole kt pass: no barrier is skipped by a subset)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int r = g + s * NGROUPS;
            if (r < nr && col_ok) C[(r0 + r) * c_rs + (int64_t)c * c_cs] = acc[s];
        }
    }
}

// Vector form for the device-native layout (row-major B and C, k a multiple of 4, 16-byte aligned):
// FOUR lanes share a matrix row, each owns four adjacent output columns and reads its 32-byte slice of
// a B row with two 16-byte loads.  Per stored entry a lane now issues 2 loads and 8 flops instead of
// 1 load and 2 flops behind the same address / predicate arithmetic -- the 8-byte-per-lane form above
// is bound by instruction issue (~11 instructions per multiply-add), not by memory.  A workgroup still
// owns 64 rows (one lane-group each), entries staged through LDS as 32-bit column ids; the entry loop
// advances two entries per step (4 independent 16-byte loads per lane).  Deeper unrolling was slower:
// 60 registers keep 7-8 wavefronts per SIMD resident, 96 (four entries per step) only 5.  Measured on
// one box (profiles/r01_spmm_variants.log), 8-byte form -> this form: 5-point matrix x 16 columns
// 1.043 -> 0.643 ms; config 5's random pattern with the 2.1 GB gather set 1.622 -> 1.516 ms; random with
// B inside the Infinity Cache 1.247 -> 1.275 ms.  Also tried and dropped: no LDS staging (entries handed
// round a lane-group with shuffles: 0.99 / 1.34 / 1.56 ms) and an XCD-sliced block order (1.03 / 1.24 /
// 1.63 ms).  Each output column is still accumulated entry by entry in stored order: same bits.
constexpr int VG = 4;                        // lanes per row
constexpr int VCPL = KT / VG;                // 4 columns per lane
constexpr int VU = 2;                        // entries per step: 4 independent 16-byte loads per lane

typedef double vdouble2 __attribute__((ext_vector_type(2)));

template <typename I, bool SPLIT>
__global__ __launch_bounds__(TPB_MM) void spmm_rowblock_vec_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ B_own, int64_t b_rs, const double *__restrict__ B_ghost, int64_t bg_rs,
    int64_t n_own, double *__restrict__ C, int64_t c_rs, int64_t nrows, int k, int base,
    const int32_t *__restrict__ block_list, uint32_t nblocks)
{
    __shared__ double s_val[CHUNK_MM];
    __shared__ int32_t s_col[CHUNK_MM];     // own: row of B; ghost: -(1 + ghost row)

    const int tid = threadIdx.x;
    const int g = tid / VG, l = tid % VG;   // g = row of the block (0..63)
    const uint32_t b = blockIdx.x;
    const int64_t blk = block_list ? (int64_t)block_list[b] : (int64_t)b;
    const int64_t r0 = blk * RPB_MM;
    const int nr = (int)((nrows - r0) < RPB_MM ? (nrows - r0) : RPB_MM);
    const int64_t p0 = (int64_t)rowptr[r0] - base;
    const int64_t p1 = (int64_t)rowptr[r0 + nr] - base;
    const int64_t total = p1 - p0;
    int64_t lo = 0, hi = 0;
    if (g < nr) {
        lo = (int64_t)rowptr[r0 + g] - base - p0;
        hi = (int64_t)rowptr[r0 + g + 1] - base - p0;
    }

    for (int kt = 0; kt < k; kt += KT) {
        const int c = kt + VCPL * l;
        const bool col_ok = c < k;
        double acc[VCPL];
#pragma unroll
        for (int q = 0; q < VCPL; ++q) acc[q] = 0.0;

        for (int64_t ch = 0; ch < total; ch += CHUNK_MM) {
            const int n = (int)((total - ch) < CHUNK_MM ? (total - ch) : CHUNK_MM);
            __syncthreads();   // previous pass finished reading LDS
            for (int i = tid; i < n; i += TPB_MM) {
                const int64_t col = (int64_t)__builtin_nontemporal_load(colval + p0 + ch + i) - base;
                s_val[i] = __builtin_nontemporal_load(nzval + p0 + ch + i);
                s_col[i] = (SPLIT && col >= n_own) ? (int32_t)(-(1 + (col - n_own))) : (int32_t)col;
            }
            __syncthreads();
            if (!col_ok) continue;
            const int a = (int)((lo > ch ? lo : ch) - ch);
            const int e = (int)((hi < ch + n ? hi : ch + n) - ch);
            for (int j = a; j < e; j += VU) {
                int32_t cj[VU];
                double v[VU];
                bool ok[VU];
                vdouble2 b0[VU], b1[VU];
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    ok[u] = j + u < e;
                    const int idx = ok[u] ? j + u : j;
                    cj[u] = s_col[idx];
                    v[u] = s_val[idx];
                }
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    b0[u] = (vdouble2)(0.0);
                    b1[u] = (vdouble2)(0.0);
                    if (ok[u]) {
                        const double *src = (SPLIT && cj[u] < 0)
                                                ? B_ghost + (int64_t)(-cj[u] - 1) * bg_rs + c
                                                : B_own + (int64_t)cj[u] * b_rs + c;
                        b0[u] = *reinterpret_cast<const vdouble2 *>(src);
                        b1[u] = *reinterpret_cast<const vdouble2 *>(src + 2);
                    }
                }
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    if (ok[u]) {
                        acc[0] += v[u] * b0[u].x;
                        acc[1] += v[u] * b0[u].y;
                        acc[2] += v[u] * b1[u].x;
                        acc[3] += v[u] * b1[u].y;
                    }
                }
            }
        }
        if (g < nr && col_ok) {
            double *dst = C + (r0 + g) * c_rs + c;
            vdouble2 o0, o1;
            o0.x = acc[0]; o0.y = acc[1]; o1.x = acc[2]; o1.y = acc[3];
            *reinterpret_cast<vdouble2 *>(dst) = o0;
            *reinterpret_cast<vdouble2 *>(dst + 2) = o1;
        }
    }
}

// tiled transpose / layout conversion: dst(i,c) = src(i,c), arbitrary (row,col) strides
__global__ __launch_bounds__(256) void relayout_kernel(const double *__restrict__ src,
                                                       int64_t s_rs, int64_t s_cs,
                                                       double *__restrict__ dst, int64_t d_rs,
                                                       int64_t d_cs, int64_t rows, int64_t cols)
{
    __shared__ double tile[32][33];
    const int64_t tr = (int64_t)blockIdx.x * 32, tc = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    // read with the source's fast axis on tx
    const bool src_row_fast = (s_rs == 1) || (s_rs < s_cs);   // column-major source: rows fastest
    for (int j = ty; j < 32; j += 8) {
        const int64_t i = src_row_fast ? tr + tx : tr + j;
        const int64_t c = src_row_fast ? tc + j : tc + tx;
        if (i < rows && c < cols) tile[src_row_fast ? j : tx][src_row_fast ? tx : j] = src[i * s_rs + c * s_cs];
    }
    __syncthreads();
    // tile[cc][rr] holds element (tr+rr, tc+cc)
    const bool dst_row_fast = (d_rs == 1) || (d_rs < d_cs);
    for (int j = ty; j < 32; j += 8) {
        const int rr = dst_row_fast ? tx : j;
        const int cc = dst_row_fast ? j : tx;
        const int64_t i = tr + rr, c = tc + cc;
        if (i < rows && c < cols) dst[i * d_rs + c * d_cs] = tile[cc][rr];
    }
}

static inline void layout_strides(int layout, int64_t ld, int64_t *rs, int64_t *cs)
{
    if (layout == HPCLA_LAYOUT_ROW) { *rs = ld; *cs = 1; }
    else { *rs = 1; *cs = ld; }
}

template <typename I>
static int spmm_launch(const I *rowptr, const I *colval, const double *nzval, const double *B_own,
                       int64_t b_rs, int64_t b_cs, const double *B_ghost, int64_t bg_rs,
                       int64_t n_own, bool split, double *C, int64_t c_rs, int64_t c_cs,
                       int64_t nrows, int64_t nnz, int k, int index_base,
                       const int32_t *block_list, int64_t n_blocks, void *stream)
{
    if (nrows < 0 || nnz < 0 || k < 0) return set_error(HPCLA_ERR_INVALID, "spmm: negative size");
    if (index_base != 0 && index_base != 1)
        return set_error(HPCLA_ERR_INVALID, "spmm: index_base must be 0 or 1");
    if (nrows == 0 || k == 0) return HPCLA_OK;
    if (!rowptr || !C) return set_error(HPCLA_ERR_INVALID, "spmm: null rowptr/C");
    if (nnz > 0 && (!colval || !nzval || !B_own))
        return set_error(HPCLA_ERR_INVALID, "spmm: null colval/nzval/B with nnz > 0");
    const int64_t all_blocks = (nrows + RPB_MM - 1) / RPB_MM;
    int64_t launch_blocks = all_blocks;
    if (block_list) {
        if (n_blocks < 0 || n_blocks > all_blocks)
            return set_error(HPCLA_ERR_INVALID, "spmm: n_blocks out of range");
        launch_blocks = n_blocks;
    }
    if (launch_blocks == 0) return HPCLA_OK;
    if (launch_blocks > 0x7fffffffLL) return set_error(HPCLA_ERR_INVALID, "spmm: too many blocks");
    hipStream_t s = as_stream(stream);
    dim3 grid((uint32_t)launch_blocks), block(TPB_MM);
    // device-native layout: row-major B / C, k % 4 == 0, everything 16-byte aligned, 32-bit column space
    const bool vec_ok = b_cs == 1 && c_cs == 1 && (k % 4) == 0 && (b_rs % 2) == 0 && (c_rs % 2) == 0 &&
                        (!split || (bg_rs % 2) == 0) &&
                        ((reinterpret_cast<uintptr_t>(B_own) | reinterpret_cast<uintptr_t>(C) |
                          (split ? reinterpret_cast<uintptr_t>(B_ghost) : 0)) & 15) == 0 &&
                        sizeof(I) == 4;
    if (vec_ok) {
        if (split)
            spmm_rowblock_vec_kernel<I, true><<<grid, block, 0, s>>>(
                rowptr, colval, nzval, B_own, b_rs, B_ghost, bg_rs, n_own, C, c_rs, nrows, k, index_base,
                block_list, (uint32_t)launch_blocks);
        else
            spmm_rowblock_vec_kernel<I, false><<<grid, block, 0, s>>>(
                rowptr, colval, nzval, B_own, b_rs, nullptr, 0, 0, C, c_rs, nrows, k, index_base,
                block_list, (uint32_t)launch_blocks);
    } else if (split)
        spmm_rowblock_kernel<I, true><<<grid, block, 0, s>>>(
            rowptr, colval, nzval, B_own, b_rs, b_cs, B_ghost, bg_rs, n_own, C, c_rs, c_cs, nrows,
            k, index_base, block_list, (uint32_t)launch_blocks);
    else
        spmm_rowblock_kernel<I, false><<<grid, block, 0, s>>>(
            rowptr, colval, nzval, B_own, b_rs, b_cs, nullptr, 0, 0, C, c_rs, c_cs, nrows, k,
            index_base, block_list, (uint32_t)launch_blocks);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int hpcla_spmm_rows_per_block(void) { return RPB_MM; }

HPCLA_API int hpcla_spmm_csr_f64_i32(const int32_t *rowptr, const int32_t *colval,
                                     const double *nzval, const double *B, int64_t ldb,
                                     int b_layout, double *C, int64_t ldc, int c_layout,
                                     int64_t nrows, int64_t nnz, int k, int index_base,
                                     void *stream)
{
    int64_t brs, bcs, crs, ccs;
    layout_strides(b_layout, ldb, &brs, &bcs);
    layout_strides(c_layout, ldc, &crs, &ccs);
    return spmm_launch<int32_t>(rowptr, colval, nzval, B, brs, bcs, nullptr, 0, 0, false, C, crs,
                                ccs, nrows, nnz, k, index_base, nullptr, 0, stream);
}

HPCLA_API int hpcla_spmm_csr_f64_i64(const int64_t *rowptr, const int64_t *colval,
                                     const double *nzval, const double *B, int64_t ldb,
                                     int b_layout, double *C, int64_t ldc, int c_layout,
                                     int64_t nrows, int64_t nnz, int k, int index_base,
                                     void *stream)
{
    int64_t brs, bcs, crs, ccs;
    layout_strides(b_layout, ldb, &brs, &bcs);
    layout_strides(c_layout, ldc, &crs, &ccs);
    return spmm_launch<int64_t>(rowptr, colval, nzval, B, brs, bcs, nullptr, 0, 0, false, C, crs,
                                ccs, nrows, nnz, k, index_base, nullptr, 0, stream);
}

HPCLA_API int hpcla_spmm_split_f64_i32(const int32_t *rowptr, const int32_t *colval_split,
                                       const double *nzval, const double *B_own, int64_t ldb_own,
                                       const double *B_ghost, int64_t ldb_ghost, int64_t n_own,
                                       double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                       int index_base, const int32_t *block_list,
                                       int64_t n_blocks, void *stream)
{
    return spmm_launch<int32_t>(rowptr, colval_split, nzval, B_own, ldb_own, 1, B_ghost, ldb_ghost,
                                n_own, true, C, ldc, 1, nrows, nnz, k, index_base, block_list,
                                n_blocks, stream);
}

HPCLA_API int hpcla_spmm_split_f64_i64(const int64_t *rowptr, const int64_t *colval_split,
                                       const double *nzval, const double *B_own, int64_t ldb_own,
                                       const double *B_ghost, int64_t ldb_ghost, int64_t n_own,
                                       double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                       int index_base, const int32_t *block_list,
                                       int64_t n_blocks, void *stream)
{
    return spmm_launch<int64_t>(rowptr, colval_split, nzval, B_own, ldb_own, 1, B_ghost, ldb_ghost,
                                n_own, true, C, ldc, 1, nrows, nnz, k, index_base, block_list,
                                n_blocks, stream);
}

HPCLA_API int hpcla_transpose_f64(const double *src, int64_t ld_src, int src_layout, double *dst,
                                  int64_t ld_dst, int dst_layout, int64_t rows, int64_t cols,
                                  void *stream)
{
    if (rows < 0 || cols < 0) return set_error(HPCLA_ERR_INVALID, "transpose: negative size");
    if (rows == 0 || cols == 0) return HPCLA_OK;
    if (!src || !dst) return set_error(HPCLA_ERR_INVALID, "transpose: null pointer");
    int64_t srs, scs, drs, dcs;
    layout_strides(src_layout, ld_src, &srs, &scs);
    layout_strides(dst_layout, ld_dst, &drs, &dcs);
    dim3 grid((uint32_t)((rows + 31) / 32), (uint32_t)((cols + 31) / 32));
    relayout_kernel<<<grid, 256, 0, as_stream(stream)>>>(src, srs, scs, dst, drs, dcs, rows, cols);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}
