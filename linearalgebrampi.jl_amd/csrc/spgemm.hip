// spgemm.hip -- sparse x sparse local product C = A * G on the device (SURVEY.md 8f "next" rank 3).
//
// Reference: Base.:*(A::HPCSparseMatrix, B::HPCSparseMatrix) (src/sparse.jl:991-1059): a MatrixPlan
// gathers the rows of B named by A.col_indices (src/sparse.jl:554-978), then the local product is
// Julia's SparseArrays multiply on the CPU (`CT = plan.AT * A_csc`, :1011), even for GPU backends.
// That multiply is Gustavson's algorithm: for every local row i, for the stored entries k of row i in
// ascending order, for every entry j of gathered row k:  C[i,j] (+)= G[k,j] * A[i,k]  -- so each C[i,j]
// is the sum over k ASCENDING of separately rounded products, the first product assigned directly.
//
// Here `G` is the gathered-rows matrix (row r of G = row A.col_indices[r] of B, GLOBAL column ids).
// One wavefront (short rows, 4 rows per workgroup) or one workgroup (long rows) owns an output row and
// walks k sequentially; its lanes insert/accumulate the entries of G's row k into an LDS hash table
// in parallel.  The columns within one row of G are distinct, so a (i,j) accumulator is touched by
// at most one lane per k and the per-entry order over k is preserved: results are bit-identical to
// the reference multiply.  The row is then sorted by column (bitonic sort in LDS; empty slots sort
// last) and written to an upper-bound-sized slot; the host turns the counts into rowptr and a
// compaction kernel produces the final CSR arrays.  No float atomics, no MFMA.
#include "common.h"

namespace hpcla {

constexpr unsigned long long EMPTY_KEY = ~0ULL;

__device__ __forceinline__ uint32_t hash_col(unsigned long long c, uint32_t mask)
{
    return (uint32_t)((c * 0x9E3779B97F4A7C15ULL) >> 40) & mask;
}

// find-or-insert with linear probing (table never fills: capacity > upper bound of the row)
__device__ __forceinline__ int table_slot(unsigned long long *keys, uint32_t mask, unsigned long long col)
{
    uint32_t s = hash_col(col, mask);
    while (true) {
        const unsigned long long k = keys[s];
        if (k == col) return (int)s;
        if (k == EMPTY_KEY) {
            const unsigned long long prev = atomicCAS(&keys[s], EMPTY_KEY, col);
            if (prev == EMPTY_KEY || prev == col) return (int)s;
        }
        s = (s + 1) & mask;
    }
}

// upper bound of row i: sum of the lengths of the gathered rows it references
template <typename I>
__global__ __launch_bounds__(256) void spgemm_ub_kernel(const I *__restrict__ a_rowptr,
                                                        const I *__restrict__ a_col, int base,
                                                        const int64_t *__restrict__ g_rowptr,
                                                        int64_t nrows, int64_t *__restrict__ ub)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    int64_t s = 0;
    for (int64_t p = (int64_t)a_rowptr[r] - base; p < (int64_t)a_rowptr[r + 1] - base; ++p) {
        const int64_t k = (int64_t)a_col[p] - base;
        s += g_rowptr[k + 1] - g_rowptr[k];
    }
    ub[r] = s;
}

// GROUP = 64: one wavefront per row (wave-level ordering only); GROUP = 256: one workgroup per row
template <int GROUP>
__device__ __forceinline__ void group_sync()
{
    if (GROUP == 256) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

template <typename I, int GROUP, int TABLE>
__global__ __launch_bounds__(256) void spgemm_numeric_kernel(
    const I *__restrict__ a_rowptr, const I *__restrict__ a_col, const double *__restrict__ a_val, int base,
    const int64_t *__restrict__ g_rowptr, const int64_t *__restrict__ g_col,
    const double *__restrict__ g_val, const int32_t *__restrict__ row_list, int64_t n_list,
    const int64_t *__restrict__ ub_prefix, int64_t *__restrict__ c_col_tmp,
    double *__restrict__ c_val_tmp, int64_t *__restrict__ cnt)
{
    constexpr int GROUPS = 256 / GROUP;
    constexpr uint32_t mask = TABLE - 1;
    __shared__ unsigned long long s_keys[GROUPS * TABLE];
    __shared__ double s_vals[GROUPS * TABLE];
    __shared__ int s_cnt[GROUPS];
    const int g = threadIdx.x / GROUP, t = threadIdx.x % GROUP;
    unsigned long long *keys = s_keys + g * TABLE;
    double *vals = s_vals + g * TABLE;
    const int64_t li = (int64_t)blockIdx.x * GROUPS + g;
    // a fresh accumulator holds -0.0: (-0.0) + p == p bitwise for every p, i.e. the reference's
    // "assign the first product" without a separate first-touch test
    for (int s = t; s < TABLE; s += GROUP) { keys[s] = EMPTY_KEY; vals[s] = -0.0; }
    if (t == 0) s_cnt[g] = 0;
    __syncthreads();
    if (li >= n_list) return;          // (GROUP == 64: only wave-level ordering below)
    const int64_t row = row_list[li];

    // Gustavson: k ascending (sequential); the entries of gathered row k in parallel over the lanes
    for (int64_t p = (int64_t)a_rowptr[row] - base; p < (int64_t)a_rowptr[row + 1] - base; ++p) {
        const int64_t k = (int64_t)a_col[p] - base;
        const double av = a_val[p];
        const int64_t q1 = g_rowptr[k + 1];
        for (int64_t q = g_rowptr[k] + t; q < q1; q += GROUP) {
            const double prod = g_val[q] * av;
            const int s = table_slot(keys, mask, (unsigned long long)g_col[q]);
            vals[s] = vals[s] + prod;
        }
        group_sync<GROUP>();           // the next k may hit the same columns
    }

    // bitonic sort of the whole table by key (empty slots = all ones sort last)
    for (int k2 = 2; k2 <= TABLE; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = t; i < TABLE; i += GROUP) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k2) == 0;
                    const unsigned long long a = keys[i], b = keys[ixj];
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[ixj] = a;
                        const double tv = vals[i];
                        vals[i] = vals[ixj];
                        vals[ixj] = tv;
                    }
                }
            }
            group_sync<GROUP>();
        }
    }
    int c = 0;
    for (int s = t; s < TABLE; s += GROUP) c += keys[s] != EMPTY_KEY ? 1 : 0;
    if (c) atomicAdd(&s_cnt[g], c);
    group_sync<GROUP>();
    const int n_out = s_cnt[g];
    const int64_t off = ub_prefix[row];
    for (int s = t; s < n_out; s += GROUP) {
        c_col_tmp[off + s] = (int64_t)keys[s];
        c_val_tmp[off + s] = vals[s];
    }
    if (t == 0) cnt[row] = n_out;
}

__global__ __launch_bounds__(256) void spgemm_compact_kernel(const int64_t *__restrict__ c_rowptr,
                                                             const int64_t *__restrict__ ub_prefix,
                                                             int64_t nrows,
                                                             const int64_t *__restrict__ c_col_tmp,
                                                             const double *__restrict__ c_val_tmp,
                                                             int64_t *__restrict__ c_col,
                                                             double *__restrict__ c_val)
{
    // one wavefront per row
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int64_t dst = c_rowptr[row], n = c_rowptr[row + 1] - dst, src = ub_prefix[row];
    for (int64_t s = lane; s < n; s += 64) {
        c_col[dst + s] = c_col_tmp[src + s];
        c_val[dst + s] = c_val_tmp[src + s];
    }
}

template <typename I>
static int numeric_launch(int bin, const I *a_rowptr, const I *a_col, const double *a_val, int base,
                          const int64_t *g_rowptr, const int64_t *g_col, const double *g_val,
                          const int32_t *row_list, int64_t n_list, const int64_t *ub_prefix,
                          int64_t *c_col_tmp, double *c_val_tmp, int64_t *cnt, hipStream_t s)
{
    if (n_list == 0) return HPCLA_OK;
#define NUM_ARGS a_rowptr, a_col, a_val, base, g_rowptr, g_col, g_val, row_list, n_list, ub_prefix, c_col_tmp, c_val_tmp, cnt
    switch (bin) {
        case 0: spgemm_numeric_kernel<I, 64, 32><<<(uint32_t)((n_list + 3) / 4), 256, 0, s>>>(NUM_ARGS); break;
        case 1: spgemm_numeric_kernel<I, 64, 128><<<(uint32_t)((n_list + 3) / 4), 256, 0, s>>>(NUM_ARGS); break;
        case 2: spgemm_numeric_kernel<I, 64, 512><<<(uint32_t)((n_list + 3) / 4), 256, 0, s>>>(NUM_ARGS); break;
        case 3: spgemm_numeric_kernel<I, 256, 8192><<<(uint32_t)n_list, 256, 0, s>>>(NUM_ARGS); break;
        default: return set_error(HPCLA_ERR_INVALID, "spgemm: bad bin %d", bin);
    }
#undef NUM_ARGS
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

}  // namespace hpcla

using namespace hpcla;

// Bin b handles rows whose upper bound is <= hpcla_spgemm_bin_cap(b) (0..3); larger rows are not
// supported by this build (HPCLA_ERR_UNSUPPORTED is the caller's to raise).
HPCLA_API int64_t hpcla_spgemm_bin_cap(int bin)
{
    static const int64_t cap[4] = {24, 96, 384, 6144};     // <= 75 % of table sizes 32 / 128 / 512 / 8192
    return (bin >= 0 && bin < 4) ? cap[bin] : -1;
}

#define SPGEMM_API(SFX, ITYPE)                                                                       \
    HPCLA_API int hpcla_spgemm_ub_##SFX(const ITYPE *a_rowptr, const ITYPE *a_col, int64_t nrows,   \
                                        int index_base, const int64_t *g_rowptr, int64_t *ub_out,   \
                                        void *stream)                                                \
    {                                                                                                \
        if (nrows < 0) return set_error(HPCLA_ERR_INVALID, "spgemm_ub: negative size");             \
        if (nrows == 0) return HPCLA_OK;                                                             \
        if (!a_rowptr || !g_rowptr || !ub_out) return set_error(HPCLA_ERR_INVALID, "spgemm_ub: null pointer"); \
        spgemm_ub_kernel<ITYPE><<<(uint32_t)((nrows + 255) / 256), 256, 0, as_stream(stream)>>>(     \
            a_rowptr, a_col, index_base, g_rowptr, nrows, ub_out);                                   \
        HPCLA_CHECK_LAUNCH();                                                                        \
        return HPCLA_OK;                                                                             \
    }                                                                                                \
    HPCLA_API int hpcla_spgemm_numeric_##SFX(                                                        \
        int bin, const ITYPE *a_rowptr, const ITYPE *a_col, const double *a_val, int index_base,    \
        const int64_t *g_rowptr, const int64_t *g_col, const double *g_val, const int32_t *row_list, \
        int64_t n_list, const int64_t *ub_prefix, int64_t *c_col_tmp, double *c_val_tmp, int64_t *cnt, \
        void *stream)                                                                                \
    {                                                                                                \
        if (n_list < 0) return set_error(HPCLA_ERR_INVALID, "spgemm_numeric: negative size");       \
        if (n_list > 0 && (!a_rowptr || !a_col || !a_val || !g_rowptr || !row_list || !ub_prefix || !cnt)) \
            return set_error(HPCLA_ERR_INVALID, "spgemm_numeric: null pointer");                    \
        return numeric_launch<ITYPE>(bin, a_rowptr, a_col, a_val, index_base, g_rowptr, g_col, g_val, \
                                     row_list, n_list, ub_prefix, c_col_tmp, c_val_tmp, cnt,         \
                                     as_stream(stream));                                             \
    }
SPGEMM_API(i32, int32_t)
SPGEMM_API(i64, int64_t)

HPCLA_API int hpcla_spgemm_compact(const int64_t *c_rowptr, const int64_t *ub_prefix, int64_t nrows,
                                   const int64_t *c_col_tmp, const double *c_val_tmp, int64_t *c_col,
                                   double *c_val, void *stream)
{
    if (nrows < 0) return set_error(HPCLA_ERR_INVALID, "spgemm_compact: negative size");
    if (nrows == 0) return HPCLA_OK;
    if (!c_rowptr || !ub_prefix) return set_error(HPCLA_ERR_INVALID, "spgemm_compact: null pointer");
    spgemm_compact_kernel<<<(uint32_t)((nrows + 3) / 4), 256, 0, as_stream(stream)>>>(
        c_rowptr, ub_prefix, nrows, c_col_tmp, c_val_tmp, c_col, c_val);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}
