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
// Rows with at most 64 candidate products take the register expand-sort-combine kernel further down.
// Otherwise one wavefront (4 rows per workgroup) or one workgroup (long rows) owns an output row and
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

// ---- short rows: expand - sort - combine inside one (sub-)wavefront, no LDS ----------------------
// A row whose upper bound is <= L is owned by L lanes (L = 16 / 32 / 64: 16 / 8 / 4 rows per
// workgroup).  Expand: the lanes first act as the row's A entries (k, length and start of gathered row
// k, A value), an inclusive scan of the lengths numbers the products, and every product slot t finds
// its A entry by a lower-bound search over the scan (register shuffles); all products of the row are
// then loaded at once -- one memory round trip instead of one per k.  Sort: bitonic network over the
// L lanes on the key (column * 64 + slot); slots ascend with k, so equal columns end up adjacent and
// in ascending-k order.  Combine: the head lane of each run adds the following products one by one
// (the reference's order: first product assigned, then += in ascending k), heads are ranked with a
// ballot and written in column order.  Same bits as the hash kernels, ~4 dependent memory latencies
// per row instead of ~3 per k, and 4 x - 16 x more rows in flight.
template <int L>
__device__ __forceinline__ unsigned long long group_ballot(bool pred, int lane)
{
    const unsigned long long m = __ballot(pred);
    if (L == 64) return m;
    return (m >> ((lane / L) * L)) & ((1ULL << L) - 1ULL);
}

template <typename I, int L>
__global__ __launch_bounds__(256) void spgemm_esc_kernel(
    const I *__restrict__ a_rowptr, const I *__restrict__ a_col, const double *__restrict__ a_val, int base,
    const int64_t *__restrict__ g_rowptr, const int64_t *__restrict__ g_col,
    const double *__restrict__ g_val, const int32_t *__restrict__ row_list, int64_t n_list,
    const int64_t *__restrict__ ub_prefix, int64_t *__restrict__ c_col_tmp,
    double *__restrict__ c_val_tmp, int64_t *__restrict__ cnt)
{
    constexpr int GROUPS = 256 / L;
    constexpr unsigned long long INVALID = ~0ULL;
    const int lane = threadIdx.x & 63;
    const int t = threadIdx.x % L;
    const int64_t li = (int64_t)blockIdx.x * GROUPS + threadIdx.x / L;
    const bool active = li < n_list;
    const int64_t row = active ? (int64_t)row_list[li] : 0;
    const int64_t pa0 = active ? (int64_t)a_rowptr[row] - base : 0;
    const int nk = active ? (int)((int64_t)a_rowptr[row + 1] - base - pa0) : 0;

    // expand
    int64_t my_q = -1;
    double my_av = 0.0;
    int base_slot = 0;
    for (int pc = 0; __any(pc < nk); pc += L) {
        const int p = pc + t;
        const bool has = p < nk;
        int64_t gs = 0;
        int len = 0;
        double av = 0.0;
        if (has) {
            const int64_t k = (int64_t)a_col[pa0 + p] - base;
            gs = g_rowptr[k];
            len = (int)(g_rowptr[k + 1] - gs);
            av = a_val[pa0 + p];
        }
        int incl = len;
#pragma unroll
        for (int d = 1; d < L; d <<= 1) {
            const int nb = __shfl_up(incl, d, L);
            if (t >= d) incl += nb;
        }
        const int total = __shfl(incl, L - 1, L);
        const int inclb = incl + base_slot;
        // smallest j with inclb_j > t (lower bound over the non-decreasing scan)
        int j = 0;
#pragma unroll
        for (int step = L / 2; step > 0; step >>= 1) {
            const int v = __shfl(inclb, j + step - 1, L);
            if (v <= t) j += step;
        }
        const int64_t o_gs = __shfl(gs, j, L);
        const int o_excl = __shfl(inclb - len, j, L);
        const double o_av = __shfl(av, j, L);
        if (my_q < 0 && t >= base_slot && t < base_slot + total) {
            my_q = o_gs + (t - o_excl);
            my_av = o_av;
        }
        base_slot += total;
    }
    unsigned long long key = INVALID;
    double val = 0.0;
    if (my_q >= 0) {
        val = g_val[my_q] * my_av;
        key = ((unsigned long long)g_col[my_q] << 6) | (unsigned long long)t;
    }
    // sort (ascending; invalid keys last)
#pragma unroll
    for (int k2 = 2; k2 <= L; k2 <<= 1) {
#pragma unroll
        for (int jj = k2 >> 1; jj > 0; jj >>= 1) {
            const unsigned long long okey = __shfl_xor(key, jj, L);
            const double oval = __shfl_xor(val, jj, L);
            const bool take_min = ((t & k2) == 0) == ((t & jj) == 0);
            if (take_min ? (okey < key) : (okey > key)) {
                key = okey;
                val = oval;
            }
        }
    }
    // combine runs of equal columns in slot (= ascending k) order
    const bool valid = key != INVALID;
    const unsigned long long col = key >> 6;
    const unsigned long long pcol = __shfl_up(col, 1, L);
    const bool pvalid = __shfl_up((int)valid, 1, L) != 0;
    const bool head = valid && (t == 0 || !pvalid || pcol != col);
    double sum = val;
    bool alive = head;
    for (int d = 1; d < L && __any(alive); ++d) {
        const unsigned long long ncol = __shfl_down(col, d, L);
        const double nval = __shfl_down(val, d, L);
        const bool nvalid = __shfl_down((int)valid, d, L) != 0;
        if (alive && t + d < L && nvalid && ncol == col) sum += nval;
        else alive = false;
    }
    const unsigned long long gm = group_ballot<L>(head, lane);
    if (active) {
        const int64_t off = ub_prefix[row];
        if (head) {
            const int rank = __popcll(gm & ((1ULL << t) - 1ULL));
            c_col_tmp[off + rank] = (int64_t)col;
            c_val_tmp[off + rank] = sum;
        }
        if (t == 0) cnt[row] = __popcll(gm);
    }
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

// Repeated products on a KNOWN structure (the MatrixPlan caches it): every result entry e owns the list of its products
// -- pairs (index into A's values, index into the gathered B values), in ascending k = ascending A entry -- built once at
// plan level (matmat.py: expand, stable sort by (row, column), run lengths).  The numeric product is then one streaming
// pass: 8 B per product + 4 or 8 B + 8 B per result entry, the two value gathers mostly from L2; no expansion, no sort
// network, no ballot ranking.  Each entry is the first product plus the others added one by one in list order: the
// accumulation of the expand-sort-combine and hash kernels above, so the same bits (tested).
template <int R>
__device__ __forceinline__ void mapped_pass(const int2 *__restrict__ src, int n, const double *__restrict__ a_val,
                                            const double *__restrict__ g_val, double *s_prod, int tid)
{
    int2 q[R];
    double av[R], gv[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int idx = u * 256 + tid;
        q[u] = src[idx < n ? idx : n - 1];
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        av[u] = a_val[q[u].x];
        gv[u] = g_val[q[u].y];
    }
#pragma unroll
    for (int u = 0; u < R; ++u) s_prod[u * 256 + tid] = gv[u] * av[u];
}

template <typename P>
__global__ __launch_bounds__(256) void spgemm_mapped_kernel(const P *__restrict__ pair_ptr, const int2 *__restrict__ pairs,
                                                            const double *__restrict__ a_val,
                                                            const double *__restrict__ g_val,
                                                            double *__restrict__ c_val, int64_t nnz_c)
{
    // The SpMV kernel's shape (spmv.hip) with result entries as rows and products as nonzeros: a workgroup owns 256
    // consecutive result entries, whose products are ONE contiguous range of the pair list; the range is streamed with
    // coalesced 8-byte loads (up to four pairs per lane per pass, all loads first, then the value gathers), every
    // product is parked in LDS, then thread t adds up entry t's products in list order.  A lane past the pass's end
    // re-reads the last pair (one broadcast line) and parks a product nobody reads: straight-line code.
    constexpr int EPB = 256, MCHUNK = EPB * 4;
    __shared__ double s_prod[MCHUNK];
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int nr = (int)((nnz_c - e0) < EPB ? (nnz_c - e0) : EPB);
    const int64_t p0 = (int64_t)pair_ptr[e0], p1 = (int64_t)pair_ptr[e0 + nr];
    P rlo = 0, rhi = 0;                                // used in the sum phase: the load runs under the stream
    if (tid < nr) {
        rlo = pair_ptr[e0 + tid];
        rhi = pair_ptr[e0 + tid + 1];
    }
    const int64_t total = p1 - p0;
    double acc = 0.0;
    for (int64_t c = 0; c < total; c += MCHUNK) {
        const int n = (int)((total - c) < MCHUNK ? (total - c) : MCHUNK);
        const int2 *src = pairs + p0 + c;
        // as many rounds of 256 products as the pass holds (workgroup-uniform), each variant straight-line
        if (n > 3 * EPB) mapped_pass<4>(src, n, a_val, g_val, s_prod, tid);
        else if (n > 2 * EPB) mapped_pass<3>(src, n, a_val, g_val, s_prod, tid);
        else if (n > EPB) mapped_pass<2>(src, n, a_val, g_val, s_prod, tid);
        else mapped_pass<1>(src, n, a_val, g_val, s_prod, tid);
        __syncthreads();
        {
            const int64_t lo = tid < nr ? (int64_t)rlo - p0 : 0, hi = tid < nr ? (int64_t)rhi - p0 : 0;
            const int64_t a = lo > c ? lo : c;
            const int64_t e = hi < c + n ? hi : c + n;
            for (int64_t j = a; j < e; ++j) acc = (j == lo) ? s_prod[j - c] : acc + s_prod[j - c];
        }
        __syncthreads();
    }
    if (tid < nr) c_val[e0 + tid] = acc;
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
        case 0: spgemm_esc_kernel<I, 16><<<(uint32_t)((n_list + 15) / 16), 256, 0, s>>>(NUM_ARGS); break;
        case 1: spgemm_esc_kernel<I, 32><<<(uint32_t)((n_list + 7) / 8), 256, 0, s>>>(NUM_ARGS); break;
        case 2: spgemm_esc_kernel<I, 64><<<(uint32_t)((n_list + 3) / 4), 256, 0, s>>>(NUM_ARGS); break;
        case 3: spgemm_numeric_kernel<I, 64, 512><<<(uint32_t)((n_list + 3) / 4), 256, 0, s>>>(NUM_ARGS); break;
        case 4: spgemm_numeric_kernel<I, 256, 8192><<<(uint32_t)n_list, 256, 0, s>>>(NUM_ARGS); break;
        default: return set_error(HPCLA_ERR_INVALID, "spgemm: bad bin %d", bin);
    }
#undef NUM_ARGS
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

}  // namespace hpcla

using namespace hpcla;

// Bin b handles rows whose upper bound is <= hpcla_spgemm_bin_cap(b) (b = 0..4, -1 beyond): bins 0-2
// are the register expand-sort-combine kernel with 16 / 32 / 64 lanes per row, bins 3-4 the LDS hash
// kernels (tables of 512 per wavefront / 8192 per workgroup, filled to <= 75 %).  Larger rows are not
// supported by this build (HPCLA_ERR_UNSUPPORTED is the caller's to raise).
HPCLA_API int64_t hpcla_spgemm_bin_cap(int bin)
{
    static const int64_t cap[5] = {16, 32, 64, 384, 6144};
    return (bin >= 0 && bin < 5) ? cap[bin] : -1;
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

HPCLA_API int hpcla_spgemm_numeric_mapped_f64(const void *pair_ptr, int ptr_is_i64, const int32_t *pairs,
                                              const double *a_val, const double *g_val, double *c_val,
                                              int64_t nnz_c, void *stream)
{
    if (nnz_c < 0) return set_error(HPCLA_ERR_INVALID, "spgemm_numeric_mapped: negative size");
    if (nnz_c == 0) return HPCLA_OK;
    if (!pair_ptr || !pairs || !a_val || !g_val || !c_val)
        return set_error(HPCLA_ERR_INVALID, "spgemm_numeric_mapped: null pointer");
    if (reinterpret_cast<uintptr_t>(pairs) & 7)
        return set_error(HPCLA_ERR_INVALID, "spgemm_numeric_mapped: the pair list must be 8-byte aligned");
    const int64_t blocks = (nnz_c + 255) / 256;
    HPCLA_CHECK_GRID(blocks, "spgemm_numeric_mapped");
    const int2 *pr = reinterpret_cast<const int2 *>(pairs);
    if (ptr_is_i64)
        spgemm_mapped_kernel<int64_t><<<(uint32_t)blocks, 256, 0, as_stream(stream)>>>(
            static_cast<const int64_t *>(pair_ptr), pr, a_val, g_val, c_val, nnz_c);
    else
        spgemm_mapped_kernel<int32_t><<<(uint32_t)blocks, 256, 0, as_stream(stream)>>>(
            static_cast<const int32_t *>(pair_ptr), pr, a_val, g_val, c_val, nnz_c);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

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
