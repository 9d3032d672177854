/*
 * hpcla_oracle.c -- CPU restatement of the HPCLinearAlgebra.jl SpMV/SpMM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP kernels in
 * linearalgebrampi.jl_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never calls it.
 *
 * Parity pin: the reference is Julia and cannot run in this environment, and it ships no
 * golden output files.  The oracle is pinned against the closed-form inputs of the
 * reference's own tests (the JSON files under tests/golden/, generated with exact rational arithmetic by
 * tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  The only deliberate change is 1-based -> 0-based indexing.
 *
 * Arithmetic: the reference accumulates `acc += nzval[j] * x[colval[j]]` (src/sparse.jl:2061)
 * in stored order; Julia does not contract mul+add into an FMA, so this file MUST be built
 * with -ffp-contract=off (see oracle/Makefile).  The HIP kernels are built the same way and
 * sum in the same order, which is why the GPU parity tests can demand bit-equality.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Counter-based RNG: u(seed,i) = (splitmix64(seed + GOLDEN*(i+1)) >> 11) * 2^-53
 * Not part of the reference (Julia's MersenneTwister stream cannot be reproduced offline);
 * it defines the synthetic inputs every language/rank/device can regenerate.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline double u01(uint64_t seed, uint64_t i)
{
    uint64_t z = splitmix64(seed + 0x9E3779B97F4A7C15ULL * (i + 1));
    return (double)(z >> 11) * 0x1.0p-53;
}

ORC_API void orc_fill_uniform(double *v, int64_t start, int64_t count, uint64_t seed)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i) v[i] = u01(seed, (uint64_t)(start + i));
}

/* ------------------------------------------------------------------------------------------
 * uniform_partition  (src/HPCLinearAlgebra.jl:279-289)
 * part has nranks+1 entries; 0-based boundaries: rank r owns [part[r], part[r+1]).
 * (reference: 1-based, partition[1]=1, partition[end]=n+1.)
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_uniform_partition(int64_t n, int nranks, int64_t *part)
{
    int64_t per_rank = n / nranks;
    int64_t remainder = n % nranks;
    part[0] = 0;
    for (int r = 1; r <= nranks; ++r) {
        int64_t extra = (r <= remainder) ? 1 : 0;
        part[r] = part[r - 1] + per_rank + extra;
    }
}

/* ------------------------------------------------------------------------------------------
 * 2-D 5-point Laplacian rows, create_2d_laplacian (test/test_factorization.jl:60-102):
 * idx = (j-1)*nx + i, diagonal 4, neighbours -1 at idx-1 (i>1), idx+1 (i<nx), idx-nx (j>1),
 * idx+nx (j<ny).  Julia's sparse() sorts entries of a row by column, so the stored order is
 * ascending column: [idx-nx, idx-1, idx, idx+1, idx+nx].
 * Emits rows [row_start,row_end) with GLOBAL 0-based columns; rowptr is 0-based, local.
 * Call with colidx==NULL to only count (returns nnz).
 * ---------------------------------------------------------------------------------------- */
ORC_API int64_t orc_poisson2d_rows(int64_t nx, int64_t ny, int64_t row_start, int64_t row_end,
                                   int64_t *rowptr, int64_t *colidx, double *vals)
{
    int64_t nloc = row_end - row_start;
    /* pass 1: counts (closed form per row) */
    int64_t nnz = 0;
    for (int64_t r = 0; r < nloc; ++r) {
        int64_t idx = row_start + r;
        int64_t i = idx % nx, j = idx / nx;
        int c = 1 + (i > 0) + (i < nx - 1) + (j > 0) + (j < ny - 1);
        if (rowptr) rowptr[r] = nnz;
        nnz += c;
    }
    if (rowptr) rowptr[nloc] = nnz;
    if (!colidx) return nnz;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nloc; ++r) {
        int64_t idx = row_start + r;
        int64_t i = idx % nx, j = idx / nx;
        int64_t p = rowptr[r];
        if (j > 0)      { colidx[p] = idx - nx; vals[p++] = -1.0; }
        if (i > 0)      { colidx[p] = idx - 1;  vals[p++] = -1.0; }
        colidx[p] = idx; vals[p++] = 4.0;
        if (i < nx - 1) { colidx[p] = idx + 1;  vals[p++] = -1.0; }
        if (j < ny - 1) { colidx[p] = idx + nx; vals[p++] = -1.0; }
    }
    return nnz;
}

/* 3-D 7-point Laplacian (diag 6, six -1 neighbours,
 * idx = ((k-1)N + (j-1))N + i) -- the 3-D analogue of create_2d_laplacian above. */
ORC_API int64_t orc_poisson3d_rows(int64_t nx, int64_t ny, int64_t nz, int64_t row_start,
                                   int64_t row_end, int64_t *rowptr, int64_t *colidx, double *vals)
{
    int64_t nloc = row_end - row_start;
    int64_t nnz = 0;
    int64_t nxy = nx * ny;
    for (int64_t r = 0; r < nloc; ++r) {
        int64_t idx = row_start + r;
        int64_t i = idx % nx, j = (idx / nx) % ny, k = idx / nxy;
        int c = 1 + (i > 0) + (i < nx - 1) + (j > 0) + (j < ny - 1) + (k > 0) + (k < nz - 1);
        if (rowptr) rowptr[r] = nnz;
        nnz += c;
    }
    if (rowptr) rowptr[nloc] = nnz;
    if (!colidx) return nnz;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nloc; ++r) {
        int64_t idx = row_start + r;
        int64_t i = idx % nx, j = (idx / nx) % ny, k = idx / nxy;
        int64_t p = rowptr[r];
        if (k > 0)      { colidx[p] = idx - nxy; vals[p++] = -1.0; }
        if (j > 0)      { colidx[p] = idx - nx;  vals[p++] = -1.0; }
        if (i > 0)      { colidx[p] = idx - 1;   vals[p++] = -1.0; }
        colidx[p] = idx; vals[p++] = 6.0;
        if (i < nx - 1) { colidx[p] = idx + 1;   vals[p++] = -1.0; }
        if (j < ny - 1) { colidx[p] = idx + nx;  vals[p++] = -1.0; }
        if (k < nz - 1) { colidx[p] = idx + nxy; vals[p++] = -1.0; }
    }
    return nnz;
}

/* ------------------------------------------------------------------------------------------
 * sprand-like rows (stands in for Julia's sprand(m,n,p), tools/benchmark_single_rank.jl:48-71 and
 * BASELINE.json configs[0], configs[4]): every (i,j) present independently with probability p,
 * columns ascending, values U[0,1).  Bernoulli(p) per entry is sampled exactly by geometric gap
 * skipping with a per-row counter-based stream, so any rank regenerates its own rows.
 * Two calls: colidx==NULL counts and fills rowptr; second call fills.
 * ---------------------------------------------------------------------------------------- */
static inline int64_t sprand_row(int64_t row, int64_t ncols, double log1mp, uint64_t seed_struct,
                                 uint64_t seed_vals, int64_t *cols, double *vals)
{
    uint64_t rs = splitmix64(seed_struct ^ (0xD1B54A32D192ED03ULL * (uint64_t)(row + 1)));
    uint64_t vs = splitmix64(seed_vals ^ (0xD1B54A32D192ED03ULL * (uint64_t)(row + 1)));
    int64_t c = -1, cnt = 0;
    for (uint64_t t = 0;; ++t) {
        double u = u01(rs, t);
        /* gap >= 1, P(gap = g) = (1-p)^(g-1) p */
        double g = floor(log1p(-u) / log1mp);
        if (!(g < 9.0e18)) break;
        c += 1 + (int64_t)g;
        if (c >= ncols) break;
        if (cols) { cols[cnt] = c; vals[cnt] = u01(vs, (uint64_t)cnt); }
        ++cnt;
    }
    return cnt;
}

ORC_API int64_t orc_sprand_rows(int64_t ncols, double p, uint64_t seed_struct, uint64_t seed_vals,
                                int64_t row_start, int64_t row_end, int64_t *rowptr,
                                int64_t *colidx, double *vals)
{
    int64_t nloc = row_end - row_start;
    double log1mp = log1p(-p);
    if (!colidx) {
        int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nloc > 0 ? nloc : 1));
#pragma omp parallel for schedule(dynamic, 1024)
        for (int64_t r = 0; r < nloc; ++r)
            cnt[r] = sprand_row(row_start + r, ncols, log1mp, seed_struct, seed_vals, NULL, NULL);
        int64_t nnz = 0;
        for (int64_t r = 0; r < nloc; ++r) { rowptr[r] = nnz; nnz += cnt[r]; }
        rowptr[nloc] = nnz;
        free(cnt);
        return nnz;
    }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t r = 0; r < nloc; ++r)
        sprand_row(row_start + r, ncols, log1mp, seed_struct, seed_vals, colidx + rowptr[r],
                   vals + rowptr[r]);
    return rowptr[nloc];
}

/* ------------------------------------------------------------------------------------------
 * _spmv_kernel!  (src/sparse.jl:2055-2066): y[row] = sum_j nzval[j] * x[colval[j]], sequential
 * in stored order, acc starts at zero, no alpha/beta.  The reference runs this loop on
 * KernelAbstractions' CPU backend over Julia threads (src/sparse.jl:2077-2080); here OpenMP
 * static over rows.  nthreads<=0 -> OpenMP default.
 * ---------------------------------------------------------------------------------------- */
#define ORC_SPMV(NAME, ITYPE, VTYPE)                                                              \
    ORC_API void NAME(const ITYPE *rowptr, const ITYPE *colval, const VTYPE *nzval,               \
                      const VTYPE *x, VTYPE *y, int64_t nrows, int base, int nthreads)            \
    {                                                                                             \
        _Pragma("omp parallel for schedule(static) if (nthreads != 1)") for (int64_t row = 0;     \
                                                                             row < nrows; ++row)  \
        {                                                                                         \
            VTYPE acc = 0;                                                                        \
            for (int64_t j = (int64_t)rowptr[row] - base; j < (int64_t)rowptr[row + 1] - base;    \
                 ++j)                                                                             \
                acc += nzval[j] * x[(int64_t)colval[j] - base];                                   \
            y[row] = acc;                                                                         \
        }                                                                                         \
    }
ORC_SPMV(orc_spmv_i32, int32_t, double)
ORC_SPMV(orc_spmv_i64, int64_t, double)
/* the same loop with T = Float32 (`acc = zero(T)`, src/sparse.jl:2059; the reference's GPU configurations run it with
 * Float32 and Float64, test/test_utils.jl:62-80): float accumulator, float product, separately rounded */
ORC_SPMV(orc_spmv_f32_i32, int32_t, float)
ORC_SPMV(orc_spmv_f32_i64, int64_t, float)

ORC_API void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* |A|*|x| row scale used by the componentwise parity bound. Not in the
 * reference; test helper. */
ORC_API void orc_abs_spmv_i32(const int32_t *rowptr, const int32_t *colval, const double *nzval,
                              const double *x, double *y, int64_t nrows, int base)
{
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < nrows; ++row) {
        double acc = 0.0;
        for (int64_t j = (int64_t)rowptr[row] - base; j < (int64_t)rowptr[row + 1] - base; ++j)
            acc += fabs(nzval[j]) * fabs(x[(int64_t)colval[j] - base]);
        y[row] = acc;
    }
}

/* ------------------------------------------------------------------------------------------
 * A * B, B dense with k columns  (src/sparse.jl:2391-2413): column loop `columns[k] = A * B[:,k]`
 * => each output column is exactly one _spmv_kernel! pass.  B and C carry explicit row/col
 * strides so both the reference's column-major Matrix (src/dense.jl:63) and a row-major device
 * layout are checkable: element (i,c) at ptr[i*rs + c*cs].
 * ---------------------------------------------------------------------------------------- */
#define ORC_SPMM(NAME, ITYPE, VTYPE)                                                              \
    ORC_API void NAME(const ITYPE *rowptr, const ITYPE *colval, const VTYPE *nzval,               \
                      const VTYPE *B, int64_t b_rs, int64_t b_cs, VTYPE *C, int64_t c_rs,         \
                      int64_t c_cs, int64_t nrows, int k, int base)                               \
    {                                                                                             \
        for (int c = 0; c < k; ++c) {                                                             \
            _Pragma("omp parallel for schedule(static)") for (int64_t row = 0; row < nrows;       \
                                                              ++row)                              \
            {                                                                                     \
                VTYPE acc = 0;                                                                    \
                for (int64_t j = (int64_t)rowptr[row] - base;                                     \
                     j < (int64_t)rowptr[row + 1] - base; ++j)                                    \
                    acc += nzval[j] * B[((int64_t)colval[j] - base) * b_rs + (int64_t)c * b_cs];  \
                C[row * c_rs + (int64_t)c * c_cs] = acc;                                          \
            }                                                                                     \
        }                                                                                         \
    }
ORC_SPMM(orc_spmm_i32, int32_t, double)
ORC_SPMM(orc_spmm_i64, int64_t, double)
ORC_SPMM(orc_spmm_f32_i32, int32_t, float)
ORC_SPMM(orc_spmm_f32_i64, int64_t, float)

/* ------------------------------------------------------------------------------------------
 * dot (src/vectors.jl:798-812): local dot(x.v, y.v) then comm_allreduce(+).
 * norm (src/vectors.jl:758-780): p=2 -> local nrm2, SQUARED, allreduce(+), sqrt; p=1 -> asum,
 * allreduce(+); p=Inf -> max|.|, allreduce(max); else sum |x|^p, allreduce, ^(1/p).
 * The local BLAS summation order (OpenBLAS ddot/dnrm2) is implementation-defined; the oracle sums
 * sequentially in long double and rounds once, i.e. it returns the correctly-rounded-ish value any
 * order must match to ~n*eps relative.  The cross-rank sum is done by the caller in rank order.
 * ---------------------------------------------------------------------------------------- */
ORC_API double orc_dot_local(const double *x, const double *y, int64_t n)
{
    long double acc = 0.0L;
    for (int64_t i = 0; i < n; ++i) acc += (long double)x[i] * (long double)y[i];
    return (double)acc;
}

ORC_API double orc_norm_local(const double *x, int64_t n, double p)
{
    /* returns the LOCAL quantity that the reference feeds to comm_allreduce:
     * p=2: (local nrm2)^2 ; p=1: asum ; p=inf: max|x| ; else sum |x|^p */
    if (n == 0) return 0.0;
    if (p == 2.0) {
        long double acc = 0.0L;
        for (int64_t i = 0; i < n; ++i) acc += (long double)x[i] * (long double)x[i];
        double nrm = (double)sqrtl(acc); /* local_nrm = norm(v.v) */
        return nrm * nrm;                /* local_sum = local_nrm * local_nrm (vectors.jl:764) */
    } else if (p == 1.0) {
        long double acc = 0.0L;
        for (int64_t i = 0; i < n; ++i) acc += fabsl((long double)x[i]);
        return (double)acc;
    } else if (isinf(p)) {
        double m = 0.0;
        for (int64_t i = 0; i < n; ++i) m = fabs(x[i]) > m ? fabs(x[i]) : m;
        return m;
    } else {
        long double acc = 0.0L;
        for (int64_t i = 0; i < n; ++i) acc += powl(fabsl((long double)x[i]), (long double)p);
        return (double)acc;
    }
}

/* ------------------------------------------------------------------------------------------
 * Vector updates used by the CG harness.  Reference semantics: `u + v`, `a * v`
 * (src/vectors.jl:868-877, 944-947) and fused broadcast `dest .= x .+ alpha .* p`
 * (src/vectors.jl:1203-1226): elementwise, mul then add, separately rounded.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_axpy(double alpha, const double *x, double *y, int64_t n) /* y .= y .+ alpha .* x */
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = y[i] + alpha * x[i];
}

ORC_API void orc_xpay(const double *x, double beta, double *y, int64_t n) /* y .= x .+ beta .* y */
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = x[i] + beta * y[i];
}

/* ------------------------------------------------------------------------------------------
 * Reference-like end-to-end staging cost: what execute_plan! does
 * around the kernel on a GPU backend with neighbours: index-list gather loop
 * (src/vectors.jl:426-428).  Used only by bench.py's cpu_baseline notes.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_gather(const double *x, const int64_t *src, const int64_t *dst, double *gathered,
                        int64_t n)
{
    for (int64_t i = 0; i < n; ++i) gathered[dst[i]] = x[src[i]];
}

/* ------------------------------------------------------------------------------------------
 * Sparse x sparse local product.  The reference gathers the needed
 * rows of B (MatrixPlan, src/sparse.jl:554-978) and multiplies with Julia's SparseArrays
 * (`CT = plan.AT * A_csc`, src/sparse.jl:1011), i.e. Gustavson's algorithm column by column of
 * A_csc = row by row of A: entries k of the row in stored (ascending) order, every entry j of gathered
 * row k: `x[j] = nzA*nzB` on first touch, `x[j] += nzA*nzB` afterwards, where nzA is the value of
 * the LEFT factor plan.AT (gathered B) and nzB the value of A.  Output columns ascending.
 * g_rowptr/g_col: gathered rows with GLOBAL columns in [0,ncols).  Two-call protocol: c_col==NULL
 * counts (fills c_rowptr), second call fills.
 * ---------------------------------------------------------------------------------------- */
ORC_API int64_t orc_spgemm(const int64_t *a_rowptr, const int64_t *a_col, const double *a_val,
                           int64_t nrows, const int64_t *g_rowptr, const int64_t *g_col,
                           const double *g_val, int64_t ncols, int64_t *c_rowptr, int64_t *c_col,
                           double *c_val)
{
    int64_t *mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncols > 0 ? ncols : 1));
    double *acc = (double *)malloc(sizeof(double) * (size_t)(ncols > 0 ? ncols : 1));
    int64_t *list = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncols > 0 ? ncols : 1));
    for (int64_t j = 0; j < ncols; ++j) mark[j] = -1;
    int64_t nnz = 0;
    for (int64_t i = 0; i < nrows; ++i) {
        int64_t n = 0;
        for (int64_t p = a_rowptr[i]; p < a_rowptr[i + 1]; ++p) {
            const int64_t k = a_col[p];
            const double av = a_val[p];
            for (int64_t q = g_rowptr[k]; q < g_rowptr[k + 1]; ++q) {
                const int64_t j = g_col[q];
                const double prod = g_val[q] * av;
                if (mark[j] != i) { mark[j] = i; acc[j] = prod; list[n++] = j; }
                else acc[j] += prod;
            }
        }
        /* ascending columns (insertion sort: rows are short) */
        for (int64_t a = 1; a < n; ++a) {
            const int64_t v = list[a];
            int64_t b = a - 1;
            while (b >= 0 && list[b] > v) { list[b + 1] = list[b]; --b; }
            list[b + 1] = v;
        }
        if (c_col)
            for (int64_t a = 0; a < n; ++a) { c_col[nnz + a] = list[a]; c_val[nnz + a] = acc[list[a]]; }
        c_rowptr[i] = nnz;
        nnz += n;
    }
    c_rowptr[nrows] = nnz;
    free(mark); free(acc); free(list);
    return nnz;
}
