// f32.hip -- the Float32 element type on the hot path (gfx950).
//
// The reference is generic in its element type T: its GPU test configurations are CUDA x {Float32, Float64} and
// Metal x Float32 (test/test_utils.jl:62-80), every one of them through the same `_spmv_kernel!`
// (src/sparse.jl:2055-2066: `acc = zero(T)`, `acc += nzval[j] * x[colval[j]]` in T), the same column loop for
// A * HPCMatrix (src/sparse.jl:2391-2413) and the same local BLAS dot / nrm2 + scalar all-reduce
// (src/vectors.jl:758-812).  Float64 is the graded type (spmv.hip, spmm.hip, vecops.hip); this file gives the SAME
// path to T = Float32 so that a Float32 backend does not fall through to the parent's host-staged path:
//
//   * rowgather_kernel<float, ...> (rowgather_t.h): the row-gather design of spmv.hip (a wave streams its 64 rows' entries into its own slice of
//     LDS with aligned 16-byte loads -- 8 B per entry here -- and every lane then walks ITS row in stored order, separate
//     multiply and add in float: the reference's bits).  One template serves A*x (KC = 1) and A*B on COLUMN-major
//     operands (Julia's Matrix: 8 or 16 columns per workgroup, grid.y column groups); rowmajor_f32_kernel is the same
//     stream for the library's ROW-major device layout (KL lanes per row, one column each).
//   * ghosts: the halo transports of comm.hip / window.hip move 8-byte words.  A Float32 exchange WIDENS the values it
//     sends (float -> double is exact) into a staging vector at the plan's send positions and posts the ordinary
//     exchange from there; the split kernels read their ghosts as doubles and narrow them back (exact).  Every transport
//     -- RCCL serial / overlap, peer-window push, probe, sticky status -- is the Float64 one, unchanged; the halo carries
//     8 instead of 4 bytes per ghost, which is noise next to the rows it overlaps with (64 KiB - 2 MiB per neighbour on
//     the stencil slabs).
//   * reductions: per-element products / squares are formed and summed in DOUBLE (a float product is exact in double),
//     two deterministic stages like vecops.hip, scalar all-reduce in double; the caller rounds to Float32 once.  That is
//     at least as accurate as any Float32 BLAS order; the reference's tolerance for Float32 is 1e-4
//     (test/test_utils.jl:156).
//   * updates: z = a*x + b*y, y = a*x in float with separate multiply and add (scalars rounded to Float32 first).
#include <string.h>

#include "comm_internal.h"
#include "common.h"
#include "rowgather_t.h"

namespace hpcla {

int allreduce_on(hpcla_comm_t *comm, double *buf, int64_t count, int op, void *stream);   // comm.hip

using F32Operand = DenseOperand<float>;

// four consecutive columns of one row of a ROW-major operand (16-byte aligned: the callers check)
template <bool SPLIT>
__device__ __forceinline__ fvec<float, 4> f32_gather4(const F32Operand &b, int64_t col, int64_t coff)
{
    if (SPLIT && col >= b.n_own) {
        const double *p = b.ghost + (col - b.n_own) * b.ghost_rs + coff;
        const fvec<double, 2> lo = *reinterpret_cast<const fvec<double, 2> *>(p);
        const fvec<double, 2> hi = *reinterpret_cast<const fvec<double, 2> *>(p + 2);
        fvec<float, 4> r;
        r.x = (float)lo.x; r.y = (float)lo.y; r.z = (float)hi.x; r.w = (float)hi.y;
        return r;
    }
    return *reinterpret_cast<const fvec<float, 4> *>(b.own + col * b.own_rs + coff);
}

// Row-major operands (the library's device layout of a dense block: element (j, c) at p[j * ld + c], c fastest): KL lanes own
// one row -- lane q of the group its column c0 + q -- so a gather instruction reads 64 / KL whole B rows (KL * 4 contiguous
// bytes each) and the result rows leave as contiguous stores; with lanes = rows (the kernel above) every lane would touch
// its own 32-byte piece of a different line (measured: 0.12 of peak on the 5-point matrix x 16, 1.55 ms against 0.39 ms
// column-major).  The wave still owns 64 rows and stages their entries in LDS once; it walks them 64 / KL rows at a time,
// the KL lanes of a row reading the row's (column, value) pairs from LDS as broadcasts.  Each (row, column) sum is one lane's
// sequential sum in stored order: the reference's bits.  V = 4 (k a multiple of 4, 16-byte aligned operands): a lane owns four
// consecutive columns and moves them as one 16-byte word -- k = 16 is 4 lanes per row, 16 rows per gather instruction, a
// quarter of the steps (5-point matrix x 16: 0.595 ms with one column per lane, 0.285 ms = 0.63 of peak with four,
// profiles/r04_float32.log).
template <typename I, bool SPLIT, int KL, int V>
__global__ __launch_bounds__(F_RPB) void rowmajor_f32_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const float *__restrict__ nzval, F32Operand b,
    float *__restrict__ C, int64_t ldc, int64_t nrows, int64_t nnz, int base, int k,
    const int32_t *__restrict__ block_list, int vec_ok)
{
    constexpr int UR = V == 4 ? 8 : 4, RPS = 64 / KL;                    // rows per step of a wave
    __shared__ __attribute__((aligned(16))) I s_col_all[(F_RPB / 64) * F_CHW];
    __shared__ __attribute__((aligned(16))) float s_val_all[(F_RPB / 64) * F_CHW];

    const int tid = threadIdx.x;
    const int64_t blk = block_list ? (int64_t)block_list[blockIdx.x] : (int64_t)blockIdx.x;
    const int c0 = (int)blockIdx.y * KL * V;
    const int kc = k - c0 < KL * V ? k - c0 : KL * V;
    const int64_t r0 = blk * F_RPB;
    const int nr = (int)((nrows - r0) < F_RPB ? (nrows - r0) : F_RPB);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    I *s_col = s_col_all + wave * F_CHW;
    float *s_val = s_val_all + wave * F_CHW;
    const int64_t rw = r0 + wave * 64;
    const int nrw = nr - wave * 64 < 0 ? 0 : (nr - wave * 64 > 64 ? 64 : nr - wave * 64);
    if (nrw <= 0) return;
    const int q = (lane % KL) * V, sub = lane / KL;                     // q: this lane's first column within the group
    const int64_t p0 = (int64_t)rowptr[rw] - base;
    const int64_t p1 = (int64_t)rowptr[rw + nrw] - base;
    const int64_t pa = vec_ok ? (p0 & ~(int64_t)3) : p0;
    const int64_t total = p1 - pa;
    const int ll = lane < nrw ? lane : nrw - 1;
    const int64_t rlo_own = (int64_t)rowptr[rw + ll] - base - pa, rhi_own = (int64_t)rowptr[rw + ll + 1] - base - pa;
    if (total == 0) {                                                    // a wave of empty rows: no pass would write them
        for (int g = 0; g < KL; ++g) {
            const int row = g * RPS + sub;
            if (row < nrw && q < kc) {
#pragma unroll
                for (int t = 0; t < V; ++t) C[(rw + row) * ldc + c0 + q + t] = 0.0f;
            }
        }
        return;
    }
    for (int64_t c = 0; c < total; c += F_CHW) {
        const int n = (int)((total - c) < F_CHW ? (total - c) : F_CHW);
        stage_pass<float, I>(colval, nzval, s_col, s_val, pa + c, n, nnz, base, lane, vec_ok);
        for (int g = 0; g < KL; ++g) {                                   // step g: rows g * RPS + sub
            const int row = g * RPS + sub;                               // < 64
            const int64_t lo64 = __shfl(rlo_own, row, 64) - c, hi64 = __shfl(rhi_own, row, 64) - c;
            const bool live = row < nrw && q < kc;
            int j = live ? (lo64 > 0 ? (int)lo64 : 0) : 0;
            const int e = live ? (hi64 < n ? (int)hi64 : n) : 0;
            // the running sum of a row that began in an earlier pass is carried in C itself (a float round trip is exact):
            // only waves with more than one pass -- more than 464 entries in 64 rows -- pay for it
            float *cp = C + (rw + row) * ldc + c0 + q;
            const bool has = j < e;                                       // the row has entries in this pass
            float acc[V];
#pragma unroll
            for (int t = 0; t < V; ++t) acc[t] = 0.0f;
            if (has && c > 0 && lo64 < 0) {
                if (V == 4) {
                    const fvec<float, 4> cv = *reinterpret_cast<const fvec<float, 4> *>(cp);
                    acc[0] = cv.x; acc[1 % V] = cv.y; acc[2 % V] = cv.z; acc[3 % V] = cv.w;
                } else {
                    acc[0] = *cp;
                }
            }
            for (; j < e; j += UR) {
                int64_t cc[UR];
                float vv[UR], xx[UR][V];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    cc[u] = 0; vv[u] = 0.0f;
                    if (j + u < e) { cc[u] = (int64_t)(I)(s_col[j + u] - (I)base); vv[u] = s_val[j + u]; }
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
#pragma unroll
                    for (int t = 0; t < V; ++t) xx[u][t] = 0.0f;
                    if (j + u < e) {
                        if (V == 4) {
                            const fvec<float, 4> g4 = f32_gather4<SPLIT>(b, cc[u], c0 + q);
                            xx[u][0] = g4.x; xx[u][1 % V] = g4.y; xx[u][2 % V] = g4.z; xx[u][3 % V] = g4.w;
                        } else {
                            xx[u][0] = operand_gather<float, SPLIT>(b, cc[u], c0 + q, c0 + q);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UR; ++u)
                    if (j + u < e) {
#pragma unroll
                        for (int t = 0; t < V; ++t) acc[t] += vv[u] * xx[u][t];
                    }
            }
            // the first pass gives every row its initial value (0 for rows that start later or are empty), later passes
            // write the rows they touch: a row's last touching pass leaves its final sum
            if (live && (c == 0 || has)) {
                if (V == 4) {
                    fvec<float, 4> cv;
                    cv.x = acc[0]; cv.y = acc[1 % V]; cv.z = acc[2 % V]; cv.w = acc[3 % V];
                    *reinterpret_cast<fvec<float, 4> *>(cp) = cv;
                } else {
                    *cp = acc[0];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename I>
static int f32_launch(const I *rowptr, const I *colval, const float *nzval, const F32Operand &b, float *C, int64_t c_rs,
                      int64_t c_cs, int64_t nrows, int64_t nnz, int k, int index_base, const int32_t *block_list,
                      int64_t n_blocks, void *stream, const char *who)
{
    if (nrows < 0 || nnz < 0 || k < 0) return set_error(HPCLA_ERR_INVALID, "%s: negative size", who);
    if (index_base != 0 && index_base != 1) return set_error(HPCLA_ERR_INVALID, "%s: index_base must be 0 or 1", who);
    if (nrows == 0 || k == 0) return HPCLA_OK;
    if (!rowptr || !C) return set_error(HPCLA_ERR_INVALID, "%s: null rowptr / result", who);
    if (nnz > 0 && (!colval || !nzval || !b.own)) return set_error(HPCLA_ERR_INVALID, "%s: null colval / nzval / operand with nnz > 0", who);
    const int64_t all_blocks = (nrows + F_RPB - 1) / F_RPB;
    int64_t launch_blocks = all_blocks;
    if (block_list) {
        if (n_blocks < 0 || n_blocks > all_blocks) return set_error(HPCLA_ERR_INVALID, "%s: n_blocks out of range", who);
        launch_blocks = n_blocks;
    }
    if (launch_blocks == 0) return HPCLA_OK;
    HPCLA_CHECK_GRID(launch_blocks, who);
    const int vec_ok = stage_vec_ok<float, I>(colval, nzval);
    const bool split = b.ghost != nullptr;
    hipStream_t s = as_stream(stream);
    if (k == 1 && b.own_rs == 1 && c_rs == 1 && (!split || b.ghost_rs == 1)) {       // one contiguous column in, one out: A*x
        dim3 grid((uint32_t)launch_blocks), block(F_RPB);
        if (split)
            rowgather_kernel<float, I, true, 1><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, 1, 0, nrows, nnz, index_base, 1,
                                                                   block_list, vec_ok, 1);
        else
            rowgather_kernel<float, I, false, 1><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, 1, 0, nrows, nnz, index_base, 1,
                                                                    block_list, vec_ok, 1);
    } else if (b.own_cs == 1 && c_cs == 1 && (!split || b.ghost_cs == 1)) {
        // row-major operands.  k a multiple of 4 and everything 16-byte aligned: a lane owns 4 columns (V = 4), KL = the power
        // of two >= k / 4 lanes per row (at most 64 columns per workgroup); else one column per lane, KL = the power of two >= k
        const bool v4 = k % 4 == 0 && b.own_rs % 4 == 0 && c_rs % 4 == 0 && reinterpret_cast<uintptr_t>(b.own) % 16 == 0 &&
                        reinterpret_cast<uintptr_t>(C) % 16 == 0 &&
                        (!split || (b.ghost_rs % 2 == 0 && reinterpret_cast<uintptr_t>(b.ghost) % 16 == 0));
        const int kq = v4 ? k / 4 : k;
        const int kl = v4 ? (kq <= 1 ? 1 : (kq <= 2 ? 2 : (kq <= 4 ? 4 : (kq <= 8 ? 8 : 16))))
                          : (kq <= 4 ? 4 : (kq <= 8 ? 8 : (kq <= 16 ? 16 : (kq <= 32 ? 32 : 64))));
        const int groups = (kq + kl - 1) / kl;
        if (groups > 65535) return set_error(HPCLA_ERR_UNSUPPORTED, "%s: more than %d columns", who, 65535 * 64);
        dim3 grid((uint32_t)launch_blocks, (uint32_t)groups), block(F_RPB);
#define HPCLA_F32_ROWMAJOR(KL, V)                                                                                           \
    do {                                                                                                                    \
        if (split)                                                                                                          \
            rowmajor_f32_kernel<I, true, KL, V><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, c_rs, nrows, nnz, index_base, \
                                                                      k, block_list, vec_ok);                               \
        else                                                                                                                \
            rowmajor_f32_kernel<I, false, KL, V><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, c_rs, nrows, nnz,      \
                                                                       index_base, k, block_list, vec_ok);                  \
    } while (0)
        if (v4) {
            switch (kl) {
            case 1: HPCLA_F32_ROWMAJOR(1, 4); break;
            case 2: HPCLA_F32_ROWMAJOR(2, 4); break;
            case 4: HPCLA_F32_ROWMAJOR(4, 4); break;
            case 8: HPCLA_F32_ROWMAJOR(8, 4); break;
            default: HPCLA_F32_ROWMAJOR(16, 4); break;
            }
        } else {
            switch (kl) {
            case 4: HPCLA_F32_ROWMAJOR(4, 1); break;
            case 8: HPCLA_F32_ROWMAJOR(8, 1); break;
            case 16: HPCLA_F32_ROWMAJOR(16, 1); break;
            case 32: HPCLA_F32_ROWMAJOR(32, 1); break;
            default: HPCLA_F32_ROWMAJOR(64, 1); break;
            }
        }
#undef HPCLA_F32_ROWMAJOR
    } else if (k <= 8) {
        dim3 grid((uint32_t)launch_blocks, 1), block(F_RPB);
        if (split)
            rowgather_kernel<float, I, true, 8><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, c_rs, c_cs, nrows, nnz,
                                                                   index_base, k, block_list, vec_ok, c_rs == 1);
        else
            rowgather_kernel<float, I, false, 8><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, c_rs, c_cs, nrows, nnz,
                                                                    index_base, k, block_list, vec_ok, c_rs == 1);
    } else {
        // lanes = rows (column-major operands: 64 consecutive rows of one column are one contiguous run), 16 columns per
        // workgroup: A is streamed once per 16 columns
        constexpr int KC = 16;
        const int groups = (k + KC - 1) / KC;
        if (groups > 65535) return set_error(HPCLA_ERR_UNSUPPORTED, "%s: more than %d columns", who, 65535 * KC);
        dim3 grid((uint32_t)launch_blocks, (uint32_t)groups), block(F_RPB);
        if (split)
            rowgather_kernel<float, I, true, KC><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, c_rs, c_cs, nrows, nnz,
                                                                    index_base, k, block_list, vec_ok, c_rs == 1);
        else
            rowgather_kernel<float, I, false, KC><<<grid, block, 0, s>>>(rowptr, colval, nzval, b, C, c_rs, c_cs, nrows, nnz,
                                                                     index_base, k, block_list, vec_ok, c_rs == 1);
    }
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

static int layout_strides(int layout, int64_t ld, int64_t *rs, int64_t *cs, const char *who)
{
    if (layout == HPCLA_LAYOUT_ROW) { *rs = ld; *cs = 1; return HPCLA_OK; }
    if (layout == HPCLA_LAYOUT_COL) { *rs = 1; *cs = ld; return HPCLA_OK; }
    return set_error(HPCLA_ERR_INVALID, "%s: layout must be HPCLA_LAYOUT_ROW or HPCLA_LAYOUT_COL", who);
}

// ---- halo: widen the values an exchange sends (colmajor.hip: stage_at_kernel, any operand strides) -----------------------------
template <typename T>
int halo_begin_strided(hpcla_halo_plan_t *plan, const T *x, int64_t x_rs, int64_t x_cs, double *stage, void *stream, const char *who);

// ---- reductions ------------------------------------------------------------------------------------------------------
constexpr int F_RT = 256;
constexpr int F_MAX_PARTIALS = 1024;
enum F32Red { F_DOT = 0, F_SQ = 1, F_ABS = 2, F_AMAX = 3, F_SUM = 4, F_MAXV = 5 };

template <int OP>
__device__ __forceinline__ double f_map(float a, float b, int negate)
{
    if (OP == F_DOT) return (double)a * (double)b;
    if (OP == F_SQ) return (double)a * (double)a;
    if (OP == F_SUM) return (double)a;
    if (OP == F_MAXV) return negate ? -(double)a : (double)a;
    return fabs((double)a);
}
template <int OP>
__device__ __forceinline__ double f_comb(double s, double v)
{
    if (OP == F_AMAX || OP == F_MAXV) return (v > s || v != v) ? v : s;      // NaN-propagating (see vecops.hip red_comb)
    return s + v;
}
template <int OP>
__device__ __forceinline__ double f_block_reduce(double v)
{
    __shared__ double s_w[F_RT / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = f_comb<OP>(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        r = s_w[0];
#pragma unroll
        for (int i = 1; i < F_RT / 64; ++i) r = f_comb<OP>(r, s_w[i]);
    }
    return r;
}

template <int OP>
__global__ __launch_bounds__(F_RT) void f32_reduce_stage1(const float *__restrict__ x, const float *__restrict__ y, int64_t n,
                                                          double *__restrict__ partial, int negate)
{
    double acc = OP == F_MAXV ? -__builtin_huge_val() : 0.0;
    const int64_t n4 = n / 4;
    const fvec<float, 4> *x4 = reinterpret_cast<const fvec<float, 4> *>(x);
    const fvec<float, 4> *y4 = reinterpret_cast<const fvec<float, 4> *>(OP == F_DOT ? y : x);
    const int64_t stride = (int64_t)gridDim.x * F_RT;
    for (int64_t i = (int64_t)blockIdx.x * F_RT + threadIdx.x; i < n4; i += stride) {
        const fvec<float, 4> a = x4[i];
        fvec<float, 4> bq = a;
        if (OP == F_DOT) bq = y4[i];
        acc = f_comb<OP>(acc, f_map<OP>(a.x, bq.x, negate));
        acc = f_comb<OP>(acc, f_map<OP>(a.y, bq.y, negate));
        acc = f_comb<OP>(acc, f_map<OP>(a.z, bq.z, negate));
        acc = f_comb<OP>(acc, f_map<OP>(a.w, bq.w, negate));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t j = n4 * 4; j < n; ++j) acc = f_comb<OP>(acc, f_map<OP>(x[j], OP == F_DOT ? y[j] : x[j], negate));
    const double r = f_block_reduce<OP>(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <int OP>
__global__ __launch_bounds__(F_RT) void f32_reduce_stage2(const double *__restrict__ partial, int np, double *__restrict__ out)
{
    double acc = OP == F_MAXV ? -__builtin_huge_val() : 0.0;
    for (int i = threadIdx.x; i < np; i += F_RT) acc = f_comb<OP>(acc, partial[i]);
    const double r = f_block_reduce<OP>(acc);
    if (threadIdx.x == 0) out[0] = r;
}

template <int OP>
static int f32_reduce(hpcla_comm_t *comm, const float *x, const float *y, int64_t n, double *out_dev, void *work,
                      void *stream, int negate = 0)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "reduce_f32: negative size");
    if (!out_dev || !work) return set_error(HPCLA_ERR_INVALID, "reduce_f32: null out / work");
    if (n > 0 && (!x || (OP == F_DOT && !y))) return set_error(HPCLA_ERR_INVALID, "reduce_f32: null input");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (OP == F_DOT && (reinterpret_cast<uintptr_t>(y) & 15)))
        return set_error(HPCLA_ERR_INVALID, "reduce_f32: inputs must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    int64_t g = (n / 4 + F_RT * 4 - 1) / (F_RT * 4);
    if (g < 1) g = 1;
    if (g > F_MAX_PARTIALS) g = F_MAX_PARTIALS;
    if (g == 1) {
        f32_reduce_stage1<OP><<<1, F_RT, 0, s>>>(x, y, n, out_dev, negate);
        HPCLA_CHECK_LAUNCH();
    } else {
        double *partial = reinterpret_cast<double *>(work);
        f32_reduce_stage1<OP><<<(uint32_t)g, F_RT, 0, s>>>(x, y, n, partial, negate);
        HPCLA_CHECK_LAUNCH();
        f32_reduce_stage2<OP><<<1, F_RT, 0, s>>>(partial, (int)g, out_dev);
        HPCLA_CHECK_LAUNCH();
    }
    if (comm) return allreduce_on(comm, out_dev, 1, (OP == F_AMAX || OP == F_MAXV) ? 1 : 0, stream);
    return HPCLA_OK;
}

// ---- updates ---------------------------------------------------------------------------------------------------------
// MODE 0: z = a*x + b*y   MODE 1: z = a*x   MODE 2: z = x / a
template <int MODE>
__global__ __launch_bounds__(256) void f32_update_kernel(float a, const float *__restrict__ x, float b_,
                                                         const float *__restrict__ y, float *__restrict__ z, int64_t n)
{
    const int64_t n4 = n / 4;
    const fvec<float, 4> *x4 = reinterpret_cast<const fvec<float, 4> *>(x);
    const fvec<float, 4> *y4 = reinterpret_cast<const fvec<float, 4> *>(y);
    fvec<float, 4> *z4 = reinterpret_cast<fvec<float, 4> *>(z);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const fvec<float, 4> xv = x4[i];
        fvec<float, 4> zv;
        if (MODE == 0) {
            const fvec<float, 4> yv = y4[i];
            zv.x = a * xv.x + b_ * yv.x; zv.y = a * xv.y + b_ * yv.y; zv.z = a * xv.z + b_ * yv.z; zv.w = a * xv.w + b_ * yv.w;
        }
        if (MODE == 1) { zv.x = a * xv.x; zv.y = a * xv.y; zv.z = a * xv.z; zv.w = a * xv.w; }
        if (MODE == 2) { zv.x = xv.x / a; zv.y = xv.y / a; zv.z = xv.z / a; zv.w = xv.w / a; }
        z4[i] = zv;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t j = n4 * 4; j < n; ++j) {
            if (MODE == 0) z[j] = a * x[j] + b_ * y[j];
            if (MODE == 1) z[j] = a * x[j];
            if (MODE == 2) z[j] = x[j] / a;
        }
}

template <int MODE>
static int f32_update(float a, const float *x, float b, const float *y, float *z, int64_t n, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "update_f32: negative size");
    if (n == 0) return HPCLA_OK;
    if (!x || !z || (MODE == 0 && !y)) return set_error(HPCLA_ERR_INVALID, "update_f32: null pointer");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(z) & 15) ||
        (MODE == 0 && (reinterpret_cast<uintptr_t>(y) & 15)))
        return set_error(HPCLA_ERR_INVALID, "update_f32: operands must be 16-byte aligned");
    int64_t g = (n / 4 + 255) / 256;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    f32_update_kernel<MODE><<<(uint32_t)g, 256, 0, as_stream(stream)>>>(a, x, b, y, z, n);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

}  // namespace hpcla

using namespace hpcla;

// ---- C ABI -----------------------------------------------------------------------------------------------------------
#define HPCLA_F32_SPMV(SFX, ITYPE)                                                                                          \
    HPCLA_API int hpcla_spmv_csr_f32_##SFX(const ITYPE *rowptr, const ITYPE *colval, const float *nzval, const float *x,   \
                                           float *y, int64_t nrows, int64_t nnz, int index_base, void *stream)             \
    {                                                                                                                       \
        const F32Operand b{x, 1, 0, nullptr, 1, 0, 0};                                                                      \
        return f32_launch<ITYPE>(rowptr, colval, nzval, b, y, 1, 0, nrows, nnz, 1, index_base, nullptr, 0, stream,          \
                                 "spmv_csr_f32");                                                                           \
    }                                                                                                                       \
    HPCLA_API int hpcla_spmv_split_f32_##SFX(const ITYPE *rowptr, const ITYPE *colval_split, const float *nzval,           \
                                             const float *x_own, const double *x_ghost_wide, int64_t n_own, float *y,      \
                                             int64_t nrows, int64_t nnz, int index_base, const int32_t *block_list,        \
                                             int64_t n_blocks, void *stream)                                               \
    {                                                                                                                       \
        if (n_own < 0) return set_error(HPCLA_ERR_INVALID, "spmv_split_f32: negative n_own");                               \
        const F32Operand b{x_own, 1, 0, x_ghost_wide, 1, 0, n_own};                                                         \
        return f32_launch<ITYPE>(rowptr, colval_split, nzval, b, y, 1, 0, nrows, nnz, 1, index_base, block_list, n_blocks, \
                                 stream, "spmv_split_f32");                                                                 \
    }                                                                                                                       \
    HPCLA_API int hpcla_spmm_csr_f32_##SFX(const ITYPE *rowptr, const ITYPE *colval, const float *nzval, const float *B,   \
                                           int64_t ldb, int b_layout, float *C, int64_t ldc, int c_layout, int64_t nrows,  \
                                           int64_t nnz, int k, int index_base, void *stream)                               \
    {                                                                                                                       \
        F32Operand b{B, 0, 0, nullptr, 0, 0, 0};                                                                            \
        int64_t c_rs = 0, c_cs = 0;                                                                                         \
        int rc = layout_strides(b_layout, ldb, &b.own_rs, &b.own_cs, "spmm_csr_f32");                                       \
        if (rc == HPCLA_OK) rc = layout_strides(c_layout, ldc, &c_rs, &c_cs, "spmm_csr_f32");                               \
        if (rc != HPCLA_OK) return rc;                                                                                      \
        if (k > 0 && (ldb < (b_layout == HPCLA_LAYOUT_ROW ? k : 1) || ldc < (c_layout == HPCLA_LAYOUT_ROW ? k : nrows)))    \
            return set_error(HPCLA_ERR_INVALID, "spmm_csr_f32: leading dimension too small");                               \
        return f32_launch<ITYPE>(rowptr, colval, nzval, b, C, c_rs, c_cs, nrows, nnz, k, index_base, nullptr, 0, stream,    \
                                 "spmm_csr_f32");                                                                           \
    }                                                                                                                       \
    HPCLA_API int hpcla_spmm_split_f32_##SFX(const ITYPE *rowptr, const ITYPE *colval_split, const float *nzval,           \
                                             const float *B_own, int64_t ldb_own, const double *B_ghost_wide,              \
                                             int64_t ldb_ghost, int64_t n_own, float *C, int64_t ldc, int64_t nrows,       \
                                             int64_t nnz, int k, int index_base, const int32_t *block_list,                \
                                             int64_t n_blocks, void *stream)                                               \
    {                                                                                                                       \
        if (n_own < 0) return set_error(HPCLA_ERR_INVALID, "spmm_split_f32: negative n_own");                               \
        if (k > 0 && (ldb_own < k || ldc < k || (B_ghost_wide && ldb_ghost < k)))                                           \
            return set_error(HPCLA_ERR_INVALID, "spmm_split_f32: leading dimension too small");                             \
        const F32Operand b{B_own, ldb_own, 1, B_ghost_wide, ldb_ghost, 1, n_own};                                           \
        return f32_launch<ITYPE>(rowptr, colval_split, nzval, b, C, ldc, 1, nrows, nnz, k, index_base, block_list,          \
                                 n_blocks, stream, "spmm_split_f32");                                                       \
    }
HPCLA_F32_SPMV(i32, int32_t)
HPCLA_F32_SPMV(i64, int64_t)

// column-major own block and result (Julia's Matrix), row-major ghost segment: hpcla_spmm_split_colmajor_f64_*'s twin
#define HPCLA_F32_COLMAJOR_SPLIT(SFX, ITYPE)                                                                                \
    HPCLA_API int hpcla_spmm_split_colmajor_f32_##SFX(const ITYPE *rowptr, const ITYPE *colval_split, const float *nzval,   \
                                                      const float *B_own, int64_t ldb_own, const double *B_ghost_wide,      \
                                                      int64_t ldb_ghost, int64_t n_own, float *C, int64_t ldc,              \
                                                      int64_t nrows, int64_t nnz, int k, int index_base,                    \
                                                      const int32_t *block_list, int64_t n_blocks, void *stream)            \
    {                                                                                                                       \
        if (n_own < 0) return set_error(HPCLA_ERR_INVALID, "spmm_split_colmajor_f32: negative n_own");                      \
        if (k > 1 && (ldb_own < n_own || ldc < nrows))                                                                      \
            return set_error(HPCLA_ERR_INVALID, "spmm_split_colmajor_f32: leading dimension smaller than the row count");   \
        if (B_ghost_wide && ldb_ghost < k) return set_error(HPCLA_ERR_INVALID, "spmm_split_colmajor_f32: ldb_ghost < k");   \
        const F32Operand b{B_own, 1, ldb_own, B_ghost_wide, ldb_ghost, 1, n_own};                                           \
        return f32_launch<ITYPE>(rowptr, colval_split, nzval, b, C, 1, ldc, nrows, nnz, k, index_base, block_list,          \
                                 n_blocks, stream, "spmm_split_colmajor_f32");                                              \
    }
HPCLA_F32_COLMAJOR_SPLIT(i32, int32_t)
HPCLA_F32_COLMAJOR_SPLIT(i64, int64_t)

// Distributed y = A*x in ONE call (the Float32 twin of hpcla_spmv_dist_f64_*; not a fused launch: begin -> interior blocks
// -> end -> boundary blocks on the caller's stream, the exchange itself on the plan's side stream / in the peers' windows)
template <typename I>
static int spmv_dist_f32(hpcla_halo_plan_t *plan, const I *rowptr, const I *colval_split, const float *nzval, const float *x,
                         int64_t n_own, float *y, int64_t nrows, int64_t nnz, int index_base, const int32_t *interior,
                         int64_t n_interior, const int32_t *boundary, int64_t n_boundary, double *stage, void *stream)
{
    if (n_own < 0) return set_error(HPCLA_ERR_INVALID, "spmv_dist_f32: negative n_own");
    F32Operand b{x, 1, 0, nullptr, 1, 0, n_own};
    if (!plan || (plan->send_ranks.empty() && plan->recv_ranks.empty()))
        return f32_launch<I>(rowptr, colval_split, nzval, b, y, 1, 0, nrows, nnz, 1, index_base, nullptr, 0, stream, "spmv_dist_f32");
    if (n_interior < 0 || n_boundary < 0 || (n_interior > 0 && !interior) || (n_boundary > 0 && !boundary))
        return set_error(HPCLA_ERR_INVALID, "spmv_dist_f32: bad block lists");
    if (!plan->single_buffer && plan->attached)
        return set_error(HPCLA_ERR_INVALID, "spmv_dist_f32: the plan must be single-buffered (hpcla_halo_plan_create_ex with "
                                            "HPCLA_HALO_SINGLE_BUFFER): its ghost pointer is taken while the exchange is in flight");
    int rc = hpcla_halo_begin_f32(plan, x, stage, stream);
    if (rc != HPCLA_OK) return rc;
    if (n_interior > 0)                                            // no ghost column in these blocks: they overlap the exchange
        rc = f32_launch<I>(rowptr, colval_split, nzval, b, y, 1, 0, nrows, nnz, 1, index_base, interior, n_interior, stream, "spmv_dist_f32");
    const int rc_end = hpcla_halo_end(plan, stream);              // always: the caller's stream must be joined to the exchange
    if (rc != HPCLA_OK) return rc;
    if (rc_end != HPCLA_OK) return rc_end;
    if (n_boundary > 0) {
        double *ghost = nullptr;
        rc = hpcla_halo_ghost_ptr(plan, &ghost, nullptr);
        if (rc != HPCLA_OK) return rc;
        b.ghost = ghost;
        if (!ghost) return set_error(HPCLA_ERR_INVALID, "spmv_dist_f32: boundary blocks without a ghost segment");
        rc = f32_launch<I>(rowptr, colval_split, nzval, b, y, 1, 0, nrows, nnz, 1, index_base, boundary, n_boundary, stream, "spmv_dist_f32");
    }
    return rc;
}

HPCLA_API int hpcla_spmv_dist_f32_i32(hpcla_halo_plan_t *plan, const int32_t *rowptr, const int32_t *colval_split, const float *nzval,
                                      const float *x, int64_t n_own, float *y, int64_t nrows, int64_t nnz, int index_base,
                                      const int32_t *interior_blocks, int64_t n_interior, const int32_t *boundary_blocks,
                                      int64_t n_boundary, double *stage, void *stream)
{
    return spmv_dist_f32<int32_t>(plan, rowptr, colval_split, nzval, x, n_own, y, nrows, nnz, index_base, interior_blocks, n_interior,
                                  boundary_blocks, n_boundary, stage, stream);
}
HPCLA_API int hpcla_spmv_dist_f32_i64(hpcla_halo_plan_t *plan, const int64_t *rowptr, const int64_t *colval_split, const float *nzval,
                                      const float *x, int64_t n_own, float *y, int64_t nrows, int64_t nnz, int index_base,
                                      const int32_t *interior_blocks, int64_t n_interior, const int32_t *boundary_blocks,
                                      int64_t n_boundary, double *stage, void *stream)
{
    return spmv_dist_f32<int64_t>(plan, rowptr, colval_split, nzval, x, n_own, y, nrows, nnz, index_base, interior_blocks, n_interior,
                                  boundary_blocks, n_boundary, stage, stream);
}

HPCLA_API int hpcla_halo_begin_f32(hpcla_halo_plan_t *plan, const float *x, double *stage, void *stream)
{
    if (!plan) return set_error(HPCLA_ERR_INVALID, "halo_begin_f32: null plan");
    return halo_begin_strided<float>(plan, x, plan->width, 1, stage, stream, "halo_begin_f32");          // row-major, ld = width
}

HPCLA_API int hpcla_halo_begin_strided_f32(hpcla_halo_plan_t *plan, const float *x, int64_t x_rs, int64_t x_cs, double *stage,
                                           void *stream)
{
    return halo_begin_strided<float>(plan, x, x_rs, x_cs, stage, stream, "halo_begin_strided_f32");
}

HPCLA_API int hpcla_dot_f32(hpcla_comm_t *comm, const float *x, const float *y, int64_t n, double *out_dev, void *work,
                            void *stream)
{
    return f32_reduce<F_DOT>(comm, x, y, n, out_dev, work, stream);
}
HPCLA_API int hpcla_nrm2sq_f32(hpcla_comm_t *comm, const float *x, int64_t n, double *out_dev, void *work, void *stream)
{
    return f32_reduce<F_SQ>(comm, x, nullptr, n, out_dev, work, stream);
}
HPCLA_API int hpcla_asum_f32(hpcla_comm_t *comm, const float *x, int64_t n, double *out_dev, void *work, void *stream)
{
    return f32_reduce<F_ABS>(comm, x, nullptr, n, out_dev, work, stream);
}
HPCLA_API int hpcla_amax_f32(hpcla_comm_t *comm, const float *x, int64_t n, double *out_dev, void *work, void *stream)
{
    return f32_reduce<F_AMAX>(comm, x, nullptr, n, out_dev, work, stream);
}
HPCLA_API int hpcla_sum_f32(hpcla_comm_t *comm, const float *x, int64_t n, double *out_dev, void *work, void *stream)
{
    return f32_reduce<F_SUM>(comm, x, nullptr, n, out_dev, work, stream);
}
HPCLA_API int hpcla_maxval_f32(hpcla_comm_t *comm, const float *x, int64_t n, int negate, double *out_dev, void *work,
                               void *stream)
{
    return f32_reduce<F_MAXV>(comm, x, nullptr, n, out_dev, work, stream, negate ? 1 : 0);
}

HPCLA_API int hpcla_axpby_f32(float a, const float *x, float b, const float *y, float *z, int64_t n, void *stream)
{
    return f32_update<0>(a, x, b, y, z, n, stream);
}
HPCLA_API int hpcla_scale_f32(float a, const float *x, float *y, int64_t n, void *stream)
{
    return f32_update<1>(a, x, 0.0f, nullptr, y, n, stream);
}
HPCLA_API int hpcla_divide_f32(const float *x, float a, float *y, int64_t n, void *stream)
{
    return f32_update<2>(a, x, 0.0f, nullptr, y, n, stream);
}
