// vecops.hip -- dot / norm reductions and fused vector updates (the CG building blocks).
//
// Replaces the reference's local BLAS dot/nrm2 + host MPI Allreduce (src/vectors.jl:758-812) and
// its allocating elementwise ops / broadcast (src/vectors.jl:868-903, 944-964, 1203-1226).
// All kernels are pure HBM streams (16-byte loads, grid-stride, <= 4096 blocks).  Reductions are
// two-stage and deterministic: fixed grid, per-block partial -> one block sums the partials in
// index order; the scalar stays on the device so a CG iteration never synchronises the host.
// Bytes per element: dot 16 (8 for x.x), nrm2sq/asum/amax 8, axpy/xpay 24, scale 16, axpby 24.
#include <stdlib.h>

#include "common.h"

namespace hpcla {

constexpr int RT = 256;            // threads per reduction block
constexpr int MAX_PARTIALS = 2048; // upper bound of stage-1 blocks

enum RedOp { RED_DOT = 0, RED_SQ = 1, RED_ABS = 2, RED_MAX = 3, RED_SUM = 4, RED_POW = 5, RED_MAXV = 6, RED_PROD = 7 };
// RED_MAX: max |x| (identity 0);  RED_MAXV: max x (identity -inf; min x = -max(-x) with negate = 1);
// RED_PROD: product (identity 1)

template <int OP>
__device__ __forceinline__ double red_map(double a, double b, double p = 0.0)
{
    if (OP == RED_POW) return pow(fabs(a), p);
    if (OP == RED_MAXV) return p != 0.0 ? -a : a;          // p doubles as the "negate" flag
    if (OP == RED_DOT) return a * b;
    if (OP == RED_SQ) return a * a;
    if (OP == RED_SUM || OP == RED_PROD) return a;
    return fabs(a);
}
template <int OP>
__device__ __forceinline__ double red_comb(double s, double v)
{
    // NaN-propagating like Julia's maximum / norm(., Inf) (src/vectors.jl:769-772): a plain `v > s ? v : s` never selects
    // a NaN v, which would let rows poisoned by an expired halo wait (halo_poison) pass as a finite maximum
    if (OP == RED_MAX || OP == RED_MAXV) return (v > s || v != v) ? v : s;
    if (OP == RED_PROD) return s * v;
    return s + v;
}

template <int OP>
__device__ __forceinline__ double block_reduce(double v)
{
    __shared__ double s_w[RT / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = red_comb<OP>(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        r = s_w[0];
#pragma unroll
        for (int i = 1; i < RT / 64; ++i) r = red_comb<OP>(r, s_w[i]);
    }
    return r;   // valid in thread 0
}

template <int OP>
__global__ __launch_bounds__(RT) void reduce_stage1(const double *__restrict__ x,
                                                    const double *__restrict__ y, int64_t n,
                                                    double *__restrict__ partial, double p = 0.0)
{
    // 16-byte loads on the aligned body, scalar tail
    double acc = OP == RED_MAXV ? -__builtin_huge_val() : (OP == RED_PROD ? 1.0 : 0.0);
    const int64_t n2 = n / 2;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    const double2 *y2 = reinterpret_cast<const double2 *>(OP == RED_DOT ? y : x);
    int64_t i = (int64_t)blockIdx.x * RT + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * RT;
    for (; i < n2; i += stride) {
        const double2 a = x2[i];
        double2 b = a;
        if (OP == RED_DOT) b = y2[i];
        acc = red_comb<OP>(acc, red_map<OP>(a.x, b.x, p));
        acc = red_comb<OP>(acc, red_map<OP>(a.y, b.y, p));
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        acc = red_comb<OP>(acc, red_map<OP>(x[n - 1], OP == RED_DOT ? y[n - 1] : x[n - 1], p));
    const double r = block_reduce<OP>(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <int OP>
__global__ __launch_bounds__(RT) void reduce_stage2(const double *__restrict__ partial, int np,
                                                    double *__restrict__ out)
{
    double acc = OP == RED_MAXV ? -__builtin_huge_val() : (OP == RED_PROD ? 1.0 : 0.0);
    for (int i = threadIdx.x; i < np; i += RT) acc = red_comb<OP>(acc, partial[i]);
    const double r = block_reduce<OP>(acc);
    if (threadIdx.x == 0) out[0] = r;
}

static inline int reduce_grid(int64_t n)
{
    int64_t g = (n / 2 + RT * 4 - 1) / (RT * 4);   // >= 4 double2 per thread
    if (g < 1) g = 1;
    if (g > MAX_PARTIALS) g = MAX_PARTIALS;
    return (int)g;
}

int allreduce_on(hpcla_comm_t *comm, double *buf, int64_t count, int op, void *stream);  // comm.hip

template <int OP>
static int reduce_impl(hpcla_comm_t *comm, const double *x, const double *y, int64_t n,
                       double *out_dev, void *work, void *stream, double p = 0.0)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "reduce: negative size");
    if (!out_dev || !work) return set_error(HPCLA_ERR_INVALID, "reduce: null out/work");
    if (n > 0 && (!x || (OP == RED_DOT && !y)))
        return set_error(HPCLA_ERR_INVALID, "reduce: null input");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (OP == RED_DOT && (reinterpret_cast<uintptr_t>(y) & 15)))
        return set_error(HPCLA_ERR_INVALID, "reduce: inputs must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    double *partial = reinterpret_cast<double *>(work);
    const int g = reduce_grid(n);
    if (g == 1) {
        // short vectors: the single stage-1 workgroup's partial IS the result (one launch, not two)
        reduce_stage1<OP><<<1, RT, 0, s>>>(x, y, n, out_dev, p);
        HPCLA_CHECK_LAUNCH();
    } else {
        reduce_stage1<OP><<<g, RT, 0, s>>>(x, y, n, partial, p);
        HPCLA_CHECK_LAUNCH();
        reduce_stage2<OP == RED_POW ? RED_SUM : OP><<<1, RT, 0, s>>>(partial, g, out_dev);
        HPCLA_CHECK_LAUNCH();
    }
    if (comm) return allreduce_on(comm, out_dev, 1, (OP == RED_MAX || OP == RED_MAXV) ? 1 : (OP == RED_PROD ? 2 : 0), stream);
    return HPCLA_OK;
}

// sum `np` per-workgroup partials (deterministic two-stage tree) into out[0]; `scratch` holds
// MAX_PARTIALS doubles.  Used by the fused SpMV+dot (spmv.hip) and the fused CG update.
int reduce_partials_sum(const double *partial, int64_t np, double *scratch, double *out, void *stream)
{
    if (np < 0) return set_error(HPCLA_ERR_INVALID, "reduce_partials: bad count");
    hipStream_t s = as_stream(stream);
    if (np <= 4 * RT) {
        reduce_stage2<RED_SUM><<<1, RT, 0, s>>>(partial, (int)np, out);
    } else {
        const int g = reduce_grid(np);
        reduce_stage1<RED_SUM><<<g, RT, 0, s>>>(partial, nullptr, np, scratch);
        HPCLA_CHECK_LAUNCH();
        reduce_stage2<RED_SUM><<<1, RT, 0, s>>>(scratch, g, out);
    }
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

// fused CG update (replaces two broadcasts + one norm of src/vectors.jl:1203-1226, 758-765):
//   a = alpha * *num / *den ;  x += a*p ;  r -= a*Ap ;  partial[block] = sum r_new^2
__global__ __launch_bounds__(RT) void cg_update_kernel(double alpha, const double *__restrict__ num,
                                                       const double *__restrict__ den,
                                                       const double *__restrict__ p,
                                                       const double *__restrict__ Ap,
                                                       double *__restrict__ x, double *__restrict__ r,
                                                       int64_t n, double *__restrict__ partial)
{
    double a = alpha;
    if (num) a = a * num[0];
    if (den) a = a / den[0];
    const int64_t n2 = n / 2;
    const double2 *p2 = reinterpret_cast<const double2 *>(p);
    const double2 *q2 = reinterpret_cast<const double2 *>(Ap);
    double2 *x2 = reinterpret_cast<double2 *>(x);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    double acc = 0.0;
    int64_t i = (int64_t)blockIdx.x * RT + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * RT;
    for (; i < n2; i += stride) {
        const double2 pv = p2[i], qv = q2[i];
        double2 xv = x2[i], rv = r2[i];
        xv.x = xv.x + a * pv.x;
        xv.y = xv.y + a * pv.y;
        rv.x = rv.x - a * qv.x;
        rv.y = rv.y - a * qv.y;
        x2[i] = xv;
        r2[i] = rv;
        acc = acc + rv.x * rv.x;
        acc = acc + rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t j = n - 1;
        x[j] = x[j] + a * p[j];
        const double rn = r[j] - a * Ap[j];
        r[j] = rn;
        acc = acc + rn * rn;
    }
    const double s = block_reduce<RED_SUM>(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__device__ __forceinline__ double dev_scalar(double alpha, const double *num, const double *den)
{
    double a = alpha;
    if (num) a = a * num[0];
    if (den) a = a / den[0];
    return a;
}

// The same CG iteration with the x update DEFERRED to the direction update (both read p_k, so p is read
// once instead of twice: SpMV + 64 B/row of vector traffic instead of 72).  Per element the operations and
// their operands are those of cg_update_kernel + update_kernel<1>: same bits.
//   residual:  a = alpha * *num / *den ;  r -= a*Ap ;  partial[block] = sum r_new^2
//   direction: a as above, b = beta * *bnum / *bden ;  x += a*p ;  p = r + b*p
typedef double v2d_nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load2(const double2 *p)
{
    const v2d_nt t = __builtin_nontemporal_load(reinterpret_cast<const v2d_nt *>(p));
    double2 r;
    r.x = t.x; r.y = t.y;
    return r;
}
__device__ __forceinline__ void nt_store2(double2 v, double2 *p)
{
    v2d_nt t;
    t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<v2d_nt *>(p));
}

// Cache policy of the CG update kernels (round 4; HPCLA_CG_NT bit mask, default 7 = all three): a vector that is not touched again
// before ~2-3 GB of other traffic has passed is loaded / stored NON-TEMPORALLY, so that L2 and the 256 MiB Infinity Cache
// keep the vectors the next kernels re-read (p: gathered by the SpMV; r, Ap: read by the next update).
//   bit 0 (1): x in the direction update (load + store) -- measured -4.4 % per CG iteration
//   bit 1 (2): r load in the direction update (next use: the residual update behind the next SpMV)
//   bit 2 (4): Ap load in the residual update (its last use)
// Config 4's slab, ms per iteration, two alternating rounds of processes on one box (profiles/r04_cg_nontemporal_x.log):
// mask 0: 0.5250 / 0.5233, 1: 0.4997 / 0.5041, 3: 0.4951 / 0.4957, 5: 0.5022 / 0.5006, 7: 0.4903 / 0.4899 (-6.5 %).
// Same operations on the same operands: same bits.
template <bool NTQ>
__global__ __launch_bounds__(RT) void cg_residual_kernel(double alpha, const double *__restrict__ num,
                                                         const double *__restrict__ den,
                                                         const double *__restrict__ Ap, double *__restrict__ r,
                                                         int64_t n, double *__restrict__ partial)
{
    const double a = dev_scalar(alpha, num, den);
    const int64_t n2 = n / 2;
    const double2 *q2 = reinterpret_cast<const double2 *>(Ap);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    double acc = 0.0;
    int64_t i = (int64_t)blockIdx.x * RT + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * RT;
    for (; i < n2; i += stride) {
        const double2 qv = NTQ ? nt_load2(q2 + i) : q2[i];
        double2 rv = r2[i];
        rv.x = rv.x - a * qv.x;
        rv.y = rv.y - a * qv.y;
        r2[i] = rv;
        acc = acc + rv.x * rv.x;
        acc = acc + rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t j = n - 1;
        const double rn = r[j] - a * Ap[j];
        r[j] = rn;
        acc = acc + rn * rn;
    }
    const double s = block_reduce<RED_SUM>(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <bool NTX, bool NTR>
__global__ __launch_bounds__(256) void cg_direction_kernel(double alpha, const double *__restrict__ num,
                                                           const double *__restrict__ den, double beta,
                                                           const double *__restrict__ bnum,
                                                           const double *__restrict__ bden,
                                                           const double *__restrict__ r, double *__restrict__ x,
                                                           double *__restrict__ p, int64_t n)
{
    const double a = dev_scalar(alpha, num, den);
    const double b = dev_scalar(beta, bnum, bden);
    const int64_t n2 = n / 2;
    const double2 *r2 = reinterpret_cast<const double2 *>(r);
    double2 *x2 = reinterpret_cast<double2 *>(x);
    double2 *p2 = reinterpret_cast<double2 *>(p);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n2; i += stride) {
        const double2 rv = NTR ? nt_load2(r2 + i) : r2[i];
        const double2 pv0 = p2[i];
        double2 xv = NTX ? nt_load2(x2 + i) : x2[i], pv = pv0;
        xv.x = xv.x + a * pv.x;
        xv.y = xv.y + a * pv.y;
        pv.x = rv.x + b * pv.x;
        pv.y = rv.y + b * pv.y;
        if (NTX) nt_store2(xv, x2 + i); else x2[i] = xv;
        p2[i] = pv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t j = n - 1;
        const double pj = p[j];
        x[j] = x[j] + a * pj;
        p[j] = r[j] + b * pj;
    }
}

// MODE 0: y = y + a*x   MODE 1: y = x + a*y   MODE 2: y = a*x   MODE 3: y = x / a
template <int MODE>
__global__ __launch_bounds__(256) void update_kernel(double alpha, const double *__restrict__ num,
                                                     const double *__restrict__ den,
                                                     const double *__restrict__ x,
                                                     double *__restrict__ y, int64_t n)
{
    const double a = dev_scalar(alpha, num, den);
    const int64_t n2 = n / 2;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    double2 *y2 = reinterpret_cast<double2 *>(y);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n2; i += stride) {
        const double2 xv = x2[i];
        double2 yv;
        if (MODE == 0) { yv = y2[i]; yv.x = yv.x + a * xv.x; yv.y = yv.y + a * xv.y; }
        if (MODE == 1) { yv = y2[i]; yv.x = xv.x + a * yv.x; yv.y = xv.y + a * yv.y; }
        if (MODE == 2) { yv.x = a * xv.x; yv.y = a * xv.y; }
        if (MODE == 3) { yv.x = xv.x / a; yv.y = xv.y / a; }
        y2[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t j = n - 1;
        if (MODE == 0) y[j] = y[j] + a * x[j];
        if (MODE == 1) y[j] = x[j] + a * y[j];
        if (MODE == 2) y[j] = a * x[j];
        if (MODE == 3) y[j] = x[j] / a;
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(double a, const double *__restrict__ x,
                                                    double b, const double *__restrict__ y,
                                                    double *__restrict__ z, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) z[i] = a * x[i] + b * y[i];
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void fill_uniform_kernel(double *__restrict__ v, int64_t start,
                                                           int64_t count, uint64_t seed)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < count; i += stride) {
        const uint64_t z = splitmix64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(start + i + 1));
        v[i] = (double)(z >> 11) * 0x1.0p-53;
    }
}

static inline uint32_t ew_grid(int64_t n_items)
{
    int64_t g = (n_items + 255) / 256;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return (uint32_t)g;
}

template <int MODE>
static int update_impl(double alpha, const double *num, const double *den, const double *x,
                       double *y, int64_t n, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "update: negative size");
    if (n == 0) return HPCLA_OK;
    if (!x || !y) return set_error(HPCLA_ERR_INVALID, "update: null pointer");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
        return set_error(HPCLA_ERR_INVALID, "update: inputs must be 16-byte aligned");
    update_kernel<MODE><<<ew_grid(n / 2), 256, 0, as_stream(stream)>>>(alpha, num, den, x, y, n);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int64_t hpcla_reduce_work_bytes(void) { return (int64_t)MAX_PARTIALS * sizeof(double); }

HPCLA_API int hpcla_dot_f64(hpcla_comm_t *comm, const double *x, const double *y, int64_t n,
                            double *out_dev, void *work, void *stream)
{
    return reduce_impl<RED_DOT>(comm, x, y, n, out_dev, work, stream);
}
HPCLA_API int hpcla_nrm2sq_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev,
                               void *work, void *stream)
{
    return reduce_impl<RED_SQ>(comm, x, nullptr, n, out_dev, work, stream);
}
HPCLA_API int hpcla_asum_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev,
                             void *work, void *stream)
{
    return reduce_impl<RED_ABS>(comm, x, nullptr, n, out_dev, work, stream);
}
HPCLA_API int hpcla_amax_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev,
                             void *work, void *stream)
{
    return reduce_impl<RED_MAX>(comm, x, nullptr, n, out_dev, work, stream);
}

// maximum(v) / minimum(v) (src/vectors.jl:815-836).  minimum = -max(-x): the kernel negates on load and the
// caller negates the all-reduced result (out holds max(-x) when negate == 1).
HPCLA_API int hpcla_maxval_f64(hpcla_comm_t *comm, const double *x, int64_t n, int negate, double *out_dev,
                               void *work, void *stream)
{
    return reduce_impl<RED_MAXV>(comm, x, nullptr, n, out_dev, work, stream, negate ? 1.0 : 0.0);
}

HPCLA_API int hpcla_sum_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev, void *work,
                            void *stream)
{
    return reduce_impl<RED_SUM>(comm, x, nullptr, n, out_dev, work, stream);
}

// prod(v) (src/vectors.jl:853-858; an empty local part contributes one(T))
HPCLA_API int hpcla_prod_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev, void *work,
                             void *stream)
{
    return reduce_impl<RED_PROD>(comm, x, nullptr, n, out_dev, work, stream);
}

HPCLA_API int hpcla_powsum_f64(hpcla_comm_t *comm, const double *x, int64_t n, double p, double *out_dev,
                               void *work, void *stream)
{
    if (!(p > 0.0)) return set_error(HPCLA_ERR_INVALID, "powsum: p must be positive");
    return reduce_impl<RED_POW>(comm, x, nullptr, n, out_dev, work, stream, p);
}

HPCLA_API int hpcla_axpy_f64(double alpha_host, const double *num_dev, const double *den_dev,
                             const double *x, double *y, int64_t n, void *stream)
{
    return update_impl<0>(alpha_host, num_dev, den_dev, x, y, n, stream);
}
HPCLA_API int hpcla_xpay_f64(const double *x, double alpha_host, const double *num_dev,
                             const double *den_dev, double *y, int64_t n, void *stream)
{
    return update_impl<1>(alpha_host, num_dev, den_dev, x, y, n, stream);
}
HPCLA_API int hpcla_scale_f64(double alpha_host, const double *x, double *y, int64_t n,
                              void *stream)
{
    return update_impl<2>(alpha_host, nullptr, nullptr, x, y, n, stream);
}
HPCLA_API int hpcla_divide_f64(const double *x, double a_host, double *y, int64_t n, void *stream)
{
    return update_impl<3>(a_host, nullptr, nullptr, x, y, n, stream);
}
HPCLA_API int hpcla_axpby_f64(double a, const double *x, double b, const double *y, double *z,
                              int64_t n, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "axpby: negative size");
    if (n == 0) return HPCLA_OK;
    if (!x || !y || !z) return set_error(HPCLA_ERR_INVALID, "axpby: null pointer");
    axpby_kernel<<<ew_grid(n), 256, 0, as_stream(stream)>>>(a, x, b, y, z, n);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_cg_update_f64(hpcla_comm_t *comm, double alpha_host, const double *num_dev,
                                  const double *den_dev, const double *p, const double *Ap, double *x,
                                  double *r, int64_t n, double *rr_out_dev, void *work, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "cg_update: negative size");
    if (!rr_out_dev || !work) return set_error(HPCLA_ERR_INVALID, "cg_update: null out/work");
    if (n > 0 && (!p || !Ap || !x || !r)) return set_error(HPCLA_ERR_INVALID, "cg_update: null vector");
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(Ap) | reinterpret_cast<uintptr_t>(x) |
         reinterpret_cast<uintptr_t>(r)) & 15)
        return set_error(HPCLA_ERR_INVALID, "cg_update: vectors must be 16-byte aligned");
    double *partial = reinterpret_cast<double *>(work);
    const int g = reduce_grid(n);
    cg_update_kernel<<<g, RT, 0, as_stream(stream)>>>(alpha_host, num_dev, den_dev, p, Ap, x, r, n, partial);
    HPCLA_CHECK_LAUNCH();
    reduce_stage2<RED_SUM><<<1, RT, 0, as_stream(stream)>>>(partial, g, rr_out_dev);
    HPCLA_CHECK_LAUNCH();
    if (comm) return allreduce_on(comm, rr_out_dev, 1, 0, stream);
    return HPCLA_OK;
}

static int cg_nt_mask()
{
    static const int m = [] {
        const char *e = getenv("HPCLA_CG_NT");
        return e ? atoi(e) : 7;
    }();
    return m;
}

HPCLA_API int hpcla_cg_residual_f64(hpcla_comm_t *comm, double alpha_host, const double *num_dev,
                                    const double *den_dev, const double *Ap, double *r, int64_t n,
                                    double *rr_out_dev, void *work, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "cg_residual: negative size");
    if (!rr_out_dev || !work) return set_error(HPCLA_ERR_INVALID, "cg_residual: null out/work");
    if (n > 0 && (!Ap || !r)) return set_error(HPCLA_ERR_INVALID, "cg_residual: null vector");
    if ((reinterpret_cast<uintptr_t>(Ap) | reinterpret_cast<uintptr_t>(r)) & 15)
        return set_error(HPCLA_ERR_INVALID, "cg_residual: vectors must be 16-byte aligned");
    double *partial = reinterpret_cast<double *>(work);
    const int g = reduce_grid(n);
    if (cg_nt_mask() & 4)
        cg_residual_kernel<true><<<g, RT, 0, as_stream(stream)>>>(alpha_host, num_dev, den_dev, Ap, r, n, partial);
    else
        cg_residual_kernel<false><<<g, RT, 0, as_stream(stream)>>>(alpha_host, num_dev, den_dev, Ap, r, n, partial);
    HPCLA_CHECK_LAUNCH();
    reduce_stage2<RED_SUM><<<1, RT, 0, as_stream(stream)>>>(partial, g, rr_out_dev);
    HPCLA_CHECK_LAUNCH();
    if (comm) return allreduce_on(comm, rr_out_dev, 1, 0, stream);
    return HPCLA_OK;
}

HPCLA_API int hpcla_cg_direction_f64(double alpha_host, const double *a_num_dev, const double *a_den_dev,
                                     double beta_host, const double *b_num_dev, const double *b_den_dev,
                                     const double *r, double *x, double *p, int64_t n, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "cg_direction: negative size");
    if (n == 0) return HPCLA_OK;
    if (!r || !x || !p) return set_error(HPCLA_ERR_INVALID, "cg_direction: null vector");
    if ((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(p)) & 15)
        return set_error(HPCLA_ERR_INVALID, "cg_direction: vectors must be 16-byte aligned");
    const int m = cg_nt_mask();
#define HPCLA_CG_DIR(NX, NR)                                                                                           \
    cg_direction_kernel<NX, NR><<<ew_grid(n / 2), 256, 0, as_stream(stream)>>>(alpha_host, a_num_dev, a_den_dev, beta_host, \
                                                                               b_num_dev, b_den_dev, r, x, p, n)
    if ((m & 1) && (m & 2)) HPCLA_CG_DIR(true, true);
    else if (m & 1) HPCLA_CG_DIR(true, false);
    else if (m & 2) HPCLA_CG_DIR(false, true);
    else HPCLA_CG_DIR(false, false);
#undef HPCLA_CG_DIR
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

// merge-combine: the five addition kernels of the reference (_copy_a_only_/_copy_b_only_/
// _negate_b_only_/_add_both_/_sub_both_kernel!, src/sparse.jl:1258-1303) as ONE pass over the result:
// entry i of the merged pattern takes a[ia[i]] (ia[i] >= 0) and/or b[ib[i]] (ib[i] >= 0); an entry that
// exists on one side only is COPIED (or negated), never added to zero, exactly like the reference's
// copy kernels (so -0.0 survives).  The result is written contiguously and both source index lists
// are ascending, so all five streams are coalesced; 8 (out) + 2*sizeof(I) (lists) + <= 16 (values)
// bytes per result entry instead of three index-mapped scatter passes.
template <typename I>
__global__ __launch_bounds__(256) void merge_combine_kernel(double *__restrict__ out,
                                                            const double *__restrict__ a,
                                                            const I *__restrict__ ia,
                                                            const double *__restrict__ b,
                                                            const I *__restrict__ ib, int64_t n,
                                                            int subtract)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const I ja = ia[i], jb = ib[i];
        double r = 0.0;
        if (ja >= 0 && jb >= 0) {
            const double av = a[ja], bv = b[jb];
            r = subtract ? av - bv : av + bv;
        } else if (ja >= 0) {
            r = a[ja];
        } else if (jb >= 0) {
            const double bv = b[jb];
            r = subtract ? -bv : bv;
        }
        out[i] = r;
    }
}

template <typename I>
static int merge_combine_launch(double *out, const double *a, const I *ia, const double *b, const I *ib,
                                int64_t n, int subtract, void *stream)
{
    if (n < 0 || (subtract != 0 && subtract != 1))
        return set_error(HPCLA_ERR_INVALID, "merge_combine: bad size/mode");
    if (n == 0) return HPCLA_OK;
    if (!out || !a || !b || !ia || !ib) return set_error(HPCLA_ERR_INVALID, "merge_combine: null pointer");
    merge_combine_kernel<I><<<ew_grid(n), 256, 0, as_stream(stream)>>>(out, a, ia, b, ib, n, subtract);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_merge_combine_f64_i32(double *out, const double *a, const int32_t *ia, const double *b,
                                          const int32_t *ib, int64_t n, int subtract, void *stream)
{
    return merge_combine_launch<int32_t>(out, a, ia, b, ib, n, subtract, stream);
}

HPCLA_API int hpcla_merge_combine_f64_i64(double *out, const double *a, const int64_t *ia, const double *b,
                                          const int64_t *ib, int64_t n, int subtract, void *stream)
{
    return merge_combine_launch<int64_t>(out, a, ia, b, ib, n, subtract, stream);
}

HPCLA_API int hpcla_fill_uniform_f64(double *v, int64_t start, int64_t count, uint64_t seed,
                                     void *stream)
{
    if (count < 0) return set_error(HPCLA_ERR_INVALID, "fill: negative size");
    if (count == 0) return HPCLA_OK;
    if (!v) return set_error(HPCLA_ERR_INVALID, "fill: null pointer");
    fill_uniform_kernel<<<ew_grid(count), 256, 0, as_stream(stream)>>>(v, start, count, seed);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}
