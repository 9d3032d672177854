// packed.hip -- OPT-IN packed copy of a CSR matrix for the SpMV hot path ("packed CSR").
//
// Plain CSR moves 12 B per stored entry (Int32).  Many matrices on this path -- stencils, graph
// Laplacians, uniform-mesh FEM -- hold only a handful of distinct values and have all columns of a
// 256-row block within +-32 K of the block's first row.  For those, a plan-time packed copy stores
//
//     dcol[j]  = colval_split[j] - 256*(row block of j)      int16   (2 B/entry)
//     code[j]  = index of nzval[j] in a <= 256-entry dictionary  uint8   (1 B/entry)
//
// i.e. 3 B/entry instead of 12; rowptr, x and y are unchanged.  The kernel is the same row-block
// stream as spmv.hip (products parked in LDS, rows summed sequentially in stored order) and the
// product is dict[code] * x[r0 + dcol] -- the SAME two fp64 numbers the CSR kernel multiplies, so the
// result stays bit-identical to the reference loop (src/sparse.jl:2055-2066).
//
// Eligibility is decided by the library (hpcla_packed_create returns HPCLA_ERR_UNSUPPORTED
// otherwise): <= 256 distinct values, and every entry of every packed row block owned (no ghost
// column) and within the int16 window.  Boundary row blocks (ghost columns) keep the CSR kernel.
// This is an optimisation of the BYTES MOVED, reported separately from the CSR headline number.
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include <string.h>

#include "comm_internal.h"

namespace hpcla {

constexpr int P_RPB = 256;             // rows per block (== spmv.hip RPB)
constexpr int P_CHUNK = P_RPB * 8;     // one aligned octet (8 entries) per lane per pass

typedef short v8s __attribute__((ext_vector_type(8)));

int spmv_split_i32(const int32_t *, const int32_t *, const double *, const double *, const double *,
                   int64_t, double *, int64_t, int64_t, int, const int32_t *, int64_t, void *,
                   double *, int64_t);
int reduce_partials_sum(const double *partial, int64_t np, double *scratch, double *out,
                        void *stream);
int allreduce_on(hpcla_comm_t *comm, double *buf, int64_t count, int op, void *stream);
bool halo_active(const hpcla_halo_plan_t *plan);   // comm.hip
bool halo_serial_mode(const hpcla_halo_plan_t *plan);   // comm.hip (HPCLA_HALO_MODE; push counts as serial)
int spmv_fused_i32(const int32_t *, const int32_t *, const double *, const double *, const double *, int64_t,
                   double *, int64_t, int64_t, int, const int32_t *, int64_t, int64_t, const int32_t *, int64_t,
                   const HaloWait &, const PushArgs &, void *, double *);     // spmv.hip
int halo_exchange_inline(hpcla_halo_plan_t *plan, const double *x, void *stream);   // comm.hip

}  // namespace hpcla

struct hpcla_packed {
    int64_t nrows = 0, nnz = 0, n_own = 0;
    int ndict = 0;
    short *dcol = nullptr;            // nnz + pad entries
    unsigned char *code = nullptr;    // nnz + pad entries
    double *dict = nullptr;           // 256 doubles
    int64_t bytes = 0;
};

namespace hpcla {

// ---- plan-time kernels -------------------------------------------------------------------------------
// encode values: code[j] = position of nzval[j] in the sorted dictionary; values not found are
// appended to `missing` (bounded list) so the host can extend the dictionary and retry.
__global__ __launch_bounds__(256) void encode_values_kernel(const double *__restrict__ nzval,
                                                            int64_t nnz,
                                                            const double *__restrict__ dict, int ndict,
                                                            unsigned char *__restrict__ code,
                                                            double *__restrict__ missing,
                                                            int *__restrict__ n_missing, int cap)
{
    __shared__ double s_dict[256];
    if ((int)threadIdx.x < ndict) s_dict[threadIdx.x] = dict[threadIdx.x];
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < nnz; i += stride) {
        const double v = nzval[i];
        int lo = 0, hi = ndict;                  // lower_bound on the bit pattern order used by the host
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_dict[mid] < v) lo = mid + 1; else hi = mid;
        }
        // compare bit patterns so that -0.0 / +0.0 and NaNs are never merged with another value
        const bool found = lo < ndict &&
                           __double_as_longlong(s_dict[lo]) == __double_as_longlong(v);
        if (found) {
            code[i] = (unsigned char)lo;
        } else {
            code[i] = 0;
            const int slot = atomicAdd(n_missing, 1);
            if (slot < cap) missing[slot] = v;
        }
    }
}

// encode columns of the listed row blocks; flags[0] |= 1 when an entry is a ghost column or outside
// the int16 window (block not packable).
template <typename I>
__global__ __launch_bounds__(256) void encode_cols_kernel(const I *__restrict__ rowptr,
                                                          const I *__restrict__ colval, int64_t nrows,
                                                          int base, int64_t n_own,
                                                          const int32_t *__restrict__ block_list,
                                                          short *__restrict__ dcol,
                                                          int *__restrict__ flags)
{
    const int64_t blk = block_list ? (int64_t)block_list[blockIdx.x] : (int64_t)blockIdx.x;
    const int64_t r0 = blk * P_RPB;
    const int nr = (int)((nrows - r0) < P_RPB ? (nrows - r0) : P_RPB);
    const int64_t p0 = (int64_t)rowptr[r0] - base, p1 = (int64_t)rowptr[r0 + nr] - base;
    int bad = 0;
    for (int64_t j = p0 + threadIdx.x; j < p1; j += 256) {
        const int64_t c = (int64_t)colval[j] - base;
        const int64_t d = c - r0;
        if (c >= n_own || d < -32768 || d > 32767) bad = 1;
        dcol[j] = (short)d;
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(flags, 1);
}

// ---- the packed SpMV kernel ----------------------------------------------------------------------------
__global__ __launch_bounds__(P_RPB) void spmv_packed_kernel(
    const int32_t *__restrict__ rowptr, const short *__restrict__ dcol,
    const unsigned char *__restrict__ code, const double *__restrict__ dict, int ndict,
    const double *__restrict__ x, int64_t n_own, double *__restrict__ y, int64_t nrows, int base,
    const int32_t *__restrict__ block_list, double *__restrict__ dot_partial)
{
    __shared__ double s_prod[P_CHUNK];
    __shared__ double s_dict[256];
    const int tid = threadIdx.x;
    if (tid < ndict) s_dict[tid] = dict[tid];
    const int64_t blk = block_list ? (int64_t)block_list[blockIdx.x] : (int64_t)blockIdx.x;
    const int64_t r0 = blk * P_RPB;
    const int nr = (int)((nrows - r0) < P_RPB ? (nrows - r0) : P_RPB);
    const int64_t p0 = (int64_t)rowptr[r0] - base, p1 = (int64_t)rowptr[r0 + nr] - base;
    const int64_t pa = p0 & ~(int64_t)7;           // octet-aligned start
    const int64_t total = p1 - pa;
    int lo = 0, hi = 0;
    if (tid < nr) {
        lo = (int)((int64_t)rowptr[r0 + tid] - base - pa);
        hi = (int)((int64_t)rowptr[r0 + tid + 1] - base - pa);
    }
    __syncthreads();
    double acc = 0.0;
    for (int64_t c = 0; c < total; c += P_CHUNK) {
        const int n = (int)((total - c) < P_CHUNK ? (total - c) : P_CHUNK);
        const int e0 = tid * 8;
        if (e0 < n) {
            const int64_t g = pa + c + e0;          // arrays are padded: the whole octet is readable
            const v8s dc = *reinterpret_cast<const v8s *>(dcol + g);
            const unsigned long long cd = *reinterpret_cast<const unsigned long long *>(code + g);
            double xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // octets shared with the neighbouring row blocks hold deltas relative to THEIR base:
                // clamp the index (those products are never summed here)
                int64_t idx = r0 + (int)dc[k];
                idx = idx < 0 ? 0 : (idx >= n_own ? n_own - 1 : idx);
                xv[k] = x[idx];
            }
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                double2 pr;
                pr.x = s_dict[(cd >> (8 * k)) & 0xff] * xv[k];
                pr.y = s_dict[(cd >> (8 * k + 8)) & 0xff] * xv[k + 1];
                *reinterpret_cast<double2 *>(&s_prod[e0 + k]) = pr;
            }
        }
        __syncthreads();
        {
            const int a = lo > c ? lo : (int)c;
            const int e = hi < c + n ? hi : (int)(c + n);
            for (int j = a; j < e; ++j) acc += s_prod[j - c];
        }
        __syncthreads();
    }
    if (tid < nr) __builtin_nontemporal_store(acc, y + r0 + tid);      // like the row-gather kernel's y (spmv.hip spmv_nt_y)
    if (dot_partial) {
        double v = tid < nr ? acc * x[r0 + tid] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) s_prod[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) dot_partial[blk] = ((s_prod[0] + s_prod[1]) + s_prod[2]) + s_prod[3];
    }
}

static void packed_free(hpcla_packed *p)
{
    if (!p) return;
    if (p->dcol) (void)hipFree(p->dcol);
    if (p->code) (void)hipFree(p->code);
    if (p->dict) (void)hipFree(p->dict);
    delete p;
}

}  // namespace hpcla

using namespace hpcla;

#define PK_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            packed_free(p);                                                                       \
            if (d_missing) (void)hipFree(d_missing);                                              \
            if (d_cnt) (void)hipFree(d_cnt);                                                      \
            return set_error(HPCLA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));       \
        }                                                                                         \
    } while (0)

HPCLA_API int hpcla_packed_create_i32(hpcla_packed_t **out, const int32_t *rowptr,
                                      const int32_t *colval_split, const double *nzval,
                                      int64_t nrows, int64_t nnz, int64_t n_own, int index_base,
                                      const int32_t *block_list, int64_t n_blocks, void *stream)
{
    if (!out) return set_error(HPCLA_ERR_INVALID, "packed_create: null output");
    if (nrows <= 0 || nnz <= 0 || n_own <= 0)
        return set_error(HPCLA_ERR_UNSUPPORTED, "packed_create: empty matrix");
    if (!rowptr || !colval_split || !nzval)
        return set_error(HPCLA_ERR_INVALID, "packed_create: null array");
    if (index_base != 0 && index_base != 1)
        return set_error(HPCLA_ERR_INVALID, "packed_create: index_base must be 0 or 1");
    const int64_t all_blocks = (nrows + P_RPB - 1) / P_RPB;
    const int64_t nb = block_list ? n_blocks : all_blocks;
    if (nb <= 0 || nb > all_blocks) return set_error(HPCLA_ERR_UNSUPPORTED, "packed_create: no blocks");
    hipStream_t s = as_stream(stream);
    hpcla_packed *p = new (std::nothrow) hpcla_packed();
    if (!p) return set_error(HPCLA_ERR_ALLOC, "packed_create: out of memory");
    double *d_missing = nullptr;
    int *d_cnt = nullptr;
    p->nrows = nrows; p->nnz = nnz; p->n_own = n_own;
    const int64_t padded = ((nnz + 7) / 8) * 8 + 8;
    const int CAP = 4096;
    PK_HIP(hipMalloc((void **)&p->dcol, padded * sizeof(short)));
    PK_HIP(hipMalloc((void **)&p->code, padded));
    PK_HIP(hipMalloc((void **)&p->dict, 256 * sizeof(double)));
    PK_HIP(hipMalloc((void **)&d_missing, CAP * sizeof(double)));
    PK_HIP(hipMalloc((void **)&d_cnt, 2 * sizeof(int)));
    PK_HIP(hipMemsetAsync(p->dcol, 0, padded * sizeof(short), s));
    PK_HIP(hipMemsetAsync(p->code, 0, padded, s));
    p->bytes = padded * 3 + 256 * 8;

    // ---- columns -----------------------------------------------------------------------------
    PK_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(int), s));
    encode_cols_kernel<int32_t><<<(uint32_t)nb, 256, 0, s>>>(rowptr, colval_split, nrows, index_base,
                                                             n_own, block_list, p->dcol, d_cnt + 1);
    PK_HIP(hipGetLastError());
    int h_flags[2] = {0, 0};
    PK_HIP(hipMemcpyAsync(h_flags, d_cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    PK_HIP(hipStreamSynchronize(s));
    if (h_flags[1]) {
        packed_free(p); (void)hipFree(d_missing); (void)hipFree(d_cnt);
        return set_error(HPCLA_ERR_UNSUPPORTED,
                         "packed_create: a listed row block has a ghost column or a column outside the 16-bit window");
    }

    // ---- value dictionary: seed from a sample, then extend with whatever the encoder misses --------
    std::vector<double> dict;
    {
        const int64_t ns = std::min<int64_t>(nnz, 1 << 16);
        std::vector<double> sample(ns);
        PK_HIP(hipMemcpy(sample.data(), nzval, ns * sizeof(double), hipMemcpyDeviceToHost));
        dict = sample;
    }
    auto bits_less = [](double a, double b) {
        if (a < b) return true;
        if (b < a) return false;
        long long ia, ib;
        memcpy(&ia, &a, 8); memcpy(&ib, &b, 8);
        return ia < ib;                        // orders -0.0 before +0.0; NaNs are rejected below
    };
    for (int iter = 0; iter < 64; ++iter) {
        for (double v : dict)
            if (v != v) {
                packed_free(p); (void)hipFree(d_missing); (void)hipFree(d_cnt);
                return set_error(HPCLA_ERR_UNSUPPORTED, "packed_create: NaN among the values");
            }
        std::sort(dict.begin(), dict.end(), bits_less);
        dict.erase(std::unique(dict.begin(), dict.end(), [](double a, double b) {
                       return memcmp(&a, &b, 8) == 0; }), dict.end());
        // the device lower_bound uses `<` on doubles: -0.0 and +0.0 compare equal there, so a
        // dictionary holding both cannot be searched -- treat as not packable
        for (size_t i = 1; i < dict.size(); ++i)
            if (!(dict[i - 1] < dict[i])) {
                packed_free(p); (void)hipFree(d_missing); (void)hipFree(d_cnt);
                return set_error(HPCLA_ERR_UNSUPPORTED, "packed_create: values contain both -0.0 and +0.0");
            }
        if (dict.size() > 256) {
            packed_free(p); (void)hipFree(d_missing); (void)hipFree(d_cnt);
            return set_error(HPCLA_ERR_UNSUPPORTED, "packed_create: more than 256 distinct values");
        }
        PK_HIP(hipMemcpyAsync(p->dict, dict.data(), dict.size() * sizeof(double), hipMemcpyHostToDevice, s));
        PK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int), s));
        int64_t g = (nnz + 255) / 256;
        if (g > 256 * 16) g = 256 * 16;
        encode_values_kernel<<<(uint32_t)g, 256, 0, s>>>(nzval, nnz, p->dict, (int)dict.size(), p->code,
                                                         d_missing, d_cnt, CAP);
        PK_HIP(hipGetLastError());
        int n_missing = 0;
        PK_HIP(hipMemcpyAsync(&n_missing, d_cnt, sizeof(int), hipMemcpyDeviceToHost, s));
        PK_HIP(hipStreamSynchronize(s));
        if (n_missing == 0) {
            p->ndict = (int)dict.size();
            (void)hipFree(d_missing); (void)hipFree(d_cnt);
            *out = p;
            return HPCLA_OK;
        }
        const int take = n_missing < CAP ? n_missing : CAP;
        std::vector<double> miss(take);
        PK_HIP(hipMemcpy(miss.data(), d_missing, take * sizeof(double), hipMemcpyDeviceToHost));
        dict.insert(dict.end(), miss.begin(), miss.end());
    }
    packed_free(p); (void)hipFree(d_missing); (void)hipFree(d_cnt);
    return set_error(HPCLA_ERR_UNSUPPORTED, "packed_create: dictionary did not converge");
}

HPCLA_API int hpcla_packed_destroy(hpcla_packed_t *p)
{
    packed_free(p);
    return HPCLA_OK;
}

HPCLA_API int hpcla_packed_info(const hpcla_packed_t *p, int64_t *bytes, int *ndict)
{
    if (!p) return set_error(HPCLA_ERR_INVALID, "packed_info: null handle");
    if (bytes) *bytes = p->bytes;
    if (ndict) *ndict = p->ndict;
    return HPCLA_OK;
}

// y = A*x with the listed (or all) row blocks read from the packed copy
HPCLA_API int hpcla_spmv_packed_f64_i32(const hpcla_packed_t *p, const int32_t *rowptr, const double *x,
                                        double *y, int index_base, const int32_t *block_list,
                                        int64_t n_blocks, double *dot_partial, void *stream)
{
    if (!p || !rowptr || !x || !y) return set_error(HPCLA_ERR_INVALID, "spmv_packed: null pointer");
    const int64_t all_blocks = (p->nrows + P_RPB - 1) / P_RPB;
    const int64_t nb = block_list ? n_blocks : all_blocks;
    if (nb < 0 || nb > all_blocks) return set_error(HPCLA_ERR_INVALID, "spmv_packed: n_blocks out of range");
    if (nb == 0) return HPCLA_OK;
    spmv_packed_kernel<<<(uint32_t)nb, P_RPB, 0, as_stream(stream)>>>(
        rowptr, p->dcol, p->code, p->dict, p->ndict, x, p->n_own, y, p->nrows, index_base, block_list,
        dot_partial);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

// distributed form: halo ∥ packed interior blocks -> CSR boundary blocks (ghost columns).
// With dot_out_dev != NULL also out = x.y (see hpcla_spmv_dist_dot_*).
HPCLA_API int hpcla_spmv_dist_packed_f64_i32(hpcla_halo_plan_t *plan, hpcla_comm_t *comm,
                                             const hpcla_packed_t *p, const int32_t *rowptr,
                                             const int32_t *colval_split, const double *nzval,
                                             const double *x, int64_t n_own, double *y, int64_t nrows,
                                             int64_t nnz, int index_base,
                                             const int32_t *interior_blocks, int64_t n_interior,
                                             const int32_t *boundary_blocks, int64_t n_boundary,
                                             double *dot_out_dev, void *work, void *stream)
{
    if (!p) return set_error(HPCLA_ERR_INVALID, "spmv_dist_packed: null packed handle");
    if (p->nrows != nrows || p->n_own != n_own)
        return set_error(HPCLA_ERR_INVALID, "spmv_dist_packed: handle does not match the matrix");
    double *scratch = nullptr, *partial = nullptr;
    const int64_t all_blocks = (nrows + P_RPB - 1) / P_RPB;
    if (dot_out_dev) {
        if (!work) return set_error(HPCLA_ERR_INVALID, "spmv_dist_packed: null work");
        if (n_own != nrows) return set_error(HPCLA_ERR_INVALID, "spmv_dist_packed: dot needs x partitioned like the rows");
        scratch = reinterpret_cast<double *>(work);
        partial = scratch + 2048;
    }
    const bool has_halo = halo_active(plan);
    int rc;
    if (!has_halo) {
        rc = hpcla_spmv_packed_f64_i32(p, rowptr, x, y, index_base, nullptr, 0, partial, stream);
        if (rc) return rc;
    } else {
        if (dot_out_dev && n_interior + n_boundary != all_blocks)
            return set_error(HPCLA_ERR_INVALID, "spmv_dist_packed: block lists must cover every row block");
        // same two orderings as the CSR step (comm.hip, spmv_dist_impl): exchange first on the caller's
        // stream, or on the side stream next to the interior blocks
        if (halo_mode_of(plan) == HALO_PUSH) {
            // push transport: the push kernel, the packed interior blocks, then the boundary blocks through the
            // CSR kernel's waiting form (they poll the flags themselves and find the ghost buffer of the epoch)
            // everything that can be refused is checked BEFORE the exchange is posted: push_post commits this step's
            // epoch readers (push workgroups + n_boundary waiting workgroups)
            if ((n_interior > 0 && !interior_blocks) || (n_boundary > 0 && !boundary_blocks))
                return set_error(HPCLA_ERR_INVALID, "spmv_dist_packed: null block list");
            if (n_interior < 0 || n_boundary < 0 || n_interior + n_boundary > all_blocks)
                return set_error(HPCLA_ERR_INVALID, "spmv_dist_packed: block counts out of range");
            if (!rowptr || !y || (nnz > 0 && (!colval_split || !nzval || !x)))
                return set_error(HPCLA_ERR_INVALID, "spmv_dist_packed: null array");
            rc = push_post(plan, x, n_boundary, stream);
            if (rc) return rc;
            if (n_interior > 0) {
                rc = hpcla_spmv_packed_f64_i32(p, rowptr, x, y, index_base, interior_blocks, n_interior, partial, stream);
                if (rc) { (void)push_abandon_waiters(plan, n_boundary, stream); return rc; }   // the posted readers still release
            }
            if (n_boundary > 0) {
                PushArgs nopush;
                memset(&nopush, 0, sizeof(nopush));
                rc = spmv_fused_i32(rowptr, colval_split, nzval, x, plan->ghost, n_own, y, nrows, nnz, index_base,
                                    nullptr, 0, 0, boundary_blocks, n_boundary, push_wait_args(plan, n_boundary),
                                    nopush, stream, partial);
                if (rc) { (void)push_abandon_waiters(plan, n_boundary, stream); return rc; }
            }
            if (dot_out_dev) {
                rc = reduce_partials_sum(partial, all_blocks, scratch, dot_out_dev, stream);
                if (rc) return rc;
                if (comm) return allreduce_on(comm, dot_out_dev, 1, 0, stream);
            }
            return HPCLA_OK;
        }
        const bool serial = halo_serial_mode(plan);
        rc = serial ? halo_exchange_inline(plan, x, stream) : hpcla_halo_begin(plan, x, stream);
        if (rc) return rc;
        if (n_interior > 0) {
            rc = hpcla_spmv_packed_f64_i32(p, rowptr, x, y, index_base, interior_blocks, n_interior, partial, stream);
            if (rc) return rc;
        }
        if (!serial) {
            rc = hpcla_halo_end(plan, stream);
            if (rc) return rc;
        }
        if (n_boundary > 0) {
            double *ghost = nullptr;
            rc = hpcla_halo_ghost_ptr(plan, &ghost, nullptr);
            if (rc) return rc;
            rc = spmv_split_i32(rowptr, colval_split, nzval, x, ghost, n_own, y, nrows, nnz, index_base,
                                boundary_blocks, n_boundary, stream, partial, -1);
            if (rc) return rc;
        }
    }
    if (dot_out_dev) {
        rc = reduce_partials_sum(partial, all_blocks, scratch, dot_out_dev, stream);
        if (rc) return rc;
        if (comm) return allreduce_on(comm, dot_out_dev, 1, 0, stream);
    }
    return HPCLA_OK;
}
