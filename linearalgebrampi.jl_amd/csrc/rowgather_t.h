// rowgather_t.h -- the "lanes = rows" row-gather stream for any value type and any operand strides (gfx950).
//
// The design of spmv.hip's row-gather kernel (a wave streams its 64 rows' entries into its own slice of LDS with aligned
// 16-byte loads, every lane then walks ITS row in stored order, separate multiply and add: the bits of the reference loop,
// src/sparse.jl:2055-2066), written once for
//   * T = float  : the Float32 element type (f32.hip), and
//   * T = double : dense operands in the CALLER's layout (colmajor.hip) -- Julia's Matrix is column-major, and with lanes =
//     rows the 64 rows of a wave read one contiguous run of a column per gather instruction, so A * B needs no layout
//     conversion around it (the tuned Float64 kernels of spmm.hip want row-major rows: two extra passes over B and C).
// The tuned Float64 SpMV (spmv.hip: fused halo wait / push, p.Ap epilogue, block orders) and row-major SpMM kernels (spmm.hip)
// stay what they are; this header serves the layouts and the type they do not.
//
// Ghost values (split column space: columns >= n_own) are read from the halo plan's ghost segment, which always holds
// doubles (the transports move 8-byte words; a Float32 exchange widens what it sends, f32.hip).
#pragma once
#include "common.h"

namespace hpcla {

constexpr int F_RPB = 256;        // rows per block == threads per block: the block lists of hpcla_classify_blocks_* apply
constexpr int F_CHW = 464;        // entries per wave pass: 64 rows x 7 + alignment slack (spmv.hip RG_CHW)
constexpr int F_NQ = (F_CHW / 4 + 63) / 64;   // quads per lane per pass

template <typename T, int N>
using fvec = T __attribute__((ext_vector_type(N)));

template <typename T>
struct DenseOperand {             // element (row j, column c) of a dense operand lives at p[j * rs + c * cs]
    const T *own;
    int64_t own_rs, own_cs;
    const double *ghost;          // ghost rows (the halo plan's buffer: doubles whatever T is), or null
    int64_t ghost_rs, ghost_cs;
    int64_t n_own;
};

template <typename T, bool SPLIT>
__device__ __forceinline__ T operand_gather(const DenseOperand<T> &b, int64_t col, int64_t coff_own, int64_t coff_ghost)
{
    if (SPLIT && col >= b.n_own) return (T)b.ghost[(col - b.n_own) * b.ghost_rs + coff_ghost];
    return b.own[col * b.own_rs + coff_own];
}

// four consecutive values, 16-byte aligned (one or two 16-byte words)
template <typename T> struct Quad;
template <> struct Quad<float> {
    fvec<float, 4> v;
    __device__ __forceinline__ void load(const float *p) { v = *reinterpret_cast<const fvec<float, 4> *>(p); }
    __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<fvec<float, 4> *>(p) = v; }
};
template <> struct Quad<double> {
    fvec<double, 2> a, b;
    __device__ __forceinline__ void load(const double *p)
    {
        a = *reinterpret_cast<const fvec<double, 2> *>(p);
        b = *reinterpret_cast<const fvec<double, 2> *>(p + 2);
    }
    __device__ __forceinline__ void store(double *p) const
    {
        *reinterpret_cast<fvec<double, 2> *>(p) = a;
        *reinterpret_cast<fvec<double, 2> *>(p + 2) = b;
    }
};

// One pass of a WAVE: entries [at, at + n) of colval / nzval into the wave's own slice of LDS (n <= F_CHW).  Aligned arrays:
// 16-byte loads, all of a lane's quads requested before the first LDS write (lanes past the end re-read the pass's last quad
// -- lines their neighbours read anyway -- and write nothing); unaligned arrays, or the one pass of the launch that reaches
// past their end: entry by entry.  LDS operations of one wave complete in order, so a wave-level fence is all the consumer
// needs.
template <typename T, typename I>
__device__ __forceinline__ void stage_pass(const I *__restrict__ colval, const T *__restrict__ nzval, I *s_col, T *s_val,
                                           int64_t at, int n, int64_t nnz, int base, int lane, int vec_ok)
{
    if (vec_ok && at + ((n + 3) & ~3) <= nnz) {
        const int last = (n - 1) & ~3;
        fvec<I, 4> cq[F_NQ];
        Quad<T> vq[F_NQ];
#pragma unroll
        for (int u = 0; u < F_NQ; ++u) {
            const int e0 = (u * 64 + lane) * 4;
            const int ee = e0 < last ? e0 : last;
            cq[u] = *reinterpret_cast<const fvec<I, 4> *>(colval + at + ee);
            vq[u].load(nzval + at + ee);
        }
#pragma unroll
        for (int u = 0; u < F_NQ; ++u) {
            const int e0 = (u * 64 + lane) * 4;
            if (e0 < n) {
                *reinterpret_cast<fvec<I, 4> *>(&s_col[e0]) = cq[u];
                vq[u].store(&s_val[e0]);
            }
        }
    } else {
        for (int e = lane; e < n; e += 64) {
            const int64_t g = at + e;
            s_col[e] = g < nnz ? colval[g] : (I)base;
            s_val[e] = g < nnz ? nzval[g] : (T)0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // this wave's LDS writes, then its LDS reads
    __builtin_amdgcn_wave_barrier();
}

// true when colval / nzval can be staged with 16-byte loads from any quad-aligned entry offset
template <typename T, typename I>
static inline int stage_vec_ok(const I *colval, const T *nzval)
{
    return (reinterpret_cast<uintptr_t>(colval) % (4 * sizeof(I)) == 0) && (reinterpret_cast<uintptr_t>(nzval) % 16 == 0) &&
           (sizeof(T) != 8 || reinterpret_cast<uintptr_t>(nzval) % 32 == 0);
}

// KC = 1: y = A*x (strides of the operand ignored: unit).  KC > 1: up to KC columns [c0, c0 + kc) of C = A*B.
template <typename T, typename I, bool SPLIT, int KC, int URX = 0>
__global__ __launch_bounds__(F_RPB) void rowgather_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const T *__restrict__ nzval, DenseOperand<T> b,
    T *__restrict__ y, int64_t y_rs, int64_t y_cs, int64_t nrows, int64_t nnz, int base, int k,
    const int32_t *__restrict__ block_list, int vec_ok, int nt_y)
{
    constexpr int UR = URX ? URX : (KC == 1 ? 8 : (KC <= 8 ? 4 : 2));   // entries per step (gathers in flight: UR x KC)
    __shared__ __attribute__((aligned(16))) I s_col_all[(F_RPB / 64) * F_CHW];
    __shared__ __attribute__((aligned(16))) T s_val_all[(F_RPB / 64) * F_CHW];

    const int tid = threadIdx.x;
    const int64_t blk = block_list ? (int64_t)block_list[blockIdx.x] : (int64_t)blockIdx.x;
    const int c0 = KC == 1 ? 0 : (int)blockIdx.y * KC;
    const int kc = KC == 1 ? 1 : (k - c0 < KC ? k - c0 : KC);
    const int64_t r0 = blk * F_RPB;
    const int nr = (int)((nrows - r0) < F_RPB ? (nrows - r0) : F_RPB);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    I *s_col = s_col_all + wave * F_CHW;
    T *s_val = s_val_all + wave * F_CHW;
    const int64_t rw = r0 + wave * 64;
    const int nrw = nr - wave * 64 < 0 ? 0 : (nr - wave * 64 > 64 ? 64 : nr - wave * 64);
    if (nrw <= 0) return;                                                // wave-uniform; the kernel has no workgroup barrier
    T acc[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) acc[c] = (T)0;
    const int64_t p0 = (int64_t)rowptr[rw] - base;
    const int64_t p1 = (int64_t)rowptr[rw + nrw] - base;
    const int64_t pa = vec_ok ? (p0 & ~(int64_t)3) : p0;                 // quad-aligned start (<= 3 entries of the rows before)
    const int64_t total = p1 - pa;
    const int ll = lane < nrw ? lane : nrw - 1;
    I rlo = rowptr[rw + ll], rhi = rowptr[rw + ll + 1];                  // unconditional; first used behind the A stream
    for (int64_t c = 0; c < total; c += F_CHW) {
        const int n = (int)((total - c) < F_CHW ? (total - c) : F_CHW);
        stage_pass<T, I>(colval, nzval, s_col, s_val, pa + c, n, nnz, base, lane, vec_ok);
        asm volatile("" : "+v"(rlo), "+v"(rhi));                         // keeps the row bounds' first use behind the stream
        {
            const int lo = lane < nrw ? (int)((int64_t)rlo - base - pa - c) : 0;
            const int hi = lane < nrw ? (int)((int64_t)rhi - base - pa - c) : 0;
            int j = lo > 0 ? lo : 0;
            const int e = hi < n ? hi : n;
            for (; j < e; j += UR) {
                int64_t cc[UR];
                T vv[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    cc[u] = 0; vv[u] = (T)0;
                    if (j + u < e) { cc[u] = (int64_t)(I)(s_col[j + u] - (I)base); vv[u] = s_val[j + u]; }
                }
                if (KC == 1) {
                    T xx[UR];
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        xx[u] = (T)0;
                        if (j + u < e) xx[u] = SPLIT && cc[u] >= b.n_own ? (T)b.ghost[cc[u] - b.n_own] : b.own[cc[u]];
                    }
#pragma unroll
                    for (int u = 0; u < UR; ++u) if (j + u < e) acc[0] += vv[u] * xx[u];
                } else {
                    T xx[UR][KC];
                    // column by column: the step's entries of ONE column leave back to back -- on a banded matrix the
                    // entries i-1, i, i+1 of a row read the same lines of that column, and with column-major operands the
                    // KC columns of one row sit a multiple of the L1's set stride apart (n * sizeof(T)), so an entry-major
                    // order lets KC - 1 other lines of the same set pass between two uses of a line
#pragma unroll
                    for (int q = 0; q < KC; ++q)
#pragma unroll
                        for (int u = 0; u < UR; ++u) {
                            xx[u][q] = (T)0;
                            if (j + u < e && q < kc)
                                xx[u][q] = operand_gather<T, SPLIT>(b, cc[u], (c0 + q) * b.own_cs, (c0 + q) * b.ghost_cs);
                        }
#pragma unroll
                    for (int u = 0; u < UR; ++u)
                        if (j + u < e) {
#pragma unroll
                            for (int q = 0; q < KC; ++q) acc[q] += vv[u] * xx[u][q];
                        }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");           // ... and the reads before the next pass's writes
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < nrw) {
        if (KC == 1) {
            if (nt_y) __builtin_nontemporal_store(acc[0], y + rw + lane);
            else y[rw + lane] = acc[0];
        } else {
#pragma unroll
            for (int q = 0; q < KC; ++q)
                if (q < kc) {
                    T *yp = y + (rw + lane) * y_rs + (c0 + q) * y_cs;
                    if (nt_y) __builtin_nontemporal_store(acc[q], yp);       // written once, read by nothing in the launch
                    else *yp = acc[q];
                }
        }
    }
}

}  // namespace hpcla
