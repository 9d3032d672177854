// spmv_variants.hip -- TUNING HARNESS ONLY (built into benchmarks/tune/libhpcla_tune.so by
// benchmarks/tune_spmv.py; never linked into libhpcla_rocm.so).  Candidate structures for the
// row-block stream SpMV, plus ablations that remove one phase at a time, all int32 / base 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr uint32_t NUM_XCD = 8;
__device__ __forceinline__ uint32_t xcd_slice_index(uint32_t b, uint32_t n)
{
    uint32_t k = b % NUM_XCD, q = b / NUM_XCD;
    uint32_t per = n / NUM_XCD, rem = n % NUM_XCD;
    return k * per + (k < rem ? k : rem) + q;
}

template <typename T, bool NT>
__device__ __forceinline__ T ld(const T *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// ---- narrow (element-per-lane) kernel: the production structure --------------------------------
// ABL: 0 full, 1 no LDS (thread sums its own products -> wrong y, same traffic), 2 no gather either
template <int TPB, int RPT, int UNROLL, bool NT, bool XCD, int ABL>
__global__ __launch_bounds__(TPB) void k_narrow(const int *__restrict__ rowptr,
                                                const int *__restrict__ colval,
                                                const double *__restrict__ nzval,
                                                const double *__restrict__ x, double *__restrict__ y,
                                                int64_t nrows, uint32_t nblocks)
{
    constexpr int R = TPB * RPT;
    constexpr int CHUNK = TPB * UNROLL;
    __shared__ double s_prod[CHUNK];
    const int tid = threadIdx.x;
    const uint32_t b = XCD ? xcd_slice_index(blockIdx.x, nblocks) : blockIdx.x;
    const int64_t r0 = (int64_t)b * R;
    const int nr = (int)((nrows - r0) < R ? (nrows - r0) : R);
    const int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    const int64_t total = p1 - p0;
    int lo[RPT], hi[RPT];
    double acc[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        lo[q] = hi[q] = 0;
        acc[q] = 0.0;
        if (r < nr) { lo[q] = (int)(rowptr[r0 + r] - p0); hi[q] = (int)(rowptr[r0 + r + 1] - p0); }
    }
    for (int64_t c = 0; c < total; c += CHUNK) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        const int *cv = colval + p0 + c;
        const double *nv = nzval + p0 + c;
        int col[UNROLL];
        double val[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = tid + u * TPB;
            if (i < n) { col[u] = ld<int, NT>(cv + i); val[u] = ld<double, NT>(nv + i); }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = tid + u * TPB;
            if (i < n) {
                if (ABL == 0) s_prod[i] = val[u] * x[col[u]];
                if (ABL == 1) acc[0] += val[u] * x[col[u]];
                if (ABL == 2) acc[0] += val[u] * (double)col[u];
            }
        }
        if (ABL == 0) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int a = lo[q] > c ? lo[q] : (int)c;
                const int e = hi[q] < c + n ? hi[q] : (int)(c + n);
                for (int j = a; j < e; ++j) acc[q] += s_prod[j - c];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        if (r < nr) y[r0 + r] = acc[q] + (ABL ? (double)(lo[q] + hi[q]) * 1e-300 : 0.0);
    }
}

// ---- wide kernel: 16-byte loads (4 entries per lane per load), aligned to 4 entries ----------------
// XCD: 0 natural order, 1 one contiguous slice per XCD, G >= 2: groups of G consecutive row blocks per XCD, groups
// dealt round-robin (the hardware deals workgroups to XCDs round-robin by blockIdx: blocks b and b +- 2 -- the
// +-512-row neighbours of config 4's grid -- then share an L2, and the launch still walks ONE moving window)
__device__ __forceinline__ uint32_t xcd_group_index(uint32_t b, uint32_t n, uint32_t G)
{
    const uint32_t span = NUM_XCD * G;
    if (b >= n - n % span) return b;                      // ragged tail: natural order
    const uint32_t xcd = b % NUM_XCD, q = b / NUM_XCD;
    return ((q / G) * NUM_XCD + xcd) * G + q % G;
}

template <int TPB, int RPT, int U, bool NT, int XCD>
__global__ __launch_bounds__(TPB) void k_wide(const int *__restrict__ rowptr,
                                              const int *__restrict__ colval,
                                              const double *__restrict__ nzval,
                                              const double *__restrict__ x, double *__restrict__ y,
                                              int64_t nrows, int64_t nnz, uint32_t nblocks)
{
    constexpr int R = TPB * RPT;
    constexpr int CHUNK = TPB * 4 * U;
    __shared__ double s_prod[CHUNK];
    const int tid = threadIdx.x;
    const uint32_t b = XCD == 0 ? blockIdx.x : XCD == 1 ? xcd_slice_index(blockIdx.x, nblocks) : xcd_group_index(blockIdx.x, nblocks, XCD);
    const int64_t r0 = (int64_t)b * R;
    const int nr = (int)((nrows - r0) < R ? (nrows - r0) : R);
    const int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    const int64_t pa = p0 & ~(int64_t)3;
    const int64_t total = p1 - pa;
    int lo[RPT], hi[RPT];
    double acc[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        lo[q] = hi[q] = 0;
        acc[q] = 0.0;
        if (r < nr) { lo[q] = (int)(rowptr[r0 + r] - pa); hi[q] = (int)(rowptr[r0 + r + 1] - pa); }
    }
    for (int64_t c = 0; c < total; c += CHUNK) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        v4i col[U];
        v2d va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e0 = (u * TPB + tid) * 4;
            const int64_t g = pa + c + e0;
            col[u] = (v4i)(0);
            va[u] = (v2d)(0.0);
            vb[u] = (v2d)(0.0);
            if (e0 < n) {
                if (g + 3 < nnz) {
                    col[u] = ld<v4i, NT>(reinterpret_cast<const v4i *>(colval + g));
                    va[u] = ld<v2d, NT>(reinterpret_cast<const v2d *>(nzval + g));
                    vb[u] = ld<v2d, NT>(reinterpret_cast<const v2d *>(nzval + g + 2));
                } else {
                    if (g + 0 < nnz) { col[u].x = colval[g + 0]; va[u].x = nzval[g + 0]; }
                    if (g + 1 < nnz) { col[u].y = colval[g + 1]; va[u].y = nzval[g + 1]; }
                    if (g + 2 < nnz) { col[u].z = colval[g + 2]; vb[u].x = nzval[g + 2]; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e0 = (u * TPB + tid) * 4;
            if (e0 < n) {
                double2 pa2, pb2;
                pa2.x = va[u].x * x[col[u].x];
                pa2.y = va[u].y * x[col[u].y];
                pb2.x = vb[u].x * x[col[u].z];
                pb2.y = vb[u].y * x[col[u].w];
                *reinterpret_cast<double2 *>(&s_prod[e0]) = pa2;
                *reinterpret_cast<double2 *>(&s_prod[e0 + 2]) = pb2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int a = lo[q] > c ? lo[q] : (int)c;
            const int e = hi[q] < c + n ? hi[q] : (int)(c + n);
            for (int j = a; j < e; ++j) acc[q] += s_prod[j - c];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        if (r < nr) y[r0 + r] = acc[q];
    }
}

// ---- ROWG (round 4): gathers issued BY ROW, from an LDS-transposed copy of the block's A entries -------------------
// Why (profiles/r04_spmv_2d_vs_3d_counters.txt): in the product-parking kernels a lane owns a QUAD of consecutive
// entries, so one gather instruction covers entries 4L + q, L = 0..63 -- ~37 rows x 7 column types of the 7-point
// matrix, i.e. seven x streams of which five are aligned to the same 4 KiB / 2 MiB power of two (i - 2 MiB, i - 4 KiB,
// i, i + 4 KiB, i + 2 MiB): ~24 lines per instruction, most of them on the same tag bank of the vector L1
// (TCP_READ_TAGCONFLICT_STALL_CYCLES 6.2 x the 5-point matrix's per entry, TA_ADDR_STALLED_BY_TC 1.6 x, SQ issue
// stalls 1.6 x).  Here the block's colval / nzval are streamed coalesced into LDS UNMULTIPLIED (12 B per entry), and
// thread t then walks ITS OWN row: gather instruction j reads, across the wave, the j-th entry of 64 consecutive rows
// -- for any banded / stencil matrix ONE x stream, 512 contiguous bytes, 4-5 lines and no conflict; for unstructured
// rows no worse than before.  The row sum is the same sequential multiply-add chain in stored order: same bits.
// LDS reads: s_col stride = row length in 4-byte words (5, 7: odd -> conflict-free), s_val in 8-byte words.
template <int TPB, int CH, int UR, int XCD>
__global__ __launch_bounds__(TPB) void k_rowg(const int *__restrict__ rowptr, const int *__restrict__ colval,
                                              const double *__restrict__ nzval, const double *__restrict__ x,
                                              double *__restrict__ y, int64_t nrows, int64_t nnz, uint32_t nblocks)
{
    static_assert(CH % 4 == 0, "whole quads");
    __shared__ __attribute__((aligned(16))) int s_col[CH];
    __shared__ __attribute__((aligned(16))) double s_val[CH];
    const int tid = threadIdx.x;
    const uint32_t b = XCD == 0 ? blockIdx.x : xcd_group_index(blockIdx.x, nblocks, XCD);
    const int64_t r0 = (int64_t)b * TPB;
    const int nr = (int)((nrows - r0) < TPB ? (nrows - r0) : TPB);
    const int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    const int64_t pa = p0 & ~(int64_t)3;
    const int64_t total = p1 - pa;
    int rlo = 0, rhi = 0;
    if (tid < nr) { rlo = rowptr[r0 + tid]; rhi = rowptr[r0 + tid + 1]; }
    double acc = 0.0;
    for (int64_t c = 0; c < total; c += CH) {
        const int n = (int)((total - c) < CH ? (total - c) : CH);
        // stream the pass's entries into LDS: ALL of a lane's quads are requested before the first is written (lanes past
        // the end re-read the pass's last quad -- lines their neighbours read anyway -- and write nothing)
        if (pa + c + ((n + 3) & ~3) <= nnz) {
            constexpr int NQ = (CH / 4 + TPB - 1) / TPB;
            const int last = (n - 1) & ~3;
            v4i cq[NQ];
            v2d va[NQ], vb[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int e0 = (u * TPB + tid) * 4;
                const int ee = e0 < last ? e0 : last;
                cq[u] = *reinterpret_cast<const v4i *>(colval + pa + c + ee);
                va[u] = *reinterpret_cast<const v2d *>(nzval + pa + c + ee);
                vb[u] = *reinterpret_cast<const v2d *>(nzval + pa + c + ee + 2);
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int e0 = (u * TPB + tid) * 4;
                if (e0 < n) {
                    *reinterpret_cast<v4i *>(&s_col[e0]) = cq[u];
                    *reinterpret_cast<v2d *>(&s_val[e0]) = va[u];
                    *reinterpret_cast<v2d *>(&s_val[e0 + 2]) = vb[u];
                }
            }
        } else {                                  // the one pass that reaches past the arrays' end: entry by entry
            for (int e = tid; e < n; e += TPB) {
                const int64_t g = pa + c + e;
                s_col[e] = g < nnz ? colval[g] : 0;
                s_val[e] = g < nnz ? nzval[g] : 0.0;
            }
        }
        __syncthreads();
        {
            const int lo = tid < nr ? (int)(rlo - pa) : 0, hi = tid < nr ? (int)(rhi - pa) : 0;
            int j = (lo > c ? lo : (int)c) - (int)c;
            const int e = (hi < c + n ? hi : (int)(c + n)) - (int)c;
            // UR entries per step, each under its own lane predicate: the step's gathers leave together (a gather none of
            // the wave's lanes needs is skipped), the sums follow in stored order
            for (; j < e; j += UR) {
                int cc[UR];
                double vv[UR], xx[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) { cc[u] = 0; vv[u] = 0.0; if (j + u < e) { cc[u] = s_col[j + u]; vv[u] = s_val[j + u]; } }
#pragma unroll
                for (int u = 0; u < UR; ++u) { xx[u] = 0.0; if (j + u < e) xx[u] = x[cc[u]]; }
#pragma unroll
                for (int u = 0; u < UR; ++u) if (j + u < e) acc += vv[u] * xx[u];
            }
        }
        if (c + CH < total) __syncthreads();
    }
    if (tid < nr) y[r0 + tid] = acc;
}

// ---- ROWG, wave-private tiles: every WAVE owns 64 rows and its own slice of LDS; no workgroup barrier at all (LDS
// operations of one wave complete in order), so a workgroup of WPB waves keeps the 64 * WPB-row block granularity of the
// callers' block lists while its waves never wait for each other
template <int WPB, int CHW, int UR, int XCD, bool DMA = false, bool NTA = false, bool NTY = false>
__global__ __launch_bounds__(64 * WPB) void k_rowg_wave(const int *__restrict__ rowptr, const int *__restrict__ colval,
                                                        const double *__restrict__ nzval, const double *__restrict__ x,
                                                        double *__restrict__ y, int64_t nrows, int64_t nnz, uint32_t nblocks)
{
    static_assert(CHW % 256 == 0, "whole quads per lane");
    __shared__ __attribute__((aligned(16))) int s_col_all[WPB * CHW];
    __shared__ __attribute__((aligned(16))) double s_val_all[WPB * CHW];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    int *s_col = s_col_all + wave * CHW;
    double *s_val = s_val_all + wave * CHW;
    const uint32_t b = XCD == 0 ? blockIdx.x : xcd_group_index(blockIdx.x, nblocks, XCD);
    const int64_t r0 = (int64_t)b * (64 * WPB) + wave * 64;
    if (r0 >= nrows) return;                               // (no barrier anywhere below)
    const int nr = (int)((nrows - r0) < 64 ? (nrows - r0) : 64);
    const int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    const int64_t pa = p0 & ~(int64_t)3;
    const int64_t total = p1 - pa;
    int rlo = 0, rhi = 0;
    if (lane < nr) { rlo = rowptr[r0 + lane]; rhi = rowptr[r0 + lane + 1]; }
    double acc = 0.0;
    for (int64_t c = 0; c < total; c += CHW) {
        const int n = (int)((total - c) < CHW ? (total - c) : CHW);
        if (DMA && pa + c + CHW <= nnz) {
            // LDS-DMA: the pass's colval / nzval pieces go straight into LDS, 1 KiB per wave-instruction (lane L carries
            // 16 bytes at L * 16 of the piece), no VGPR destination and no ds_write; pieces past the pass's end copy the
            // next wave's entries (inside the arrays: checked above), which nobody reads
#pragma unroll
            for (int i = 0; i < CHW / 256; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(colval + pa + c + i * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(s_col + i * 256), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < CHW / 128; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nzval + pa + c + i * 128 + lane * 2),
                                                 (__attribute__((address_space(3))) void *)(s_val + i * 128), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (pa + c + ((n + 3) & ~3) <= nnz) {
            constexpr int NQ = CHW / 256;
            const int last = (n - 1) & ~3;
            v4i cq[NQ];
            v2d va[NQ], vb[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int e0 = (u * 64 + lane) * 4;
                const int ee = e0 < last ? e0 : last;
                cq[u] = ld<v4i, NTA>(reinterpret_cast<const v4i *>(colval + pa + c + ee));
                va[u] = ld<v2d, NTA>(reinterpret_cast<const v2d *>(nzval + pa + c + ee));
                vb[u] = ld<v2d, NTA>(reinterpret_cast<const v2d *>(nzval + pa + c + ee + 2));
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int e0 = (u * 64 + lane) * 4;
                if (e0 < n) {
                    *reinterpret_cast<v4i *>(&s_col[e0]) = cq[u];
                    *reinterpret_cast<v2d *>(&s_val[e0]) = va[u];
                    *reinterpret_cast<v2d *>(&s_val[e0 + 2]) = vb[u];
                }
            }
        } else {
            for (int e = lane; e < n; e += 64) {
                const int64_t g = pa + c + e;
                s_col[e] = g < nnz ? colval[g] : 0;
                s_val[e] = g < nnz ? nzval[g] : 0.0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // this wave's LDS writes before its LDS reads (in-order per wave)
        __builtin_amdgcn_wave_barrier();
        {
            const int lo = lane < nr ? (int)(rlo - pa) : 0, hi = lane < nr ? (int)(rhi - pa) : 0;
            int j = (lo > c ? lo : (int)c) - (int)c;
            const int e = (hi < c + n ? hi : (int)(c + n)) - (int)c;
            for (; j < e; j += UR) {
                int cc[UR];
                double vv[UR], xx[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) { cc[u] = 0; vv[u] = 0.0; if (j + u < e) { cc[u] = s_col[j + u]; vv[u] = s_val[j + u]; } }
#pragma unroll
                for (int u = 0; u < UR; ++u) { xx[u] = 0.0; if (j + u < e) xx[u] = x[cc[u]]; }
#pragma unroll
                for (int u = 0; u < UR; ++u) if (j + u < e) acc += vv[u] * xx[u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // the reads before the next pass's writes
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < nr) { if (NTY) __builtin_nontemporal_store(acc, y + r0 + lane); else y[r0 + lane] = acc; }
}

// ---- wide kernel + compact per-block pointer array (16 block starts per 64-byte line)
// ---- wide kernel: 16-byte loads (4 entries per lane per load), aligned to 4 entries ----------------
template <int TPB, int RPT, int U, bool NT, bool XCD>
__global__ __launch_bounds__(TPB) void k_wide_bptr(const int *__restrict__ rowptr,
                                              const int *__restrict__ colval,
                                              const double *__restrict__ nzval,
                                              const double *__restrict__ x, double *__restrict__ y,
                                              int64_t nrows, int64_t nnz, uint32_t nblocks, const int *__restrict__ bptr)
{
    constexpr int R = TPB * RPT;
    constexpr int CHUNK = TPB * 4 * U;
    __shared__ double s_prod[CHUNK];
    const int tid = threadIdx.x;
    const uint32_t b = XCD ? xcd_slice_index(blockIdx.x, nblocks) : blockIdx.x;
    const int64_t r0 = (int64_t)b * R;
    const int nr = (int)((nrows - r0) < R ? (nrows - r0) : R);
    const int64_t p0 = bptr[b], p1 = bptr[b + 1];
    const int64_t pa = p0 & ~(int64_t)3;
    const int64_t total = p1 - pa;
    int lo[RPT], hi[RPT];
    double acc[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        lo[q] = hi[q] = 0;
        acc[q] = 0.0;
        if (r < nr) { lo[q] = (int)(rowptr[r0 + r] - pa); hi[q] = (int)(rowptr[r0 + r + 1] - pa); }
    }
    for (int64_t c = 0; c < total; c += CHUNK) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        v4i col[U];
        v2d va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e0 = (u * TPB + tid) * 4;
            const int64_t g = pa + c + e0;
            col[u] = (v4i)(0);
            va[u] = (v2d)(0.0);
            vb[u] = (v2d)(0.0);
            if (e0 < n) {
                if (g + 3 < nnz) {
                    col[u] = ld<v4i, NT>(reinterpret_cast<const v4i *>(colval + g));
                    va[u] = ld<v2d, NT>(reinterpret_cast<const v2d *>(nzval + g));
                    vb[u] = ld<v2d, NT>(reinterpret_cast<const v2d *>(nzval + g + 2));
                } else {
                    if (g + 0 < nnz) { col[u].x = colval[g + 0]; va[u].x = nzval[g + 0]; }
                    if (g + 1 < nnz) { col[u].y = colval[g + 1]; va[u].y = nzval[g + 1]; }
                    if (g + 2 < nnz) { col[u].z = colval[g + 2]; vb[u].x = nzval[g + 2]; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e0 = (u * TPB + tid) * 4;
            if (e0 < n) {
                double2 pa2, pb2;
                pa2.x = va[u].x * x[col[u].x];
                pa2.y = va[u].y * x[col[u].y];
                pb2.x = vb[u].x * x[col[u].z];
                pb2.y = vb[u].y * x[col[u].w];
                *reinterpret_cast<double2 *>(&s_prod[e0]) = pa2;
                *reinterpret_cast<double2 *>(&s_prod[e0 + 2]) = pb2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int a = lo[q] > c ? lo[q] : (int)c;
            const int e = hi[q] < c + n ? hi[q] : (int)(c + n);
            for (int j = a; j < e; ++j) acc[q] += s_prod[j - c];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        if (r < nr) y[r0 + r] = acc[q];
    }
}

// ---- pure stream: read the three arrays + write y with 16-byte accesses (ceiling for this byte mix)
__global__ __launch_bounds__(256) void k_copy(const v4i *__restrict__ colval4,
                                              const v2d *__restrict__ nz2,
                                              const v4i *__restrict__ rowptr4, double2 *__restrict__ y2,
                                              const double2 *__restrict__ x2, int64_t nnz, int64_t nrows)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    double acc = 0.0;
    for (int64_t j = i; j < nnz / 4; j += stride) {
        v4i c = __builtin_nontemporal_load(colval4 + j);
        v2d a = __builtin_nontemporal_load(nz2 + 2 * j), b = __builtin_nontemporal_load(nz2 + 2 * j + 1);
        acc += a.x + a.y + b.x + b.y + (double)(c.x ^ c.y ^ c.z ^ c.w);
    }
    for (int64_t j = i; j < nrows / 4; j += stride) {
        v4i r = __builtin_nontemporal_load(rowptr4 + j);
        acc += (double)(r.x ^ r.y ^ r.z ^ r.w);
    }
    for (int64_t j = i; j < nrows / 2; j += stride) {
        double2 xv = x2[j];
        y2[j] = make_double2(acc + xv.x, acc + xv.y);
    }
}

// ---- software-pipelined persistent kernel ---------------------------------------------------------
// Each workgroup walks row blocks b = blockIdx.x, +gridDim.x, ...; work item = one CHUNK of one row
// block.  While the products of the current item go through LDS and are summed, the colval/nzval
// loads of the NEXT item (and the rowptr entries of the next row block) are already in flight.
// vmcnt is in-order, so per iteration: issue the x gathers of the current item FIRST, then the
// prefetch loads, then wait only for the gathers.
template <int U>
struct ChunkRegs {
    v4i col[U];
    v2d va[U], vb[U];
};

template <int TPB, int U, bool NT>
__device__ __forceinline__ void load_chunk(ChunkRegs<U> &r, const int *__restrict__ colval,
                                           const double *__restrict__ nzval, int64_t pa, int64_t c,
                                           int64_t total, int64_t nnz, int tid)
{
    const int n = (int)((total - c) < (int64_t)TPB * 4 * U ? (total - c) : (int64_t)TPB * 4 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e0 = (u * TPB + tid) * 4;
        const int64_t g = pa + c + e0;
        r.col[u] = (v4i)(0);
        r.va[u] = (v2d)(0.0);
        r.vb[u] = (v2d)(0.0);
        if (e0 < n) {
            if (g + 3 < nnz) {
                r.col[u] = ld<v4i, NT>(reinterpret_cast<const v4i *>(colval + g));
                r.va[u] = ld<v2d, NT>(reinterpret_cast<const v2d *>(nzval + g));
                r.vb[u] = ld<v2d, NT>(reinterpret_cast<const v2d *>(nzval + g + 2));
            } else {
                if (g + 0 < nnz) { r.col[u].x = colval[g + 0]; r.va[u].x = nzval[g + 0]; }
                if (g + 1 < nnz) { r.col[u].y = colval[g + 1]; r.va[u].y = nzval[g + 1]; }
                if (g + 2 < nnz) { r.col[u].z = colval[g + 2]; r.vb[u].x = nzval[g + 2]; }
            }
        }
    }
}

template <int TPB, int RPT, int U, bool NT>
__global__ __launch_bounds__(TPB) void k_pipe(const int *__restrict__ rowptr,
                                              const int *__restrict__ colval,
                                              const double *__restrict__ nzval,
                                              const double *__restrict__ x, double *__restrict__ y,
                                              int64_t nrows, int64_t nnz, int64_t nblk)
{
    constexpr int R = TPB * RPT;
    constexpr int CHUNK = TPB * 4 * U;
    __shared__ double s_prod[CHUNK];
    const int tid = threadIdx.x;
    int64_t b = blockIdx.x;
    if (b >= nblk) return;
    const int64_t gstride = gridDim.x;

    // current row block
    int64_t r0 = b * R;
    int nr = (int)((nrows - r0) < R ? (nrows - r0) : R);
    int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    int64_t pa = p0 & ~(int64_t)3, total = p1 - pa;
    int lo[RPT], hi[RPT];
    double acc[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        lo[q] = hi[q] = 0;
        acc[q] = 0.0;
        if (r < nr) { lo[q] = (int)(rowptr[r0 + r] - pa); hi[q] = (int)(rowptr[r0 + r + 1] - pa); }
    }
    // rowptr ends of the NEXT row block (prefetched one block ahead)
    int64_t nb = b + gstride;
    int64_t nb_p0 = 0, nb_p1 = 0;
    if (nb < nblk) {
        const int64_t q0 = nb * R;
        const int qn = (int)((nrows - q0) < R ? (nrows - q0) : R);
        nb_p0 = rowptr[q0];
        nb_p1 = rowptr[q0 + qn];
    }
    int64_t c = 0;
    ChunkRegs<U> cur, nxt;
    load_chunk<TPB, U, NT>(cur, colval, nzval, pa, c, total, nnz, tid);

    while (true) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        // 1. gathers of the current item
        double xg[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e0 = (u * TPB + tid) * 4;
            if (e0 < n) {
                xg[u][0] = x[cur.col[u].x];
                xg[u][1] = x[cur.col[u].y];
                xg[u][2] = x[cur.col[u].z];
                xg[u][3] = x[cur.col[u].w];
            }
        }
        // 2. prefetch the next item
        const bool last_chunk = (c + CHUNK >= total);
        const bool has_next = !last_chunk || nb < nblk;
        int64_t n_pa = pa, n_total = total, n_c = c + CHUNK;
        int nlo[RPT], nhi[RPT];
        int n_nr = nr;
        int64_t n_r0 = r0;
        int64_t nn_p0 = 0, nn_p1 = 0;
        if (last_chunk && nb < nblk) {
            n_r0 = nb * R;
            n_nr = (int)((nrows - n_r0) < R ? (nrows - n_r0) : R);
            n_pa = nb_p0 & ~(int64_t)3;
            n_total = nb_p1 - n_pa;
            n_c = 0;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + q * TPB;
                nlo[q] = nhi[q] = 0;
                if (r < n_nr) { nlo[q] = (int)(rowptr[n_r0 + r] - n_pa); nhi[q] = (int)(rowptr[n_r0 + r + 1] - n_pa); }
            }
            const int64_t nnb = nb + gstride;
            if (nnb < nblk) {
                const int64_t q0 = nnb * R;
                const int qn = (int)((nrows - q0) < R ? (nrows - q0) : R);
                nn_p0 = rowptr[q0];
                nn_p1 = rowptr[q0 + qn];
            }
        }
        if (has_next) load_chunk<TPB, U, NT>(nxt, colval, nzval, n_pa, n_c, n_total, nnz, tid);

        // 3. products -> LDS -> per-row sequential sums
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e0 = (u * TPB + tid) * 4;
            if (e0 < n) {
                double2 pa2, pb2;
                pa2.x = cur.va[u].x * xg[u][0];
                pa2.y = cur.va[u].y * xg[u][1];
                pb2.x = cur.vb[u].x * xg[u][2];
                pb2.y = cur.vb[u].y * xg[u][3];
                *reinterpret_cast<double2 *>(&s_prod[e0]) = pa2;
                *reinterpret_cast<double2 *>(&s_prod[e0 + 2]) = pb2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int a = lo[q] > c ? lo[q] : (int)c;
            const int e = hi[q] < c + n ? hi[q] : (int)(c + n);
            for (int j = a; j < e; ++j) acc[q] += s_prod[j - c];
        }
        __syncthreads();

        if (last_chunk) {
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + q * TPB;
                if (r < nr) y[r0 + r] = acc[q];
                acc[q] = 0.0;
            }
            if (nb >= nblk) break;
            b = nb; r0 = n_r0; nr = n_nr; pa = n_pa; total = n_total; c = 0;
#pragma unroll
            for (int q = 0; q < RPT; ++q) { lo[q] = nlo[q]; hi[q] = nhi[q]; }
            nb = b + gstride; nb_p0 = nn_p0; nb_p1 = nn_p1;
        } else {
            c += CHUNK;
        }
        cur = nxt;
    }
}

// ---- branch-free pipelined kernel -------------------------------------------------------------------
// Same structure as k_pipe, but no exec-masked branches around loads (hipcc waits vmcnt(0) at their
// joins, which drains the prefetch): lanes past the end of a chunk re-read the chunk's first quad
// (same line as lane 0, no extra HBM traffic) and write garbage products that are never summed.
// Requires colval 16-byte and nzval 32-byte aligned (a quad never straddles a page).
template <int TPB, int U, bool NT>
__device__ __forceinline__ void load_chunk_bf(ChunkRegs<U> &r, const int *__restrict__ colval,
                                              const double *__restrict__ nzval, int64_t base_elem,
                                              int n, int tid)
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int e0 = (u * TPB + tid) * 4;
        e0 = e0 < n ? e0 : 0;
        const int64_t g = base_elem + e0;
        r.col[u] = ld<v4i, NT>(reinterpret_cast<const v4i *>(colval + g));
        r.va[u] = ld<v2d, NT>(reinterpret_cast<const v2d *>(nzval + g));
        r.vb[u] = ld<v2d, NT>(reinterpret_cast<const v2d *>(nzval + g + 2));
    }
}

template <int TPB, int RPT, int U, bool NT, int PEEL>
__global__ __launch_bounds__(TPB) void k_pipe2(const int *__restrict__ rowptr,
                                               const int *__restrict__ colval,
                                               const double *__restrict__ nzval,
                                               const double *__restrict__ x, double *__restrict__ y,
                                               int64_t nrows, int64_t nnz, int64_t nblk)
{
    constexpr int R = TPB * RPT;
    constexpr int CHUNK = TPB * 4 * U;
    __shared__ double s_prod[CHUNK];
    const int tid = threadIdx.x;
    int64_t b = blockIdx.x;
    if (b >= nblk) return;
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));   // opaque 0: keeps block-uniform rowptr reads on the vector
                                                     // memory path (in-order vmcnt) instead of SMEM + lgkmcnt(0)
    const int64_t gstride = gridDim.x;

    int64_t r0 = b * R;
    int nr = (int)((nrows - r0) < R ? (nrows - r0) : R);
    int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    int64_t pa = p0 & ~(int64_t)3, total = p1 - pa;
    int lo[RPT], hi[RPT];
    double acc[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        const int rc = r < nr ? r : nr - 1;
        const int a = rowptr[r0 + rc], e = rowptr[r0 + rc + 1];
        lo[q] = r < nr ? (int)(a - pa) : 0;
        hi[q] = r < nr ? (int)(e - pa) : 0;
        acc[q] = 0.0;
    }
    int64_t nb = b + gstride;
    int64_t nb_p0 = 0, nb_p1 = 0;
    {
        const int64_t nbc = nb < nblk ? nb : b;
        const int64_t q0 = nbc * R;
        const int qn = (int)((nrows - q0) < R ? (nrows - q0) : R);
        nb_p0 = rowptr[q0];
        nb_p1 = rowptr[q0 + qn];
    }
    int64_t c = 0;
    ChunkRegs<U> cur, nxt;
    load_chunk_bf<TPB, U, NT>(cur, colval, nzval, pa, (int)(total < CHUNK ? total : CHUNK), tid);

    while (true) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        // 1. gathers of the current item (every lane: masked lanes hold the first quad's columns)
        double xg[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xg[u][0] = x[cur.col[u].x];
            xg[u][1] = x[cur.col[u].y];
            xg[u][2] = x[cur.col[u].z];
            xg[u][3] = x[cur.col[u].w];
        }
        // 2. prefetch the next item (uniform control flow only)
        const bool last_chunk = (c + CHUNK >= total);
        const bool more_blocks = nb < nblk;
        int64_t n_pa = pa, n_total = total, n_c = c + CHUNK, n_r0 = r0;
        int n_nr = nr;
        if (last_chunk) {
            const int64_t nbc = more_blocks ? nb : b;       // clamp: re-read own block when finished
            n_r0 = nbc * R;
            n_nr = (int)((nrows - n_r0) < R ? (nrows - n_r0) : R);
            n_pa = (more_blocks ? nb_p0 : p0) & ~(int64_t)3;
            n_total = (more_blocks ? nb_p1 : p1) - n_pa;
            n_c = 0;
        }
        int ra[RPT], re[RPT];      // raw rowptr entries of the next row block, consumed at the switch
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tid + q * TPB;
            const int rc = r < n_nr ? r : n_nr - 1;
            ra[q] = rowptr[n_r0 + rc];
            re[q] = rowptr[n_r0 + rc + 1];
        }
        int nn_p0, nn_p1;
        {
            const int64_t nnb = nb + gstride;
            const int64_t nnbc = nnb < nblk ? nnb : b;
            const int64_t q0 = nnbc * R;
            const int qn = (int)((nrows - q0) < R ? (nrows - q0) : R);
            nn_p0 = rowptr[q0 + vzero];
            nn_p1 = rowptr[q0 + qn + vzero];
        }
        {
            const int64_t rem = n_total - n_c;
            load_chunk_bf<TPB, U, NT>(nxt, colval, nzval, n_pa + n_c, (int)(rem < CHUNK ? rem : CHUNK), tid);
        }
        // 3. products -> LDS -> per-row sequential sums
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e0 = (u * TPB + tid) * 4;
            double2 pa2, pb2;
            pa2.x = cur.va[u].x * xg[u][0];
            pa2.y = cur.va[u].y * xg[u][1];
            pb2.x = cur.vb[u].x * xg[u][2];
            pb2.y = cur.vb[u].y * xg[u][3];
            *reinterpret_cast<double2 *>(&s_prod[e0]) = pa2;
            *reinterpret_cast<double2 *>(&s_prod[e0 + 2]) = pb2;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int a = (lo[q] > c ? lo[q] : (int)c) - (int)c;
            const int e = (hi[q] < c + n ? hi[q] : (int)(c + n)) - (int)c;
            // first PEEL entries: independent LDS reads, predicated adds (order and rounding unchanged)
            double pv[PEEL];
#pragma unroll
            for (int t = 0; t < PEEL; ++t) {
                int idx = a + t;
                idx = idx < CHUNK ? idx : CHUNK - 1;
                pv[t] = s_prod[idx];
            }
#pragma unroll
            for (int t = 0; t < PEEL; ++t) acc[q] = (a + t < e) ? acc[q] + pv[t] : acc[q];
            for (int j = a + PEEL; j < e; ++j) acc[q] += s_prod[j];
        }
        __syncthreads();

        if (last_chunk) {
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + q * TPB;
                if (r < nr) y[r0 + r] = acc[q];
                acc[q] = 0.0;
            }
            if (!more_blocks) break;
            b = nb; r0 = n_r0; nr = n_nr; pa = n_pa; total = n_total; c = 0; p0 = nb_p0; p1 = nb_p1;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + q * TPB;
                lo[q] = r < nr ? (int)(ra[q] - pa) : 0;
                hi[q] = r < nr ? (int)(re[q] - pa) : 0;
            }
            nb = b + gstride;
            nb_p0 = __builtin_amdgcn_readfirstlane(nn_p0);
            nb_p1 = __builtin_amdgcn_readfirstlane(nn_p1);
        } else {
            c += CHUNK;
        }
        cur = nxt;
    }
}

// ---- wave-independent kernel: every 64-lane wave owns its own rows and LDS slice, no workgroup
// barrier (waves of a workgroup drift apart freely, which hides more latency) ---------------------------
template <int RPL, int QPL>
__global__ __launch_bounds__(256) void k_wave(const int *__restrict__ rowptr,
                                              const int *__restrict__ colval,
                                              const double *__restrict__ nzval,
                                              const double *__restrict__ x, double *__restrict__ y,
                                              int64_t nrows, int64_t nnz)
{
    constexpr int R = 64 * RPL;            // rows per wave
    constexpr int CHUNK = 64 * 4 * QPL;    // entries per pass per wave
    __shared__ double s_all[4 * CHUNK];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *s_prod = s_all + w * CHUNK;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + w;
    const int64_t r0 = wave_id * R;
    if (r0 >= nrows) return;
    const int nr = (int)((nrows - r0) < R ? (nrows - r0) : R);
    const int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    const int64_t pa = p0 & ~(int64_t)3;
    const int64_t total = p1 - pa;
    int lo[RPL], hi[RPL];
    double acc[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int r = lane + q * 64;
        lo[q] = hi[q] = 0;
        acc[q] = 0.0;
        if (r < nr) { lo[q] = (int)(rowptr[r0 + r] - pa); hi[q] = (int)(rowptr[r0 + r + 1] - pa); }
    }
    for (int64_t c = 0; c < total; c += CHUNK) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        v4i col[QPL];
        v2d va[QPL], vb[QPL];
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            const int e0 = (u * 64 + lane) * 4;
            const int64_t g = pa + c + e0;
            col[u] = (v4i)(0);
            va[u] = (v2d)(0.0);
            vb[u] = (v2d)(0.0);
            if (e0 < n) {
                if (g + 3 < nnz) {
                    col[u] = *reinterpret_cast<const v4i *>(colval + g);
                    va[u] = *reinterpret_cast<const v2d *>(nzval + g);
                    vb[u] = *reinterpret_cast<const v2d *>(nzval + g + 2);
                } else {
                    if (g + 0 < nnz) { col[u].x = colval[g + 0]; va[u].x = nzval[g + 0]; }
                    if (g + 1 < nnz) { col[u].y = colval[g + 1]; va[u].y = nzval[g + 1]; }
                    if (g + 2 < nnz) { col[u].z = colval[g + 2]; vb[u].x = nzval[g + 2]; }
                }
            }
        }
        double xv[QPL][4];
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            const int e0 = (u * 64 + lane) * 4;
            if (e0 < n) {
                xv[u][0] = x[col[u].x];
                xv[u][1] = x[col[u].y];
                xv[u][2] = x[col[u].z];
                xv[u][3] = x[col[u].w];
            }
        }
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            const int e0 = (u * 64 + lane) * 4;
            if (e0 < n) {
                double2 pa2, pb2;
                pa2.x = va[u].x * xv[u][0];
                pa2.y = va[u].y * xv[u][1];
                pb2.x = vb[u].x * xv[u][2];
                pb2.y = vb[u].y * xv[u][3];
                *reinterpret_cast<double2 *>(&s_prod[e0]) = pa2;
                *reinterpret_cast<double2 *>(&s_prod[e0 + 2]) = pb2;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            const int a = lo[q] > c ? lo[q] : (int)c;
            const int e = hi[q] < c + n ? hi[q] : (int)(c + n);
            for (int j = a; j < e; ++j) acc[q] += s_prod[j - c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int r = lane + q * 64;
        if (r < nr) y[r0 + r] = acc[q];
    }
}

// ---- packed prototype: 16-bit block-relative columns + 8-bit value codes (dictionary in LDS) ----------
// Same row-block stream structure; a lane handles one aligned octet (8 entries): 16 B of columns +
// 8 B of codes instead of 96 B.  Products are dict[code] * x[r0 + dcol] -- identical values, so the
// result is bit-identical to the plain CSR kernel.
typedef short v8s __attribute__((ext_vector_type(8)));
template <int OCT>   // octets per lane per pass
__global__ __launch_bounds__(256) void k_packed(const int *__restrict__ rowptr,
                                                const short *__restrict__ dcol,
                                                const unsigned char *__restrict__ code,
                                                const double *__restrict__ dict, int ndict,
                                                const double *__restrict__ x, double *__restrict__ y,
                                                int64_t nrows, int64_t nnz)
{
    constexpr int CHUNK = 256 * 8 * OCT;
    __shared__ double s_prod[CHUNK];
    __shared__ double s_dict[256];
    const int tid = threadIdx.x;
    if (tid < ndict) s_dict[tid] = dict[tid];
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int nr = (int)((nrows - r0) < 256 ? (nrows - r0) : 256);
    const int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    const int64_t pa = p0 & ~(int64_t)7;
    const int64_t total = p1 - pa;
    int lo = 0, hi = 0;
    if (tid < nr) { lo = (int)(rowptr[r0 + tid] - pa); hi = (int)(rowptr[r0 + tid + 1] - pa); }
    __syncthreads();
    double acc = 0.0;
    for (int64_t c = 0; c < total; c += CHUNK) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        v8s dc[OCT];
        unsigned long long cd[OCT];
#pragma unroll
        for (int u = 0; u < OCT; ++u) {
            const int e0 = (u * 256 + tid) * 8;
            const int64_t g = pa + c + e0;
            dc[u] = (v8s)(0);
            cd[u] = 0;
            if (e0 < n) {      // arrays are padded to a multiple of 8 entries by the packer
                dc[u] = *reinterpret_cast<const v8s *>(dcol + g);
                cd[u] = *reinterpret_cast<const unsigned long long *>(code + g);
            }
        }
        double xv[OCT][8];
#pragma unroll
        for (int u = 0; u < OCT; ++u) {
            const int e0 = (u * 256 + tid) * 8;
            if (e0 < n) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // octets shared with the neighbouring row blocks hold deltas relative to THEIR base:
                    // clamp the index (their products are never summed here)
                    int64_t idx = r0 + (int)dc[u][k];
                    idx = idx < 0 ? 0 : (idx >= nrows ? nrows - 1 : idx);
                    xv[u][k] = x[idx];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < OCT; ++u) {
            const int e0 = (u * 256 + tid) * 8;
            if (e0 < n) {
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    double2 pr;
                    pr.x = s_dict[(cd[u] >> (8 * k)) & 0xff] * xv[u][k];
                    pr.y = s_dict[(cd[u] >> (8 * k + 8)) & 0xff] * xv[u][k + 1];
                    *reinterpret_cast<double2 *>(&s_prod[e0 + k]) = pr;
                }
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
    if (tid < nr) y[r0 + tid] = acc;
}

// ---- packed variants: threads per block / rows per thread -----------------------------------------------
template <int TPB, int RPT>
__global__ __launch_bounds__(TPB) void k_packed2(const int *__restrict__ rowptr,
                                                 const short *__restrict__ dcol,
                                                 const unsigned char *__restrict__ code,
                                                 const double *__restrict__ dict, int ndict,
                                                 const double *__restrict__ x, double *__restrict__ y,
                                                 int64_t nrows, int64_t nnz)
{
    constexpr int R = 256 * ((TPB * RPT + 255) / 256);      // rows per block: multiple of the 256-row packing base
    static_assert(TPB * RPT == R, "rows per block must be a multiple of 256");
    constexpr int CHUNK = TPB * 8;
    __shared__ double s_prod[CHUNK];
    __shared__ double s_dict[256];
    const int tid = threadIdx.x;
    for (int i = tid; i < ndict; i += TPB) s_dict[i] = dict[i];
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nr = (int)((nrows - r0) < R ? (nrows - r0) : R);
    const int64_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    const int64_t pa = p0 & ~(int64_t)7;
    const int64_t total = p1 - pa;
    int lo[RPT], hi[RPT];
    double acc[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        lo[q] = hi[q] = 0;
        acc[q] = 0.0;
        if (r < nr) { lo[q] = (int)(rowptr[r0 + r] - pa); hi[q] = (int)(rowptr[r0 + r + 1] - pa); }
    }
    __syncthreads();
    for (int64_t c = 0; c < total; c += CHUNK) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        const int e0 = tid * 8;
        if (e0 < n) {
            const int64_t g = pa + c + e0;
            const v8s dc = *reinterpret_cast<const v8s *>(dcol + g);
            const unsigned long long cd = *reinterpret_cast<const unsigned long long *>(code + g);
            // the packing base of entry g is the 256-row block of its row; all entries of this pass
            // belong to rows r0 .. r0+R-1, whose bases are r0 + 256*k: recover k from the row? not
            // available per entry -> prototype only valid for R == 256 bases; for R > 256 the packer
            // below uses base = R-row block instead (see tune_spmv.py).
            double xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int64_t idx = r0 + (int)dc[k];
                idx = idx < 0 ? 0 : (idx >= nrows ? nrows - 1 : idx);
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
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int a = lo[q] > c ? lo[q] : (int)c;
            const int e = hi[q] < c + n ? hi[q] : (int)(c + n);
            for (int j = a; j < e; ++j) acc[q] += s_prod[j - c];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + q * TPB;
        if (r < nr) y[r0 + r] = acc[q];
    }
}

extern "C" __attribute__((visibility("default"))) int hpcla_tune_spmv(
    int variant, const int *rowptr, const int *colval, const double *nzval, const double *x, double *y,
    int64_t nrows, int64_t nnz, void *stream, const int *bptr, const short *dcol, const unsigned char *code,
    const double *dict, int ndict, const short *dcol512, const short *dcol1024)
{
    hipStream_t s = (hipStream_t)stream;
#define NARROW(TPB, RPT, UN, NT, XCD, ABL)                                                        \
    {                                                                                             \
        uint32_t nb = (uint32_t)((nrows + TPB * RPT - 1) / (TPB * RPT));                          \
        k_narrow<TPB, RPT, UN, NT, XCD, ABL><<<nb, TPB, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nb); \
    }
#define WIDE(TPB, RPT, U, NT, XCD)                                                                \
    {                                                                                             \
        uint32_t nb = (uint32_t)((nrows + TPB * RPT - 1) / (TPB * RPT));                          \
        k_wide<TPB, RPT, U, NT, XCD><<<nb, TPB, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, nb); \
    }
    switch (variant) {
        case 0: NARROW(256, 1, 8, true, true, 0) break;    // production structure
        case 1: NARROW(256, 1, 8, false, true, 0) break;   // no nontemporal
        case 2: NARROW(256, 1, 8, true, false, 0) break;   // no XCD slice mapping
        case 3: NARROW(256, 2, 12, true, true, 0) break;   // 512 rows / block
        case 4: NARROW(256, 1, 8, true, true, 1) break;    // ablation: no LDS round trip
        case 5: NARROW(256, 1, 8, true, true, 2) break;    // ablation: no LDS, no gather
        case 6: NARROW(512, 1, 6, true, true, 0) break;    // 512 threads, 512 rows
        case 7: NARROW(128, 1, 8, true, true, 0) break;    // 128 threads, 128 rows
        case 10: WIDE(256, 1, 2, true, true) break;        // 16-byte loads, 256 rows, 2048 chunk
        case 11: WIDE(256, 2, 3, true, true) break;        // 16-byte loads, 512 rows, 3072 chunk
        case 12: WIDE(256, 1, 2, false, true) break;
        case 13: WIDE(256, 4, 6, true, true) break;        // 1024 rows, 6144 chunk (48 KiB LDS)
        case 14: WIDE(128, 2, 3, true, true) break;        // 128 threads, 256 rows
        case 15: WIDE(256, 2, 3, true, false) break;
        case 16: WIDE(256, 1, 2, false, false) break;      // wide, no NT, no XCD
        case 17: NARROW(256, 1, 8, false, false, 0) break; // narrow, no NT, no XCD
        case 18: WIDE(256, 1, 1, false, false) break;      // wide, 1024 chunk (8 KiB LDS)
        case 19: WIDE(256, 2, 3, false, false) break;      // wide, 512 rows, no NT, no XCD
        case 21: WIDE(256, 1, 2, true, 0) break;       // wide, NT stream loads, natural order (3-D: does x survive in L2 beside a streaming-hinted A?)
#define PIPE(TPB, RPT, U, NT, PERCU)                                                              \
    {                                                                                             \
        int64_t nb = (nrows + TPB * RPT - 1) / (TPB * RPT);                                       \
        int64_t g = nb < 256 * PERCU ? nb : 256 * PERCU;                                          \
        k_pipe<TPB, RPT, U, NT><<<(uint32_t)g, TPB, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, nb); \
    }
        case 30: PIPE(256, 1, 2, false, 4) break;
        case 31: PIPE(256, 1, 2, false, 6) break;
        case 32: PIPE(256, 1, 2, false, 8) break;
        case 33: PIPE(256, 1, 2, true, 6) break;
        case 34: PIPE(256, 2, 3, false, 4) break;
        case 35: PIPE(256, 2, 3, false, 6) break;
        case 36: PIPE(256, 1, 2, false, 5) break;
        case 37: PIPE(512, 1, 2, false, 3) break;
#define PIPE2(TPB, RPT, U, NT, PERCU, PEEL)                                                            \
    {                                                                                             \
        int64_t nb = (nrows + TPB * RPT - 1) / (TPB * RPT);                                       \
        int64_t g = nb < 256 * PERCU ? nb : 256 * PERCU;                                          \
        k_pipe2<TPB, RPT, U, NT, PEEL><<<(uint32_t)g, TPB, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, nb); \
    }
        case 40: PIPE2(256, 1, 2, false, 4, 8) break;
        case 41: PIPE2(256, 1, 2, false, 5, 8) break;
        case 42: PIPE2(256, 1, 2, false, 8, 8) break;
        case 43: PIPE2(256, 1, 2, true, 4, 8) break;
        case 44: PIPE2(256, 2, 3, false, 3, 8) break;
        case 45: PIPE2(256, 2, 3, false, 4, 8) break;
        case 46: PIPE2(256, 1, 1, false, 6, 8) break;
        case 47: PIPE2(256, 1, 1, false, 8, 8) break;
        case 48: PIPE2(512, 1, 2, false, 2, 8) break;
        case 49: PIPE2(256, 1, 2, false, 16, 8) break;
        case 50: PIPE2(256, 1, 2, false, 4, 1) break;
        case 51: PIPE2(256, 1, 2, false, 5, 1) break;
        case 52: PIPE2(256, 1, 2, false, 6, 8) break;
#define WAVE(RPL, QPL)                                                                            \
    {                                                                                             \
        int64_t rows_per_block = 4 * 64 * RPL;                                                    \
        uint32_t nb = (uint32_t)((nrows + rows_per_block - 1) / rows_per_block);                  \
        k_wave<RPL, QPL><<<nb, 256, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz);             \
    }
        case 60: WAVE(1, 2) break;     // 64 rows / wave, 512-entry chunk
        case 61: WAVE(2, 2) break;     // 128 rows / wave
        case 62: WAVE(2, 3) break;     // 128 rows / wave, 768-entry chunk
        case 63: WAVE(4, 2) break;     // 256 rows / wave
        case 64: WAVE(4, 3) break;
        case 65: WAVE(1, 1) break;
        case 66: WAVE(2, 4) break;
        case 70: {
            uint32_t nb = (uint32_t)((nrows + 255) / 256);
            k_wide_bptr<256, 1, 2, false, false><<<nb, 256, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, nb, bptr);
        } break;
        case 80: k_packed<1><<<(uint32_t)((nrows + 255) / 256), 256, 0, s>>>(rowptr, dcol, code, dict, ndict, x, y, nrows, nnz); break;
        case 81: k_packed<2><<<(uint32_t)((nrows + 255) / 256), 256, 0, s>>>(rowptr, dcol, code, dict, ndict, x, y, nrows, nnz); break;
        case 82: k_packed2<256, 2><<<(uint32_t)((nrows + 511) / 512), 256, 0, s>>>(rowptr, dcol512, code, dict, ndict, x, y, nrows, nnz); break;
        case 83: k_packed2<512, 1><<<(uint32_t)((nrows + 511) / 512), 512, 0, s>>>(rowptr, dcol512, code, dict, ndict, x, y, nrows, nnz); break;
        case 84: k_packed2<256, 4><<<(uint32_t)((nrows + 1023) / 1024), 256, 0, s>>>(rowptr, dcol1024, code, dict, ndict, x, y, nrows, nnz); break;
        case 85: k_packed2<512, 2><<<(uint32_t)((nrows + 1023) / 1024), 512, 0, s>>>(rowptr, dcol1024, code, dict, ndict, x, y, nrows, nnz); break;
        case 86: k_packed2<1024, 1><<<(uint32_t)((nrows + 1023) / 1024), 1024, 0, s>>>(rowptr, dcol1024, code, dict, ndict, x, y, nrows, nnz); break;
#define ROWG(TPB, CH, UR, XCD)                                                                    \
    {                                                                                             \
        uint32_t nb = (uint32_t)((nrows + TPB - 1) / TPB);                                        \
        k_rowg<TPB, CH, UR, XCD><<<nb, TPB, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, nb); \
    }
        // ROWG: row-wise gathers from LDS-transposed A.  <threads = rows per block, entries per LDS pass, entries per gather step, XCD group>
        case 90: ROWG(256, 2048, 8, 0) break;              // 24 KiB LDS: 6 workgroups per CU
        case 91: ROWG(256, 1792, 8, 0) break;              // 21 KiB: 7 per CU; the 7-point block in one pass
        case 92: ROWG(256, 1344, 8, 0) break;              // 15.75 KiB: 8 per CU; the 7-point block in two passes
        case 93: ROWG(128, 1024, 8, 0) break;              // 128-row blocks, 12 KiB: 13 workgroups = 26 waves per CU
        case 94: ROWG(256, 1792, 4, 0) break;              // four entries per gather step
        case 95: ROWG(256, 1792, 8, 32) break;             // + XCD groups of 32 / 64 row blocks
        case 96: ROWG(256, 1792, 8, 64) break;
        case 97: ROWG(128, 1024, 8, 64) break;
        case 98: ROWG(128, 1024, 8, 128) break;
        case 99: ROWG(512, 3584, 8, 0) break;              // 512-row blocks, 42 KiB: 3 workgroups = 24 waves per CU
#define ROWGW(WPB, CHW, UR, XCD)                                                                  \
    {                                                                                             \
        uint32_t nb = (uint32_t)((nrows + 64 * WPB - 1) / (64 * WPB));                            \
        k_rowg_wave<WPB, CHW, UR, XCD><<<nb, 64 * WPB, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, nb); \
    }
        case 120: ROWGW(4, 512, 8, 0) break;               // wave-private tiles: 256-row workgroups, 24 KiB, no barrier
        case 121: ROWGW(2, 512, 8, 0) break;               // 128-row workgroups
        case 122: ROWGW(4, 512, 8, 32) break;              // 256-row workgroups in XCD groups of 32 / 64
        case 123: ROWGW(4, 512, 8, 64) break;
        case 124: ROWGW(1, 512, 8, 0) break;               // one wave per workgroup
        case 125: ROWGW(4, 256, 8, 0) break;               // 12 KiB per workgroup (a 7-point wave takes two passes)
        case 126: ROWGW(8, 512, 8, 0) break;               // 512-row workgroups
#define ROWGWX(WPB, CHW, UR, XCD, DMA, NTA)                                                       \
    {                                                                                             \
        uint32_t nb = (uint32_t)((nrows + 64 * WPB - 1) / (64 * WPB));                            \
        k_rowg_wave<WPB, CHW, UR, XCD, DMA, NTA><<<nb, 64 * WPB, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, nb); \
    }
        case 130: ROWGWX(4, 512, 8, 64, true, false) break;    // A entries by LDS-DMA (no VGPR round trip, no ds_write), groups of 64
        case 131: ROWGWX(4, 512, 8, 32, true, false) break;
        case 132: ROWGWX(4, 512, 8, 0, true, false) break;
        case 133: ROWGWX(2, 512, 8, 128, true, false) break;   // ... in 128-row workgroups
        case 134: ROWGWX(4, 512, 8, 64, false, true) break;    // A entries with non-temporal loads
        case 135: ROWGWX(4, 512, 8, 32, false, true) break;
        case 138: k_rowg_wave<4, 512, 8, 64, false, false, true><<<(uint32_t)((nrows + 255) / 256), 256, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, (uint32_t)((nrows + 255) / 256)); break;   // y stored non-temporally
        case 139: k_rowg_wave<4, 512, 8, 32, false, false, true><<<(uint32_t)((nrows + 255) / 256), 256, 0, s>>>(rowptr, colval, nzval, x, y, nrows, nnz, (uint32_t)((nrows + 255) / 256)); break;
        case 136: ROWGWX(4, 512, 4, 64, false, false) break;   // four entries per gather step
        case 137: ROWGWX(4, 512, 4, 32, false, false) break;
        case 110: ROWG(128, 896, 8, 128) break;            // 128 rows x 7 entries exactly: 10.5 KiB, 15 workgroups = 30 waves per CU
        case 111: ROWG(128, 1024, 8, 256) break;
        case 112: ROWG(64, 512, 8, 128) break;             // one wave per workgroup (the barrier is free)
        case 113: ROWG(64, 512, 8, 256) break;
        case 114: ROWG(128, 1024, 4, 128) break;
        case 115: ROWG(128, 1024, 8, 32) break;
        case 116: ROWG(64, 512, 8, 512) break;
        case 117: ROWG(128, 768, 8, 128) break;            // 9 KiB: 16 workgroups = 32 waves per CU (7-point block in two passes)
        case 22: WIDE(256, 1, 2, false, 4) break;          // natural kernel, XCD groups of 4 / 8 / 16 / 32 / 64 / 128 blocks
        case 23: WIDE(256, 1, 2, false, 8) break;
        case 24: WIDE(256, 1, 2, false, 16) break;
        case 25: WIDE(256, 1, 2, false, 32) break;
        case 26: WIDE(256, 1, 2, false, 64) break;
        case 27: WIDE(256, 1, 2, false, 128) break;
        case 28: WIDE(256, 1, 2, true, 16) break;          // groups + NT stream loads
        case 29: WIDE(256, 1, 2, true, 64) break;
        case 20:
            k_copy<<<256 * 16, 256, 0, s>>>((const v4i *)colval, (const v2d *)nzval, (const v4i *)rowptr,
                                            (double2 *)y, (const double2 *)x, nnz, nrows);
            break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
