// spmv.hip -- CSR SpMV for MI355X (gfx950), fp64 values, int32/int64 indices.
//
// Replaces the reference's one-work-item-per-row KernelAbstractions kernel
// (_spmv_kernel!, src/sparse.jl:2055-2066), whose lanes read nzval/colval with a stride of one
// row length (uncoalesced).  Design ("row-block stream", row-gather form since round 4):
//
//   * a 256-thread workgroup owns RPB = 256 consecutive rows, each WAVE 64 of them; the nonzeros of a wave's
//     rows are one contiguous range of colval/nzval, which the wave streams into its own slice of LDS with
//     coalesced 16-byte loads (spmv_rowgather_kernel below);
//   * lane t then walks ITS OWN row out of LDS: gather instruction j of a wave reads the j-th entry of 64
//     consecutive rows -- for a banded / stencil matrix one x stream of 512 contiguous bytes;
//   * the row sum is the reference's multiply-add chain in stored order.
//
// The sum therefore runs in exactly the reference's order with a separately rounded multiply and
// add (this file is compiled with -ffp-contract=off): results are bit-identical to the reference
// loop, not merely within tolerance.  No MFMA: 2 flop per 12 bytes is a bandwidth-bound gather.
//
// HBM traffic per row block = the algorithmic bytes: 12 B/nnz (int32) + 4 B/row rowptr + 8 B/row y
// (+ x once).  Rows longer than a pass are handled by the pass loop (the row's running sum is
// carried in a register), so there is no row-length limit and no preprocessing.
//
// Measured choices (benchmarks/tune_spmv.py, tune_spmv_lib.py; profiles/): 16-byte loads beat
// element-per-lane loads; plain loads beat nontemporal ones (again in round 3: +10 ... +14 %); one
// moving window over the matrix beats one slice per XCD (DRAM pages and the x window stay hot for all 8
// XCDs through the 256 MiB Infinity Cache) -- WITHIN that window the row blocks are dealt to the XCDs in
// groups whose size is measured per matrix at plan time (xcd_group_index, hpcla_spmv_tune_block_order_*);
// a software-pipelined persistent variant loses to plain high occupancy.
// Rounds 1-3 shipped a product-parking "quad" kernel (a lane owned four consecutive stored entries and parked
// their products in LDS); it lost to the row gather on every measured matrix from round 4 on (-1.5 ... -4.8 %,
// profiles/r04_spmv_rowg.log) and was retired in round 6 with its switch (hpcla_set_spmv_kernel,
// HPCLA_SPMV_KERNEL); the history of its tuning is in profiles/MEASUREMENTS_r01_r02.md / _r03.md.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "common.h"
#include "halo_wait.h"

namespace hpcla {

constexpr int RPB = 256;      // rows per block == threads per block
constexpr int CHUNK = RPB * 8;           // unaligned fallback kernel: 2048 products parked in LDS per pass (16 KiB)
constexpr int UNROLL = CHUNK / RPB;      // ... entries per lane per pass

template <typename T, int N>
using vec = T __attribute__((ext_vector_type(N)));

template <bool SPLIT>
__device__ __forceinline__ double gather_x(const double *__restrict__ x_own,
                                           const double *__restrict__ x_ghost, int64_t n_own,
                                           int64_t col)
{
    if (SPLIT) {
        const double *p = col < n_own ? x_own + col : x_ghost + (col - n_own);
        return *p;
    }
    return x_own[col];
}

// Which row block a workgroup owns.  Plain launches: list[b] or base + b.  Fused distributed launches
// (WAIT, peer-window transport): the first n_first workgroups take the INTERIOR blocks and never wait;
// the rest take the BOUNDARY blocks -- dispatched last, so the neighbours' pushes have long landed --
// and poll the plan's flag lines once before their first ghost gather (halo_wait.h).
struct BlockSel {
    const int32_t *list;         // blocks of this launch (plain) / boundary blocks (WAIT); null: base + b
    int64_t base;
    const int32_t *first_list;   // WAIT only: interior blocks; null: first_base + b
    int64_t first_base;
    int64_t n_first;
    int64_t n_run;               // length of the contiguous run the group order applies to (plain: the launch; WAIT: n_first)
    int32_t group_log2;          // > 0: XCD-grouped order of the contiguous run, groups of 2^group_log2 row blocks
    int32_t nt_y;                // != 0: y is stored non-temporally (row-gather kernel; see spmv_nt_y below)
};

// XCD-grouped order of a contiguous run of n row blocks.  The hardware deals workgroups to the 8 XCDs round-robin by
// workgroup id, so in the natural order block b runs on XCD b % 8 and the x lines two blocks shared are fetched into
// two L2s unless their distance is a multiple of 8 (a 4096-wide 2-D grid: +-16 blocks -- fine; config 4's 512 x 512
// planes: +-2 blocks -- every x line came into ~4 L2s, 1.26 x the algorithmic reads).  Here every XCD takes GROUPS of
// G consecutive blocks, groups dealt round-robin: neighbours at distance << G share an L2, distances that are
// multiples of 8 G still do, and the launch still walks ONE moving window through the matrix (the property that made
// the natural order beat one slice per XCD in round 1).  A bijection on [0, n): the ragged tail keeps its order.
// Chosen per matrix at plan time (hpcla_spmv_block_order_hint); profiles/r03_spmv_xcd_group_order.log.
constexpr int NUM_XCD_LOG2 = 3;
__device__ __forceinline__ int64_t xcd_group_index(int64_t b, int64_t n, int group_log2)
{
    const int64_t span = (int64_t)1 << (NUM_XCD_LOG2 + group_log2);
    if (b >= n - (n & (span - 1))) return b;
    const int64_t xcd = b & ((1 << NUM_XCD_LOG2) - 1), q = b >> NUM_XCD_LOG2;
    return ((((q >> group_log2) << NUM_XCD_LOG2) + xcd) << group_log2) + (q & (((int64_t)1 << group_log2) - 1));
}

template <bool WAIT>
__device__ __forceinline__ int64_t select_block(const BlockSel &bs, bool &wait, int lead)
{
    const int64_t b = (int64_t)blockIdx.x - lead;     // `lead` push workgroups come first in a fused launch
    wait = false;
    if (WAIT) {
        if (b < bs.n_first) {
            if (bs.first_list) return (int64_t)bs.first_list[b];
            return bs.first_base + (bs.group_log2 > 0 ? xcd_group_index(b, bs.n_run, bs.group_log2) : b);
        }
        wait = true;
        return bs.list ? (int64_t)bs.list[b - bs.n_first] : bs.base + (b - bs.n_first);
    }
    if (bs.list) return (int64_t)bs.list[b];
    return bs.base + (bs.group_log2 > 0 ? xcd_group_index(b, bs.n_run, bs.group_log2) : b);
}

// Fused x.y epilogue (CG's p.Ap): every workgroup leaves the
// deterministic tree sum of x[r]*y[r] over its rows in dot_partial[row block]; hpcla_spmv_dot_finish
// sums the partials in index order.  Valid when x is partitioned like A's rows (x_own[r] is x at row r).
__device__ __forceinline__ void block_dot_epilogue(double *s_scratch, double *__restrict__ dot_partial,
                                                   int64_t blk, double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_scratch[w] = v;       // s_prod is free: the chunk loop ended with a barrier
    __syncthreads();
    if (threadIdx.x == 0) dot_partial[blk] = ((s_scratch[0] + s_scratch[1]) + s_scratch[2]) + s_scratch[3];
}

// ---- round 4: gathers issued BY ROW ("row gather"), from a wave-private LDS copy of the A entries -------------------
// Counters of the product-parking quad kernel of rounds 1-3 on the 7-point matrix against the 5-point one
// (profiles/r04_spmv_2d_vs_3d_counters.txt): per stored entry 6.2 x the L1 tag-conflict stall cycles
// (TCP_READ_TAGCONFLICT_STALL_CYCLES: 21 % of the launch per TCP against 3.5 %), 1.6 x the cycles the address unit waits for
// the cache, 1.6 x the issue stalls -- while LDS conflicts, TLB misses, L2 misses and HBM requests per entry are equal
// or lower.  Cause: a lane owns a QUAD of consecutive stored entries, so one gather instruction covers entries 4 L + q,
// L = 0..63: ~37 rows x all 7 column types = seven x streams in ONE instruction, five of them aligned to the same power
// of two (i - 2 MiB, i - 4 KiB, i, i + 4 KiB, i + 2 MiB) -- ~24 lines per instruction, most on one tag bank.
// Here the A entries are streamed into LDS UNMULTIPLIED (coalesced 16-byte loads as before; 12 B per entry) and lane t
// then walks ITS OWN row: gather instruction j of a wave reads the j-th entry of 64 consecutive rows -- for any banded /
// stencil matrix ONE x stream, 512 contiguous bytes, 4-5 lines; for unstructured rows no worse than before.  The row sum
// is the same multiply-add chain in stored order: the reference's bits (src/sparse.jl:2055-2066).
// WAVE-PRIVATE: every wave owns 64 of the block's 256 rows and its own slice of LDS; LDS operations of one wave complete
// in order, so the kernel has no workgroup barrier at all (the dot epilogue and the halo wait keep theirs) -- and the
// 256-row block of the callers' block lists, of the p.Ap partials and of the packed copy stays what it was.
// Harness (benchmarks/tune/spmv_variants.hip k_rowg / k_rowg_wave, profiles/r04_spmv_rowg.log): 7-point slab -4.3 ... -4.8 %,
// 4096^2 -1.5 %, 8192^2 -2.2 % against the quad kernel under its measured block order.
constexpr int RG_CHW = 464;                         // entries per wave and pass: 64 rows x 7 + the <= 3 entries in front of the aligned start, rounded up (5.4 KiB of LDS per wave: 7 workgroups per CU)
constexpr int RG_NQ = (RG_CHW / 4 + 63) / 64;       // quads per lane per pass
constexpr int RG_UR = 8;                            // entries per gather step

// LONGR (round 5, OPT-IN: hpcla_spmv_longrows_*): rows of at least `long_min` entries are LEFT OUT here -- their lanes sum
// nothing and store nothing, and a pass that lies wholly inside such a row is skipped (the wave jumps to the pass that holds
// the row's end) -- and are summed by spmv_longrow_partial/final_kernel below in TREE order instead.  The default
// instantiations (LONGR = false) are the code they were: every row, however long, is one lane's sequential sum in stored
// order -- the reference's bits, and its cliff (one work-item per row, src/sparse.jl:2055-2066).
template <typename I, bool SPLIT, bool WAIT, bool LONGR = false>
__global__ __launch_bounds__(RPB) void spmv_rowgather_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ x_own, const double *x_ghost, int64_t n_own,
    double *__restrict__ y, int64_t nrows, int64_t nnz, int base,
    BlockSel bs, double *__restrict__ dot_partial, HaloWait hw, PushArgs push, int64_t long_min = 0)
{
    static_assert(!LONGR || !WAIT, "the long-row form is a plain launch");
    __shared__ __attribute__((aligned(16))) I s_col_all[(RPB / 64) * RG_CHW];
    __shared__ __attribute__((aligned(16))) double s_val_all[(RPB / 64) * RG_CHW];
    __shared__ double s_red[RPB / 64];

    if (WAIT && (int)blockIdx.x < push.n_blocks) {
        halo_push_block<I, RPB>(push, (int)blockIdx.x);
        return;
    }
    const int tid = threadIdx.x;
    bool wait_ghosts;
    const int64_t blk = select_block<WAIT>(bs, wait_ghosts, WAIT ? push.n_blocks : 0);
    uint32_t waited = 0;
    if (WAIT && wait_ghosts) {     // reader index: the plan's push workgroups come first (window.hip), then the boundary blocks
        waited = halo_wait_block(hw, hw.first_wait_reader + (uint32_t)(blockIdx.x - push.n_blocks - bs.n_first));
        x_ghost = hw.ghost0 + (int64_t)(waited & ~HALO_WAIT_TIMED_OUT) * hw.buf_stride;
    }
    const int64_t r0 = blk * RPB;
    const int nr = (int)((nrows - r0) < RPB ? (nrows - r0) : RPB);
    if (WAIT && (waited & HALO_WAIT_TIMED_OUT)) {
        // the neighbours' values never arrived (wait expired): this block's rows are POISONED, never computed from stale ghosts
        if ((int)threadIdx.x < nr) y[r0 + threadIdx.x] = halo_poison();
        if (dot_partial && threadIdx.x == 0) dot_partial[blk] = halo_poison();
        return;
    }

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    I *s_col = s_col_all + wave * RG_CHW;
    double *s_val = s_val_all + wave * RG_CHW;
    const int64_t rw = r0 + wave * 64;                                   // this wave's rows
    const int nrw = nr - wave * 64 < 0 ? 0 : (nr - wave * 64 > 64 ? 64 : nr - wave * 64);
    double acc = 0.0, x_row = 0.0;
    if (nrw > 0) {                                                       // wave-uniform
        const int64_t p0 = (int64_t)rowptr[rw] - base;
        const int64_t p1 = (int64_t)rowptr[rw + nrw] - base;
        const int64_t pa = p0 & ~(int64_t)3;                             // quad-aligned start (<= 3 entries of the rows before)
        const int64_t total = p1 - pa;
        // this lane's row bounds, raw and UNCONDITIONAL (lanes past the wave's last row read that row's pair: a predicated
        // load is a branch whose merge copies the loaded register, i.e. waits for it on the spot); first used behind the A stream
        const int ll = lane < nrw ? lane : nrw - 1;
        I rlo = rowptr[rw + ll], rhi = rowptr[rw + ll + 1];
        if (dot_partial) x_row = x_own[rw + ll];                         // the epilogue's x rides along with the stream
        const bool is_long = LONGR && lane < nrw && (int64_t)rhi - (int64_t)rlo >= long_min;
        for (int64_t c = 0; c < total; c += RG_CHW) {
            if (LONGR) {
                // a pass wholly inside a long row holds nothing for this kernel: jump to the pass that holds the row's end
                const int64_t lo_rel = (int64_t)rlo - base - pa, hi_rel = (int64_t)rhi - base - pa;
                const uint64_t covers = __ballot(is_long && lo_rel <= c && hi_rel >= c + RG_CHW);
                if (covers) {                                            // wave-uniform
                    const int64_t hi_l = __shfl(hi_rel, __ffsll((unsigned long long)covers) - 1, 64);
                    c = (hi_l / RG_CHW) * RG_CHW - RG_CHW;               // (>= c: the row reaches the end of this pass)
                    continue;
                }
            }
            const int n = (int)((total - c) < RG_CHW ? (total - c) : RG_CHW);
            if (pa + c + ((n + 3) & ~3) <= nnz) {
                // every quad of the pass lies inside the arrays: ALL of a lane's quads are requested before the first is
                // written (lanes past the end re-read the pass's last quad -- lines their neighbours read anyway -- and
                // write nothing)
                const int last = (n - 1) & ~3;
                vec<I, 4> cq[RG_NQ];
                vec<double, 2> va[RG_NQ], vb[RG_NQ];
#pragma unroll
                for (int u = 0; u < RG_NQ; ++u) {
                    const int e0 = (u * 64 + lane) * 4;
                    const int ee = e0 < last ? e0 : last;
                    cq[u] = *reinterpret_cast<const vec<I, 4> *>(colval + pa + c + ee);
                    va[u] = *reinterpret_cast<const vec<double, 2> *>(nzval + pa + c + ee);
                    vb[u] = *reinterpret_cast<const vec<double, 2> *>(nzval + pa + c + ee + 2);
                }
#pragma unroll
                for (int u = 0; u < RG_NQ; ++u) {
                    const int e0 = (u * 64 + lane) * 4;
                    if (e0 < n) {
                        *reinterpret_cast<vec<I, 4> *>(&s_col[e0]) = cq[u];
                        *reinterpret_cast<vec<double, 2> *>(&s_val[e0]) = va[u];
                        *reinterpret_cast<vec<double, 2> *>(&s_val[e0 + 2]) = vb[u];
                    }
                }
            } else {
                // the one pass of the launch that reaches past the end of the arrays: entry by entry
                for (int e = lane; e < n; e += 64) {
                    const int64_t g = pa + c + e;
                    s_col[e] = g < nnz ? colval[g] : (I)base;
                    s_val[e] = g < nnz ? nzval[g] : 0.0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // this wave's LDS writes, then its LDS reads (in order per wave)
            __builtin_amdgcn_wave_barrier();
            // the lane's row bounds are first USED here, behind the A stream: the empty asm makes them loop-variant, or the
            // compiler hoists the (loop-invariant) subtraction in front of the loop and with it a full memory round trip
            asm volatile("" : "+v"(rlo), "+v"(rhi));
            {
                // (64-bit until clamped: a row of > 2^31 entries may lie in front of or behind this pass)
                const int64_t lo64 = lane < nrw ? (int64_t)rlo - base - pa - c : 0;
                const int64_t hi64 = lane < nrw ? (int64_t)rhi - base - pa - c : 0;
                const int lo = lo64 < 0 ? 0 : (lo64 > n ? n : (int)lo64);
                const int hi = hi64 < 0 ? 0 : (hi64 > n ? n : (int)hi64);
                // ONE row owns the whole pass (a row of more than a pass's entries: round 5, the arrow matrix's dense row):
                // a lane walking it alone has eight gathers in flight and waits a memory round trip for each step (1.7 s for
                // a row of 16.7 M entries, benchmarks/bench_arrow.py).  The wave multiplies the pass out TOGETHER -- every
                // lane its share of the entries, the rounded products parked over the values in LDS -- and the owner then
                // adds them in stored order: the same products, the same additions in the same order, the same bits.
                const uint64_t whole = __ballot(lane < nrw && lo == 0 && hi == n && !(LONGR && is_long));
                if (whole) {                                             // wave-uniform
                    // (loops four wide at most: this branch must not cost the stencil path a register -- 61-64 VGPRs, 7 workgroups per
                    // CU; unrolled freely it took 96 and the headline lost 7 %, profiles/r05_arrow_and_long_rows.log)
#pragma unroll 4
                    for (int e = lane; e < n; e += 64)
                        s_val[e] = s_val[e] * gather_x<SPLIT>(x_own, x_ghost, n_own, (int64_t)(I)(s_col[e] - (I)base));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane == __ffsll((unsigned long long)whole) - 1) {
                        int j = 0;
#pragma unroll 1
                        for (; j + 4 <= n; j += 4) {
                            const double p0 = s_val[j], p1 = s_val[j + 1], p2 = s_val[j + 2], p3 = s_val[j + 3];
                            acc += p0; acc += p1; acc += p2; acc += p3;
                        }
#pragma unroll 1
                        for (; j < n; ++j) acc += s_val[j];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
                int j = lo;
                int e = hi;
                if (LONGR && is_long) e = j;                             // summed by the long-row kernels
                // RG_UR entries per step, each under its own lane predicate: the step's gathers leave together (a gather
                // none of the wave's lanes needs is skipped), the sums follow in stored order
                for (; j < e; j += RG_UR) {
                    int64_t cc[RG_UR];
                    double vv[RG_UR], xx[RG_UR];
#pragma unroll
                    for (int u = 0; u < RG_UR; ++u) {
                        cc[u] = 0; vv[u] = 0.0;
                        if (j + u < e) { cc[u] = (int64_t)(I)(s_col[j + u] - (I)base); vv[u] = s_val[j + u]; }
                    }
#pragma unroll
                    for (int u = 0; u < RG_UR; ++u) { xx[u] = 0.0; if (j + u < e) xx[u] = gather_x<SPLIT>(x_own, x_ghost, n_own, cc[u]); }
#pragma unroll
                    for (int u = 0; u < RG_UR; ++u) if (j + u < e) acc += vv[u] * xx[u];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // ... and the reads before the next pass's writes
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < nrw && !(LONGR && is_long)) {
            if (bs.nt_y) __builtin_nontemporal_store(acc, y + rw + lane);
            else y[rw + lane] = acc;
        } else if (LONGR && lane < nrw) {
            // a row left to the long-row kernels: NaN now, its tree sum over it later in stream order.  A qualifying row the
            // caller's list omits therefore reads NaN -- never the stale value of an earlier product
            y[rw + lane] = __builtin_nan("");
        }
    }
    if (dot_partial) block_dot_epilogue(s_red, dot_partial, blk, lane < nrw ? acc * x_row : 0.0);
}

// ---- OPT-IN long rows (round 5): y[r] for the listed rows, summed in TREE order ---------------------------------------
// north_star's "wavefront-level __shfl / segmented-scan row reductions", used where they are needed: a row of millions of
// entries (an arrow matrix's dense row) is one lane's sequential sum in the default kernel -- the reference's order and its
// cliff.  Here a long row is cut into LR_CHUNKS contiguous pieces (at least LR_MIN_CHUNK entries each), one workgroup per
// piece: every thread sums its strided share in four independent chains, then a shuffle tree per wave and the four waves
// in order; a second kernel adds the pieces' sums in order.  Deterministic, but NOT the reference's order: the result is
// within k u / (1 - k u) of |A||x| with k ~ len / 1024 + 20 instead of len -- tests hold it to 1e-12 (|A||x|)_r.
constexpr int LR_CHUNKS = 1024;
constexpr int64_t LR_MIN_CHUNK = 4096;

template <typename I, bool SPLIT>
__global__ __launch_bounds__(256) void spmv_longrow_partial_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ x_own, const double *x_ghost, int64_t n_own, int base,
    const int64_t *__restrict__ long_rows, double *__restrict__ partial)
{
    __shared__ double s_w[4];
    const int64_t r = long_rows[blockIdx.y];
    const int64_t p0 = (int64_t)rowptr[r] - base, p1 = (int64_t)rowptr[r + 1] - base, len = p1 - p0;
    int64_t chunk = (len + LR_CHUNKS - 1) / LR_CHUNKS;
    if (chunk < LR_MIN_CHUNK) chunk = LR_MIN_CHUNK;
    const int64_t s = p0 + (int64_t)blockIdx.x * chunk;
    const int64_t e = s + chunk < p1 ? s + chunk : p1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t i = s + threadIdx.x;
    for (; i + 768 < e; i += 1024) {
        const int64_t c0 = (int64_t)(I)(colval[i] - (I)base), c1 = (int64_t)(I)(colval[i + 256] - (I)base),
                      c2 = (int64_t)(I)(colval[i + 512] - (I)base), c3 = (int64_t)(I)(colval[i + 768] - (I)base);
        const double v0 = nzval[i], v1 = nzval[i + 256], v2 = nzval[i + 512], v3 = nzval[i + 768];
        a0 += v0 * gather_x<SPLIT>(x_own, x_ghost, n_own, c0);
        a1 += v1 * gather_x<SPLIT>(x_own, x_ghost, n_own, c1);
        a2 += v2 * gather_x<SPLIT>(x_own, x_ghost, n_own, c2);
        a3 += v3 * gather_x<SPLIT>(x_own, x_ghost, n_own, c3);
    }
    for (; i < e; i += 256) a0 += nzval[i] * gather_x<SPLIT>(x_own, x_ghost, n_own, (int64_t)(I)(colval[i] - (I)base));
    double v = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * LR_CHUNKS + blockIdx.x] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

__global__ __launch_bounds__(256) void spmv_longrow_final_kernel(const int64_t *__restrict__ long_rows,
                                                                 const double *__restrict__ partial, double *__restrict__ y)
{
    __shared__ double s_w[4];
    static_assert(LR_CHUNKS == 1024, "four pieces per thread");
    const double *p = partial + (int64_t)blockIdx.x * LR_CHUNKS + threadIdx.x * 4;
    double v = ((p[0] + p[1]) + p[2]) + p[3];              // contiguous pieces stay neighbours in the tree
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) y[long_rows[blockIdx.x]] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

// ---- fallback kernel: element-per-lane loads, no alignment requirement --------------------------------
template <typename I, bool SPLIT, bool WAIT>
__global__ __launch_bounds__(RPB) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void spmv_rowblock_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ x_own, const double *x_ghost, int64_t n_own,
    double *__restrict__ y, int64_t nrows, int base, BlockSel bs,
    double *__restrict__ dot_partial, HaloWait hw, PushArgs push)
{
    __shared__ double s_prod[CHUNK];

    if (WAIT && (int)blockIdx.x < push.n_blocks) {
        halo_push_block<I, RPB>(push, (int)blockIdx.x);
        return;
    }
    const int tid = threadIdx.x;
    bool wait_ghosts;
    const int64_t blk = select_block<WAIT>(bs, wait_ghosts, WAIT ? push.n_blocks : 0);
    // boundary workgroup of a fused distributed launch (workgroup-uniform branch): the ghosts may be read once
    // every neighbour has published this step; the buffer of the step's epoch is computed here, uniformly, so
    // the loop below is the same code as in the plain kernel
    uint32_t waited = 0;
    if (WAIT && wait_ghosts) {     // reader index: the plan's push workgroups come first (window.hip), then the boundary blocks
        waited = halo_wait_block(hw, hw.first_wait_reader + (uint32_t)(blockIdx.x - push.n_blocks - bs.n_first));
        x_ghost = hw.ghost0 + (int64_t)(waited & ~HALO_WAIT_TIMED_OUT) * hw.buf_stride;
    }
    const int64_t r0 = blk * RPB;
    const int nr = (int)((nrows - r0) < RPB ? (nrows - r0) : RPB);
    if (WAIT && (waited & HALO_WAIT_TIMED_OUT)) {
        // the neighbours' values never arrived (wait expired): this block's rows are POISONED, never computed from
        // stale ghosts (workgroup-uniform branch, taken before the row loop: no register held across it)
        if ((int)threadIdx.x < nr) y[r0 + threadIdx.x] = halo_poison();
        if (dot_partial && threadIdx.x == 0) dot_partial[blk] = halo_poison();
        return;
    }

    const int64_t p0 = (int64_t)rowptr[r0] - base;
    const int64_t p1 = (int64_t)rowptr[r0 + nr] - base;
    const int64_t total = p1 - p0;

    int64_t lo = 0, hi = 0;
    if (tid < nr) {
        lo = (int64_t)rowptr[r0 + tid] - base - p0;
        hi = (int64_t)rowptr[r0 + tid + 1] - base - p0;
    }

    double acc = 0.0;
    for (int64_t c = 0; c < total; c += CHUNK) {
        const int n = (int)((total - c) < CHUNK ? (total - c) : CHUNK);
        const I *cv = colval + p0 + c;
        const double *nv = nzval + p0 + c;
        int64_t col[UNROLL];
        double val[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = tid + u * RPB;
            if (i < n) {
                col[u] = (int64_t)cv[i] - base;
                val[u] = nv[i];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = tid + u * RPB;
            if (i < n) s_prod[i] = val[u] * gather_x<SPLIT>(x_own, x_ghost, n_own, col[u]);
        }
        __syncthreads();
        if (tid < nr) {
            const int64_t a = lo > c ? lo : c;
            const int64_t e = hi < c + n ? hi : c + n;
            for (int64_t j = a; j < e; ++j) acc += s_prod[j - c];
        }
        __syncthreads();
    }
    if (tid < nr) y[r0 + tid] = acc;
    if (dot_partial) block_dot_epilogue(s_prod, dot_partial, blk, tid < nr ? acc * x_own[r0 + tid] : 0.0);
}

// IN != OUT: plan-time NARROWING of an Int64 matrix (the reference's default Ti = Int, src/backends.jl:348,369) whose
// nonzero count and split column space fit Int32 -- the kernels then stream 4-byte indices (hpcla_remap_i64_to_i32)
template <typename IN, typename OUT>
__global__ __launch_bounds__(256) void remap_kernel(const IN *__restrict__ in,
                                                    const OUT *__restrict__ map, OUT *__restrict__ out,
                                                    int64_t n, int base)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = map[(int64_t)in[i] - base];
}

__global__ __launch_bounds__(256) void narrow_kernel(const int64_t *__restrict__ in, int32_t *__restrict__ out,
                                                     int64_t n, unsigned int *__restrict__ overflow)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < n; i += stride) {
        const int64_t v = in[i];
        bad |= (v < INT32_MIN || v > INT32_MAX);
        out[i] = (int32_t)v;
    }
    if (overflow && bad) atomicOr(overflow, 1u);
}

template <typename I>
__global__ __launch_bounds__(RPB) void classify_blocks_kernel(const I *__restrict__ rowptr,
                                                              const I *__restrict__ colval,
                                                              int64_t nrows, int base,
                                                              int64_t n_own, int rpb,
                                                              int32_t *__restrict__ flags)
{
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int nr = (int)((nrows - r0) < rpb ? (nrows - r0) : rpb);
    const int64_t p0 = (int64_t)rowptr[r0] - base;
    const int64_t p1 = (int64_t)rowptr[r0 + nr] - base;
    int any = 0;
    for (int64_t j = p0 + threadIdx.x; j < p1; j += RPB) any |= ((int64_t)colval[j] - base >= n_own);
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flags[blockIdx.x] = any ? 1 : 0;
}

template <typename I>
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ x,
                                                     const I *__restrict__ src,
                                                     const I *__restrict__ dst,
                                                     double *__restrict__ out, int64_t n, int base)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int64_t d = dst ? (int64_t)dst[i] - base : i;
        out[d] = x[(int64_t)src[i] - base];
    }
}

static inline uint32_t stream_grid(int64_t n, int threads)
{
    int64_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;   // 256 CUs x 16: grid-stride the rest
    return (uint32_t)g;
}

// ---- per-matrix block order (a performance hint, never a correctness matter: every order is a bijection) ----------
static std::mutex g_order_mu;
static std::unordered_map<const void *, int> g_order;          // rowptr device pointer -> log2(group)
static std::atomic<int> g_order_count{0};

static int block_order_of(const void *rowptr)
{
    static const int forced = [] {                              // HPCLA_SPMV_XCD_GROUP=G: every launch (experiments)
        const char *e = getenv("HPCLA_SPMV_XCD_GROUP");
        int g = e ? atoi(e) : 0, l = 0;
        while (g > 1) { g >>= 1; ++l; }
        return l > 10 ? 10 : l;
    }();
    if (forced) return forced;
    if (g_order_count.load(std::memory_order_relaxed) == 0) return 0;
    std::lock_guard<std::mutex> lock(g_order_mu);
    auto it = g_order.find(rowptr);
    return it == g_order.end() ? 0 : it->second;
}

// Store policy of y (row-gather kernel): NON-TEMPORAL by default (HPCLA_SPMV_NT_Y=0: plain stores).  y is written once per
// launch and read by nothing inside it, so its lines only displace x and A lines from L2 / Infinity Cache; measured with
// alternating processes on one box (profiles/r04_spmv_nontemporal_y.log): headline 0.2309 / 0.2276 -> 0.2153 / 0.2156 ms
// (0.73-0.74 -> 0.78 of peak), and the CG iteration -- whose NEXT kernel reads y -- 0.4840 / 0.4881 -> 0.4799 / 0.4801: the
// 134 MB vector does not survive the iteration's other traffic in the Infinity Cache anyway.  (Non-temporal LOADS of the A
// stream are the opposite: +10 % in the harness, profiles/r04_spmv_rowgather_library.log.)
static int spmv_nt_y(bool /*with_dot*/)
{
    static const int on = [] {
        const char *e = getenv("HPCLA_SPMV_NT_Y");
        return e ? (atoi(e) ? 1 : 0) : 1;
    }();
    return on;
}

template <typename I>
static int spmv_launch(const I *rowptr, const I *colval, const double *nzval, const double *x_own,
                       const double *x_ghost, int64_t n_own, bool split, double *y, int64_t nrows,
                       int64_t nnz, int index_base, const int32_t *block_list, int64_t n_blocks,
                       void *stream, double *dot_partial = nullptr, int64_t block_base = -1)
{
    if (nrows < 0 || nnz < 0) return set_error(HPCLA_ERR_INVALID, "spmv: negative size");
    if (index_base != 0 && index_base != 1)
        return set_error(HPCLA_ERR_INVALID, "spmv: index_base must be 0 or 1");
    if (nrows == 0) return HPCLA_OK;
    if (!rowptr || !y) return set_error(HPCLA_ERR_INVALID, "spmv: null rowptr/y");
    if (nnz > 0 && (!colval || !nzval || !x_own))
        return set_error(HPCLA_ERR_INVALID, "spmv: null colval/nzval/x with nnz > 0");
    const int64_t all_blocks = (nrows + RPB - 1) / RPB;
    int64_t launch_blocks = all_blocks;
    if (block_list) {
        if (n_blocks < 0 || n_blocks > all_blocks)
            return set_error(HPCLA_ERR_INVALID, "spmv: n_blocks out of range");
        launch_blocks = n_blocks;
        block_base = 0;
    } else if (block_base >= 0) {               // contiguous run [block_base, block_base + n_blocks)
        if (n_blocks < 0 || block_base + n_blocks > all_blocks)
            return set_error(HPCLA_ERR_INVALID, "spmv: block range out of bounds");
        launch_blocks = n_blocks;
    } else {
        block_base = 0;
    }
    if (launch_blocks == 0) return HPCLA_OK;
    if (launch_blocks > 0x7fffffffLL) return set_error(HPCLA_ERR_INVALID, "spmv: too many blocks");
    dim3 grid((uint32_t)launch_blocks), block(RPB);
    hipStream_t s = as_stream(stream);
    // no ghost segment => no column can be >= n_own: take the plain-x kernel (no per-entry select)
    if (split && !x_ghost) split = false;
    // 16-byte staging loads from any quad-aligned entry offset: colval 4*sizeof(I)-, nzval 32-byte aligned (else: the fallback kernel)
    const bool aligned = (reinterpret_cast<uintptr_t>(colval) % (4 * sizeof(I)) == 0) &&
                         (reinterpret_cast<uintptr_t>(nzval) % 32 == 0);
    const BlockSel bs{block_list, block_base, nullptr, 0, 0, launch_blocks, block_list ? 0 : block_order_of(rowptr),
                      spmv_nt_y(dot_partial != nullptr)};
    HaloWait nowait;
    memset(&nowait, 0, sizeof(nowait));
    PushArgs nopush;
    memset(&nopush, 0, sizeof(nopush));
    if (aligned) {
        if (split)
            spmv_rowgather_kernel<I, true, false><<<grid, block, 0, s>>>(
                rowptr, colval, nzval, x_own, x_ghost, n_own, y, nrows, nnz, index_base, bs, dot_partial, nowait, nopush);
        else
            spmv_rowgather_kernel<I, false, false><<<grid, block, 0, s>>>(
                rowptr, colval, nzval, x_own, nullptr, 0, y, nrows, nnz, index_base, bs, dot_partial, nowait, nopush);
    } else {
        if (split)
            spmv_rowblock_kernel<I, true, false><<<grid, block, 0, s>>>(
                rowptr, colval, nzval, x_own, x_ghost, n_own, y, nrows, index_base, bs, dot_partial, nowait, nopush);
        else
            spmv_rowblock_kernel<I, false, false><<<grid, block, 0, s>>>(
                rowptr, colval, nzval, x_own, nullptr, 0, y, nrows, index_base, bs, dot_partial, nowait, nopush);
    }
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

// Fused distributed launch (peer-window transport, comm.hip): ONE grid = interior blocks (a contiguous
// run from interior_base, or interior_list) followed by the boundary blocks, whose workgroups wait for
// the neighbours' flags before their first ghost gather.  The exchange overlaps every interior block
// without a second stream or a second launch.
template <typename I>
static int spmv_launch_fused(const I *rowptr, const I *colval, const double *nzval, const double *x_own,
                             const double *x_ghost, int64_t n_own, double *y, int64_t nrows, int64_t nnz,
                             int index_base, const int32_t *interior_list, int64_t interior_base,
                             int64_t n_interior, const int32_t *boundary_list, int64_t n_boundary,
                             const HaloWait &hw, const PushArgs &pa, void *stream, double *dot_partial)
{
    if (nrows < 0 || nnz < 0) return set_error(HPCLA_ERR_INVALID, "spmv: negative size");
    if (index_base != 0 && index_base != 1)
        return set_error(HPCLA_ERR_INVALID, "spmv: index_base must be 0 or 1");
    if (nrows > 0 && (!rowptr || !y)) return set_error(HPCLA_ERR_INVALID, "spmv: null rowptr/y");
    if (nnz > 0 && (!colval || !nzval || !x_own))
        return set_error(HPCLA_ERR_INVALID, "spmv: null colval/nzval/x with nnz > 0");
    const int64_t all_blocks = (nrows + RPB - 1) / RPB;
    if (n_interior < 0 || n_boundary < 0 || n_interior + n_boundary > all_blocks)
        return set_error(HPCLA_ERR_INVALID, "spmv: block counts out of range");
    if (n_boundary > 0 && !boundary_list) return set_error(HPCLA_ERR_INVALID, "spmv: null boundary list");
    if (n_boundary > 0 && !x_ghost) return set_error(HPCLA_ERR_INVALID, "spmv: boundary blocks without a ghost segment");
    if (!interior_list && n_interior > 0 && (interior_base < 0 || interior_base + n_interior > all_blocks))
        return set_error(HPCLA_ERR_INVALID, "spmv: interior run out of bounds");
    const int64_t launch_blocks = pa.n_blocks + n_interior + n_boundary;   // push workgroups lead the grid
    if (launch_blocks == 0) return HPCLA_OK;
    if (launch_blocks > 0x7fffffffLL) return set_error(HPCLA_ERR_INVALID, "spmv: too many blocks");
    dim3 grid((uint32_t)launch_blocks), block(RPB);
    hipStream_t s = as_stream(stream);
    const bool aligned = (reinterpret_cast<uintptr_t>(colval) % (4 * sizeof(I)) == 0) &&
                         (reinterpret_cast<uintptr_t>(nzval) % 32 == 0);
    const BlockSel bs{boundary_list, 0, interior_list, interior_base, n_interior, n_interior,
                      interior_list ? 0 : block_order_of(rowptr), spmv_nt_y(dot_partial != nullptr)};
    if (aligned)
        spmv_rowgather_kernel<I, true, true><<<grid, block, 0, s>>>(
            rowptr, colval, nzval, x_own, x_ghost, n_own, y, nrows, nnz, index_base, bs, dot_partial, hw, pa);
    else
        spmv_rowblock_kernel<I, true, true><<<grid, block, 0, s>>>(
            rowptr, colval, nzval, x_own, x_ghost, n_own, y, nrows, index_base, bs, dot_partial, hw, pa);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

int spmv_fused_i32(const int32_t *rowptr, const int32_t *colval, const double *nzval, const double *x_own,
                   const double *x_ghost, int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base,
                   const int32_t *interior_list, int64_t interior_base, int64_t n_interior,
                   const int32_t *boundary_list, int64_t n_boundary, const HaloWait &hw, const PushArgs &pa,
                   void *stream, double *dot_partial)
{
    return spmv_launch_fused<int32_t>(rowptr, colval, nzval, x_own, x_ghost, n_own, y, nrows, nnz, index_base,
                                      interior_list, interior_base, n_interior, boundary_list, n_boundary, hw,
                                      pa, stream, dot_partial);
}
int spmv_fused_i64(const int64_t *rowptr, const int64_t *colval, const double *nzval, const double *x_own,
                   const double *x_ghost, int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base,
                   const int32_t *interior_list, int64_t interior_base, int64_t n_interior,
                   const int32_t *boundary_list, int64_t n_boundary, const HaloWait &hw, const PushArgs &pa,
                   void *stream, double *dot_partial)
{
    return spmv_launch_fused<int64_t>(rowptr, colval, nzval, x_own, x_ghost, n_own, y, nrows, nnz, index_base,
                                      interior_list, interior_base, n_interior, boundary_list, n_boundary, hw,
                                      pa, stream, dot_partial);
}

// exported to comm.hip (spmv_dist)
int spmv_split_i32(const int32_t *rowptr, const int32_t *colval, const double *nzval,
                   const double *x_own, const double *x_ghost, int64_t n_own, double *y,
                   int64_t nrows, int64_t nnz, int index_base, const int32_t *bl, int64_t nb,
                   void *stream, double *dot_partial, int64_t block_base)
{
    return spmv_launch<int32_t>(rowptr, colval, nzval, x_own, x_ghost, n_own, true, y, nrows, nnz,
                                index_base, bl, nb, stream, dot_partial, block_base);
}
int spmv_split_i64(const int64_t *rowptr, const int64_t *colval, const double *nzval,
                   const double *x_own, const double *x_ghost, int64_t n_own, double *y,
                   int64_t nrows, int64_t nnz, int index_base, const int32_t *bl, int64_t nb,
                   void *stream, double *dot_partial, int64_t block_base)
{
    return spmv_launch<int64_t>(rowptr, colval, nzval, x_own, x_ghost, n_own, true, y, nrows, nnz,
                                index_base, bl, nb, stream, dot_partial, block_base);
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int hpcla_spmv_rows_per_block(void) { return RPB; }

static void set_block_order(const void *rowptr, int group_log2)
{
    std::lock_guard<std::mutex> lock(g_order_mu);
    if (group_log2 <= 0) g_order.erase(rowptr);
    else g_order[rowptr] = group_log2;
    g_order_count.store((int)g_order.size(), std::memory_order_relaxed);
}

// Plan-time choice of the block order BY MEASUREMENT: the same launch the plan will make, timed under the natural order
// and groups of 8 / 32 / 64 row blocks, interleaved over a few rounds; the fastest stays registered for `rowptr`.  The
// natural order is kept unless a grouped one is >= 1 % faster.  (A model of which x lines two XCDs share predicts
// config 4's planes and the 256^3 cube, but not that 4096- and 8192-wide 2-D grids gain 3 % from groups of 32 / 64
// although their neighbours already share an XCD in the natural order, nor that a 1000-wide grid loses 1 % with
// groups of 64 -- profiles/r03_spmv_xcd_group_order.log -- so the choice is measured, like an FFT plan's.)
template <typename I>
static int tune_block_order(const I *rowptr, const I *colval_split, const double *nzval, const double *x_own,
                            const double *x_ghost, int64_t n_own, double *y_scratch, int64_t nrows, int64_t nnz,
                            int index_base, void *stream, int *chosen_group)
{
    if (chosen_group) *chosen_group = 1;
    if (nrows < 0 || nnz < 0 || !rowptr) return set_error(HPCLA_ERR_INVALID, "spmv_tune_block_order: bad arguments");
    set_block_order(rowptr, 0);
    const int64_t n_blocks = (nrows + RPB - 1) / RPB;
    if (n_blocks < 4096 || nnz == 0) return HPCLA_OK;      // small matrices live in the caches whatever the order
    if (!y_scratch) return set_error(HPCLA_ERR_INVALID, "spmv_tune_block_order: null scratch vector");
    constexpr int NC = 4, ROUNDS = 4, REPS = 4;            // round 0 warms up (clocks, TLBs) and is not counted
    const int cand[NC] = {0, 3, 5, 6};
    float ms[NC][ROUNDS];
    hipEvent_t e0, e1;
    HPCLA_CHECK_HIP(hipEventCreate(&e0));
    HPCLA_CHECK_HIP(hipEventCreate(&e1));
    hipStream_t s = as_stream(stream);
    int rc = HPCLA_OK;
    for (int r = 0; r < ROUNDS && rc == HPCLA_OK; ++r)
        for (int c = 0; c < NC && rc == HPCLA_OK; ++c) {
            set_block_order(rowptr, cand[c]);
            if (hipEventRecord(e0, s) != hipSuccess) { rc = set_error(HPCLA_ERR_HIP, "spmv_tune_block_order: event"); break; }
            for (int i = 0; i < REPS && rc == HPCLA_OK; ++i)
                rc = spmv_launch<I>(rowptr, colval_split, nzval, x_own, x_ghost, n_own, true, y_scratch, nrows, nnz,
                                    index_base, nullptr, 0, stream);
            if (rc != HPCLA_OK) break;
            if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                hipEventElapsedTime(&ms[c][r], e0, e1) != hipSuccess)
                rc = set_error(HPCLA_ERR_HIP, "spmv_tune_block_order: timing");
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    set_block_order(rowptr, 0);
    if (rc != HPCLA_OK) return rc;
    float best_t[NC];
    for (int c = 0; c < NC; ++c) {                         // median of the three counted rounds
        float a = ms[c][1], b = ms[c][2], d = ms[c][3];
        best_t[c] = a > b ? (b > d ? b : (a > d ? d : a)) : (a > d ? a : (b > d ? d : b));
    }
    int best = 0;
    for (int c = 1; c < NC; ++c)
        if (best_t[c] < best_t[best]) best = c;
    if (best != 0 && best_t[best] > 0.99f * best_t[0]) best = 0;
    set_block_order(rowptr, cand[best]);
    if (chosen_group) *chosen_group = 1 << cand[best];
    return HPCLA_OK;
}

HPCLA_API int hpcla_spmv_tune_block_order_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                                  const double *x_own, const double *x_ghost, int64_t n_own,
                                                  double *y_scratch, int64_t nrows, int64_t nnz, int index_base,
                                                  void *stream, int *chosen_group)
{
    return tune_block_order<int32_t>(rowptr, colval_split, nzval, x_own, x_ghost, n_own, y_scratch, nrows, nnz, index_base,
                                     stream, chosen_group);
}

HPCLA_API int hpcla_spmv_tune_block_order_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                                  const double *x_own, const double *x_ghost, int64_t n_own,
                                                  double *y_scratch, int64_t nrows, int64_t nnz, int index_base,
                                                  void *stream, int *chosen_group)
{
    return tune_block_order<int64_t>(rowptr, colval_split, nzval, x_own, x_ghost, n_own, y_scratch, nrows, nnz, index_base,
                                     stream, chosen_group);
}

HPCLA_API int hpcla_spmv_block_order_hint(const void *rowptr, int group)
{
    if (!rowptr) return set_error(HPCLA_ERR_INVALID, "spmv_block_order_hint: null rowptr");
    if (group < 0 || group > 1024 || (group & (group - 1)) != 0)
        return set_error(HPCLA_ERR_INVALID, "spmv_block_order_hint: group must be 0 or a power of two <= 1024");
    int l = 0;
    for (int g = group; g > 1; g >>= 1) ++l;
    set_block_order(rowptr, l);
    return HPCLA_OK;
}

HPCLA_API int hpcla_spmv_csr_f64_i32(const int32_t *rowptr, const int32_t *colval,
                                     const double *nzval, const double *x, double *y,
                                     int64_t nrows, int64_t nnz, int index_base, void *stream)
{
    return spmv_launch<int32_t>(rowptr, colval, nzval, x, nullptr, 0, false, y, nrows, nnz,
                                index_base, nullptr, 0, stream);
}

HPCLA_API int hpcla_spmv_csr_f64_i64(const int64_t *rowptr, const int64_t *colval,
                                     const double *nzval, const double *x, double *y,
                                     int64_t nrows, int64_t nnz, int index_base, void *stream)
{
    return spmv_launch<int64_t>(rowptr, colval, nzval, x, nullptr, 0, false, y, nrows, nnz,
                                index_base, nullptr, 0, stream);
}

HPCLA_API int hpcla_spmv_split_f64_i32(const int32_t *rowptr, const int32_t *colval_split,
                                       const double *nzval, const double *x_own,
                                       const double *x_ghost, int64_t n_own, double *y,
                                       int64_t nrows, int64_t nnz, int index_base,
                                       const int32_t *block_list, int64_t n_blocks, void *stream)
{
    return spmv_split_i32(rowptr, colval_split, nzval, x_own, x_ghost, n_own, y, nrows, nnz,
                          index_base, block_list, n_blocks, stream, nullptr, -1);
}

HPCLA_API int hpcla_spmv_split_f64_i64(const int64_t *rowptr, const int64_t *colval_split,
                                       const double *nzval, const double *x_own,
                                       const double *x_ghost, int64_t n_own, double *y,
                                       int64_t nrows, int64_t nnz, int index_base,
                                       const int32_t *block_list, int64_t n_blocks, void *stream)
{
    return spmv_split_i64(rowptr, colval_split, nzval, x_own, x_ghost, n_own, y, nrows, nnz,
                          index_base, block_list, n_blocks, stream, nullptr, -1);
}

// ---- OPT-IN long rows: y = A*x with the listed rows summed in tree order (see spmv_longrow_partial_kernel) ----------------
template <typename I>
static int spmv_longrows(const I *rowptr, const I *colval, const double *nzval, const double *x_own, const double *x_ghost,
                         int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base, const int64_t *long_rows,
                         int64_t n_long, int64_t long_min, double *work, void *stream)
{
    if (nrows < 0 || nnz < 0 || n_long < 0 || n_long > 65535)
        return set_error(HPCLA_ERR_INVALID, "spmv_longrows: bad sizes (at most 65535 long rows)");
    if (index_base != 0 && index_base != 1) return set_error(HPCLA_ERR_INVALID, "spmv_longrows: index_base must be 0 or 1");
    if (long_min < 2 * RG_CHW) return set_error(HPCLA_ERR_INVALID, "spmv_longrows: long_min must be at least 928 entries");
    if (nrows == 0) return HPCLA_OK;
    if (!rowptr || !y || (nnz > 0 && (!colval || !nzval || !x_own)) || (n_long > 0 && (!long_rows || !work)))
        return set_error(HPCLA_ERR_INVALID, "spmv_longrows: null pointer");
    const bool aligned = (reinterpret_cast<uintptr_t>(colval) % (4 * sizeof(I)) == 0) &&
                         (reinterpret_cast<uintptr_t>(nzval) % 32 == 0);
    // unaligned arrays: the default entry's fallback kernel sums EVERY row sequentially in stored order -- inside this entry's
    // tolerance by definition (it is the sum the tolerance is measured against) -- instead of refusing the product
    if (!aligned)
        return spmv_launch<I>(rowptr, colval, nzval, x_own, x_ghost, n_own, x_ghost != nullptr, y, nrows, nnz, index_base, nullptr,
                              0, stream);
    const int64_t all_blocks = (nrows + RPB - 1) / RPB;
    if (all_blocks > 0x7fffffffLL) return set_error(HPCLA_ERR_INVALID, "spmv_longrows: too many blocks");
    hipStream_t s = as_stream(stream);
    const bool split = x_ghost != nullptr;
    const BlockSel bs{nullptr, 0, nullptr, 0, 0, all_blocks, block_order_of(rowptr), spmv_nt_y(false)};
    HaloWait nowait;
    memset(&nowait, 0, sizeof(nowait));
    PushArgs nopush;
    memset(&nopush, 0, sizeof(nopush));
    if (split)
        spmv_rowgather_kernel<I, true, false, true><<<dim3((uint32_t)all_blocks), dim3(RPB), 0, s>>>(
            rowptr, colval, nzval, x_own, x_ghost, n_own, y, nrows, nnz, index_base, bs, nullptr, nowait, nopush, long_min);
    else
        spmv_rowgather_kernel<I, false, false, true><<<dim3((uint32_t)all_blocks), dim3(RPB), 0, s>>>(
            rowptr, colval, nzval, x_own, nullptr, 0, y, nrows, nnz, index_base, bs, nullptr, nowait, nopush, long_min);
    HPCLA_CHECK_LAUNCH();
    if (n_long > 0) {
        if (split)
            spmv_longrow_partial_kernel<I, true><<<dim3(LR_CHUNKS, (uint32_t)n_long), dim3(256), 0, s>>>(
                rowptr, colval, nzval, x_own, x_ghost, n_own, index_base, long_rows, work);
        else
            spmv_longrow_partial_kernel<I, false><<<dim3(LR_CHUNKS, (uint32_t)n_long), dim3(256), 0, s>>>(
                rowptr, colval, nzval, x_own, nullptr, 0, index_base, long_rows, work);
        HPCLA_CHECK_LAUNCH();
        spmv_longrow_final_kernel<<<dim3((uint32_t)n_long), dim3(256), 0, s>>>(long_rows, work, y);
        HPCLA_CHECK_LAUNCH();
    }
    return HPCLA_OK;
}

HPCLA_API int64_t hpcla_spmv_longrows_work_bytes(int64_t n_long)
{
    return (n_long > 0 ? n_long : 1) * (int64_t)LR_CHUNKS * (int64_t)sizeof(double);
}

HPCLA_API int hpcla_spmv_longrows_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                          const double *x_own, const double *x_ghost, int64_t n_own, double *y,
                                          int64_t nrows, int64_t nnz, int index_base, const int64_t *long_rows,
                                          int64_t n_long, int64_t long_min, double *work, void *stream)
{
    return spmv_longrows<int32_t>(rowptr, colval_split, nzval, x_own, x_ghost, n_own, y, nrows, nnz, index_base, long_rows,
                                  n_long, long_min, work, stream);
}

HPCLA_API int hpcla_spmv_longrows_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                          const double *x_own, const double *x_ghost, int64_t n_own, double *y,
                                          int64_t nrows, int64_t nnz, int index_base, const int64_t *long_rows,
                                          int64_t n_long, int64_t long_min, double *work, void *stream)
{
    return spmv_longrows<int64_t>(rowptr, colval_split, nzval, x_own, x_ghost, n_own, y, nrows, nnz, index_base, long_rows,
                                  n_long, long_min, work, stream);
}

template <typename IN, typename OUT>
static int remap_impl(const IN *in, const OUT *map, OUT *out, int64_t n, int index_base, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "remap: negative size");
    if (index_base != 0 && index_base != 1) return set_error(HPCLA_ERR_INVALID, "remap: index_base must be 0 or 1");
    if (n == 0) return HPCLA_OK;
    if (!in || !map || !out) return set_error(HPCLA_ERR_INVALID, "remap: null pointer");
    remap_kernel<IN, OUT><<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(in, map, out, n, index_base);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_remap_i32(const int32_t *in, const int32_t *map, int32_t *out, int64_t n,
                              int index_base, void *stream)
{
    return remap_impl<int32_t, int32_t>(in, map, out, n, index_base, stream);
}
HPCLA_API int hpcla_remap_i64(const int64_t *in, const int64_t *map, int64_t *out, int64_t n,
                              int index_base, void *stream)
{
    return remap_impl<int64_t, int64_t>(in, map, out, n, index_base, stream);
}

// ---- plan-time narrowing of Int64 structures (the kernels' index bytes: 16 -> 12 B per stored entry) ----------------
HPCLA_API int hpcla_remap_i64_to_i32(const int64_t *in, const int32_t *map, int32_t *out, int64_t n,
                                     int index_base, void *stream)
{
    return remap_impl<int64_t, int32_t>(in, map, out, n, index_base, stream);
}

HPCLA_API int hpcla_narrow_i64_to_i32(const int64_t *in, int32_t *out, int64_t n, uint32_t *overflow_dev, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "narrow: negative size");
    if (n == 0) return HPCLA_OK;
    if (!in || !out) return set_error(HPCLA_ERR_INVALID, "narrow: null pointer");
    narrow_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(in, out, n, overflow_dev);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

template <typename I>
static int classify_impl(const I *rowptr, const I *colval, int64_t nrows, int index_base,
                         int64_t n_own, int rpb, int32_t *flags, void *stream)
{
    if (nrows < 0 || rpb < 1) return set_error(HPCLA_ERR_INVALID, "classify: bad size");
    if (nrows == 0) return HPCLA_OK;
    if (!rowptr || !flags) return set_error(HPCLA_ERR_INVALID, "classify: null pointer");
    const int64_t nb = (nrows + rpb - 1) / rpb;
    classify_blocks_kernel<I><<<(uint32_t)nb, RPB, 0, as_stream(stream)>>>(
        rowptr, colval, nrows, index_base, n_own, rpb, flags);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_classify_blocks_i32(const int32_t *rowptr, const int32_t *colval_split,
                                        int64_t nrows, int index_base, int64_t n_own,
                                        int rows_per_block, int32_t *flags, void *stream)
{
    return classify_impl<int32_t>(rowptr, colval_split, nrows, index_base, n_own, rows_per_block,
                                  flags, stream);
}
HPCLA_API int hpcla_classify_blocks_i64(const int64_t *rowptr, const int64_t *colval_split,
                                        int64_t nrows, int index_base, int64_t n_own,
                                        int rows_per_block, int32_t *flags, void *stream)
{
    return classify_impl<int64_t>(rowptr, colval_split, nrows, index_base, n_own, rows_per_block,
                                  flags, stream);
}

template <typename I>
static int gather_impl(const double *x, const I *src, const I *dst, double *out, int64_t n,
                       int index_base, void *stream)
{
    if (n < 0) return set_error(HPCLA_ERR_INVALID, "gather: negative size");
    if (n == 0) return HPCLA_OK;
    if (!x || !src || !out) return set_error(HPCLA_ERR_INVALID, "gather: null pointer");
    gather_kernel<I><<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(x, src, dst, out, n,
                                                                          index_base);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_gather_f64_i32(const double *x, const int32_t *src, const int32_t *dst,
                                   double *out, int64_t n, int index_base, void *stream)
{
    return gather_impl<int32_t>(x, src, dst, out, n, index_base, stream);
}
HPCLA_API int hpcla_gather_f64_i64(const double *x, const int64_t *src, const int64_t *dst,
                                   double *out, int64_t n, int index_base, void *stream)
{
    return gather_impl<int64_t>(x, src, dst, out, n, index_base, stream);
}
