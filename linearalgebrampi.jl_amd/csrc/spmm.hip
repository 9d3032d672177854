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
#include <limits.h>
#include <stdlib.h>

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
// a B row with two 16-byte loads: per stored entry a lane issues 2 loads and 8 flops.  A workgroup owns 64
// rows (one lane-group each).
//
// Instruction diet (round 2).  The round-1 form of this kernel staged 32-bit column ids and let every one
// of a row's four lanes redo, per entry, the own/ghost select, the 64-bit row-stride multiply and the
// pointer add, next to per-element tail predicates: 1.06e8 VALU wave-instructions on the 5-point matrix x 16
// columns where the arithmetic needs 2.1e7 (profiles/r01_pmc_spmm_vec_stencil.txt), 0.52 of the HBM
// roofline.  Now the staging pass -- which touches every entry ONCE -- resolves it to the ADDRESS of its
// B row and parks {address, value} as one 16-byte LDS record; a lane's inner step is then: one
// ds_read_b128, one 64-bit add of its (loop-invariant) column byte offset, two 16-byte loads, eight
// flops.  Tail handling is outside the loop (pairs, then at most one single), so there are no per-element
// predicates.  Each output column is still accumulated entry by entry in stored order: same bits.
// Measured after the diet (profiles/r02_pmc_spmm_vec_stencil.txt): VALU instructions -26 %, LDS instructions
// -54 %, time -5 % (0.643 -> 0.61 ms on the 5-point matrix x 16): the kernel is not issue-bound.  Its L2 misses
// are exactly the compulsory lines (A + B + C once), so there is no redundant fabric traffic either.  Tried on
// top of it and dropped, with the numbers (profiles/r02_spmm_*.log): an XCD-grouped block order (runs of G
// consecutive row blocks per XCD, G = 16..2048: 0.6185..0.659 ms vs 0.6138 natural -- nothing to win when
// the misses are compulsory); software-pipelined tiles (a workgroup owning T = 2/4/8 tiles, rowptr slices
// up front, next pass's A entries requested before the current pass's B gathers, double-buffered LDS
// records: 0.583 / 0.587 / 0.592 vs 0.595 ms on the stencil, and 1.44 / 1.50 / 1.80 vs 1.38 ms on config 5's
// random pattern -- the dependent chain is not the limit); eight lanes per row, one 16-byte load per entry,
// a whole 5-entry row's gathers issued in ONE round trip (62 VGPRs, 8 waves/SIMD: 0.626 vs 0.589 ms, slower --
// the number of gather rounds is not the limit either); kept: the HALF64 lane mapping below (+1.5 %).  For
// scale: on the same box a device copy B -> C runs at 4.8 TB/s, a read-only pass at 5.9, a fill at 6.3
// (profiles/r02_stream_mix_ceiling.log); this kernel's byte mix (40 % writes) moved as separate ideal streams
// takes 0.544 ms there, the kernel 0.612.
// Also tried in round 1 and dropped: no LDS staging (entries handed round a lane-group with shuffles) and an
// XCD-sliced block order.
constexpr int VG = 4;                        // lanes per row
constexpr int VCPL = KT / VG;                // 4 columns per lane
// records staged per pass (template CH): 1536 * 16 B = 24 KiB (6 workgroups per CU), or 512 * 16 B = 8 KiB
// for matrices with <= 8 entries per row on average, whose 64-row blocks fit one small pass (8 per CU)

typedef double vdouble2 __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(16))) SpmmEntry {
    const double *row;                       // start of the entry's B row (own block or ghost segment)
    double val;
};

typedef const vdouble2 __attribute__((address_space(1))) *gvec2_ptr;   // global address space: global_load, not flat_load

template <typename I, bool SPLIT, int VU, int CHUNK_V, bool HALF64>
__global__ __launch_bounds__(TPB_MM) void spmm_rowblock_vec_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ B_own, int64_t b_rs, const double *__restrict__ B_ghost, int64_t bg_rs,
    int64_t n_own, double *__restrict__ C, int64_t c_rs, int64_t nrows, int k, int base,
    const int32_t *__restrict__ block_list, uint32_t nblocks)
{
    __shared__ SpmmEntry s_ent[CHUNK_V];

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
        // lane -> columns.  Default: four adjacent columns (32 contiguous bytes of the B row, two 16-byte loads).
        // HALF64 (full 16-column tiles only): columns {2l, 2l+1} and {8+2l, 8+2l+1}, i.e. the four lanes of a row
        // read 64 CONTIGUOUS bytes per load instruction (one half line each) instead of every other 16 bytes
        // of the whole line.
        const int c = HALF64 ? kt + 2 * l : kt + VCPL * l;
        const bool col_ok = c < k;
        constexpr int SECOND = HALF64 ? 64 : 16;                           // byte offset of the lane's second load
        const int64_t lane_bytes = (int64_t)c * (int64_t)sizeof(double);   // this lane's slice of every B row
        double acc[VCPL];
#pragma unroll
        for (int q = 0; q < VCPL; ++q) acc[q] = 0.0;

        for (int64_t ch = 0; ch < total; ch += CHUNK_V) {
            const int n = (int)((total - ch) < CHUNK_V ? (total - ch) : CHUNK_V);
            __syncthreads();   // previous pass finished reading LDS
            for (int i = tid; i < n; i += TPB_MM) {
                const int64_t col = (int64_t)__builtin_nontemporal_load(colval + p0 + ch + i) - base;
                SpmmEntry e;
                e.val = __builtin_nontemporal_load(nzval + p0 + ch + i);
                e.row = (SPLIT && col >= n_own) ? B_ghost + (col - n_own) * bg_rs : B_own + col * b_rs;
                s_ent[i] = e;
            }
            __syncthreads();
            if (!col_ok) continue;
            int j = (int)((lo > ch ? lo : ch) - ch);
            const int e = (int)((hi < ch + n ? hi : ch + n) - ch);
            // VU entries per step: 2*VU independent 16-byte loads in flight per lane
            for (; j + VU <= e; j += VU) {
                SpmmEntry en[VU];
                vdouble2 b0[VU], b1[VU];
#pragma unroll
                for (int u = 0; u < VU; ++u) en[u] = s_ent[j + u];
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    const char *src = reinterpret_cast<const char *>(en[u].row) + lane_bytes;
                    b0[u] = *(gvec2_ptr)(src);
                    b1[u] = *(gvec2_ptr)(src + SECOND);
                }
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    acc[0] += en[u].val * b0[u].x;
                    acc[1] += en[u].val * b0[u].y;
                    acc[2] += en[u].val * b1[u].x;
                    acc[3] += en[u].val * b1[u].y;
                }
            }
            for (; j < e; ++j) {              // at most VU-1 leftovers
                const SpmmEntry en = s_ent[j];
                const char *src = reinterpret_cast<const char *>(en.row) + lane_bytes;
                const vdouble2 b0 = *(gvec2_ptr)(src);
                const vdouble2 b1 = *(gvec2_ptr)(src + SECOND);
                acc[0] += en.val * b0.x;
                acc[1] += en.val * b0.y;
                acc[2] += en.val * b1.x;
                acc[3] += en.val * b1.y;
            }
        }
        if (g < nr && col_ok) {
            double *dst = C + (r0 + g) * c_rs + c;
            vdouble2 o0, o1;
            o0.x = acc[0]; o0.y = acc[1]; o1.x = acc[2]; o1.y = acc[3];
            *reinterpret_cast<vdouble2 *>(dst) = o0;
            *reinterpret_cast<vdouble2 *>(dst + SECOND / 8) = o1;
        }
    }
}

// LDS-tiled form for banded / stencil-like matrices, k = 16 (north star: "LDS-staged tiles").  A pure
// streaming kernel moves this workload's byte mix at 6.1 TB/s with the same 64-row granularity
// (benchmarks/tune/stream_mix.hip: 0.438 ms) while the gather form above takes 0.60 ms: what separates them is
// the 5x re-read of B rows through L1/L2 as dependent 16-byte gathers (7.1e7 L2 requests where a stream needs
// 2.1e7).  Here a workgroup finds the DISTINCT B rows its 64 matrix rows touch ON THE FLY -- no plan-time
// preprocessing, no extra arrays: a presence bitmap over the block's column window in LDS, a popcount scan,
// and every entry gets the slot of its column -- then streams exactly those rows into LDS ONCE, whole
// 128-byte rows, all loads in flight together, and computes from LDS.  A 5-point block touches 194 distinct
// rows (64 own-line, 2 edge, 2 x 64 neighbour-line) instead of issuing 320 gathers in three dependent rounds.
// Blocks that do not qualify (more than T_CH entries, a column window wider than T_W, more than T_MAX
// distinct rows: random patterns, 3-D planes) take the gather loop inside the same kernel.  Per (row,
// column) the products are still added entry by entry in stored order: same bits.
constexpr int T_CH = 512;                    // entries of a block handled by the tile path
constexpr int T_W = 16384;                   // column window (bits of the presence bitmap)
constexpr int T_WORDS = T_W / 32;
constexpr int T_MAX = 200;                   // distinct B rows held in LDS
constexpr int T_STRIDE = KT + 2;             // doubles per tile row: 144 bytes, so consecutive slots shift by 4 banks

template <typename I, bool SPLIT>
__global__ __launch_bounds__(TPB_MM) __attribute__((amdgpu_waves_per_eu(4, 8))) void spmm_rowblock_tile_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ B_own, int64_t b_rs, const double *__restrict__ B_ghost, int64_t bg_rs,
    int64_t n_own, double *__restrict__ C, int64_t c_rs, int64_t nrows, int base,
    const int32_t *__restrict__ block_list, uint32_t nblocks)
{
    // LDS: tile | values | slots | bitmap | prefix | list | scratch.  The gather fallback re-uses the tile area for
    // its {address, value} records.
    __shared__ __attribute__((aligned(16))) double s_tile[T_MAX * T_STRIDE];       // 28 800 B
    __shared__ double s_val[T_CH];                                                   //  4 096 B
    __shared__ uint16_t s_slot[T_CH];                                                //  1 024 B
    __shared__ uint32_t s_bits[T_WORDS];                                             //  2 048 B
    __shared__ uint16_t s_pref[T_WORDS];                                             //  1 024 B
    __shared__ int32_t s_list[T_MAX];                                                //    800 B
    __shared__ int64_t s_red[8];                                                     // min / max per wave
    __shared__ int32_t s_wsum[4];
    static_assert(sizeof(SpmmEntry) * T_CH <= sizeof(double) * T_MAX * T_STRIDE, "fallback records alias the tile");

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
    const int c = 2 * l;                                   // HALF64 mapping: columns {2l, 2l+1} and {8+2l, 9+2l}
    double acc[VCPL] = {0.0, 0.0, 0.0, 0.0};

    bool tiled = total <= T_CH;                            // workgroup-uniform from here on
    int64_t col[2] = {0, 0};
    double val[2] = {0.0, 0.0};
    int64_t cmin = 0;
    int D = 0;
    if (tiled) {
        // ---- entries (<= 2 per thread) and the block's column window ---------------------------------------
        int64_t mn = INT64_MAX, mx = INT64_MIN;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * TPB_MM;
            if (i < total) {
                col[u] = (int64_t)__builtin_nontemporal_load(colval + p0 + i) - base;
                val[u] = __builtin_nontemporal_load(nzval + p0 + i);
                mn = col[u] < mn ? col[u] : mn;
                mx = col[u] > mx ? col[u] : mx;
            }
        }
        for (int i = tid; i < T_WORDS; i += TPB_MM) s_bits[i] = 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int64_t a = __shfl_xor(mn, off, 64), bb = __shfl_xor(mx, off, 64);
            mn = a < mn ? a : mn;
            mx = bb > mx ? bb : mx;
        }
        if ((tid & 63) == 0) { s_red[tid >> 6] = mn; s_red[4 + (tid >> 6)] = mx; }
        __syncthreads();
        int64_t cmax = s_red[4];
        cmin = s_red[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            cmin = s_red[w] < cmin ? s_red[w] : cmin;
            cmax = s_red[4 + w] > cmax ? s_red[4 + w] : cmax;
        }
        tiled = total > 0 && cmax - cmin < T_W;
        if (tiled) {
            // ---- presence bitmap -> popcount scan -> slot of every column -------------------------------------
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + u * TPB_MM;
                if (i < total) {
                    const uint32_t rel = (uint32_t)(col[u] - cmin);
                    atomicOr(&s_bits[rel >> 5], 1u << (rel & 31));
                }
            }
            __syncthreads();
            const uint32_t w0 = s_bits[2 * tid], w1 = s_bits[2 * tid + 1];
            const int c0 = __popc(w0), c1 = __popc(w1);
            int incl = c0 + c1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off, 64);
                if ((tid & 63) >= off) incl += up;
            }
            if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
            __syncthreads();
            int wbase = 0;
            for (int w = 0; w < (tid >> 6); ++w) wbase += s_wsum[w];
            D = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
            const int excl = wbase + incl - (c0 + c1);
            s_pref[2 * tid] = (uint16_t)excl;
            s_pref[2 * tid + 1] = (uint16_t)(excl + c0);
            tiled = D <= T_MAX;
            if (tiled) {
                int pos = excl;
                uint32_t w = w0;
                while (w) { s_list[pos++] = (2 * tid) * 32 + __builtin_ctz(w); w &= w - 1; }
                w = w1;
                while (w) { s_list[pos++] = (2 * tid + 1) * 32 + __builtin_ctz(w); w &= w - 1; }
            }
            __syncthreads();
        }
    }

    if (tiled) {
        // ---- park {slot, value}; stream the D distinct B rows into LDS, all loads in flight together ---------
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * TPB_MM;
            if (i < total) {
                const uint32_t rel = (uint32_t)(col[u] - cmin);
                const uint32_t word = rel >> 5, bit = rel & 31;
                s_slot[i] = (uint16_t)(s_pref[word] + __popc(s_bits[word] & ((1u << bit) - 1u)));
                s_val[i] = val[u];
            }
        }
        constexpr int FILL = (T_MAX * 8 + TPB_MM - 1) / TPB_MM;       // 16-byte pieces per thread (7)
        vdouble2 piece[FILL];
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int i = tid + f * TPB_MM;
            if (i < D * 8) {
                const int64_t cc = cmin + s_list[i >> 3];
                const double *src = (SPLIT && cc >= n_own) ? B_ghost + (cc - n_own) * bg_rs : B_own + cc * b_rs;
                piece[f] = *(gvec2_ptr)(src + 2 * (i & 7));
            }
        }
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int i = tid + f * TPB_MM;
            if (i < D * 8) *reinterpret_cast<vdouble2 *>(&s_tile[(i >> 3) * T_STRIDE + 2 * (i & 7)]) = piece[f];
        }
        __syncthreads();
        // ---- compute from LDS ---------------------------------------------------------------------------------
        for (int j = (int)lo; j < (int)hi; ++j) {
            const double v = s_val[j];
            const double *row = &s_tile[(int)s_slot[j] * T_STRIDE];
            const vdouble2 b0 = *reinterpret_cast<const vdouble2 *>(row + c);
            const vdouble2 b1 = *reinterpret_cast<const vdouble2 *>(row + 8 + c);
            acc[0] += v * b0.x;
            acc[1] += v * b0.y;
            acc[2] += v * b1.x;
            acc[3] += v * b1.y;
        }
    } else {
        // ---- gather fallback (the vector kernel's loop; its records alias the tile area) ----------------------
        SpmmEntry *s_ent = reinterpret_cast<SpmmEntry *>(s_tile);
        const int64_t lane_bytes = (int64_t)c * (int64_t)sizeof(double);
        for (int64_t ch = 0; ch < total; ch += T_CH) {
            const int n = (int)((total - ch) < T_CH ? (total - ch) : T_CH);
            __syncthreads();
            for (int i = tid; i < n; i += TPB_MM) {
                const int64_t cc = (int64_t)__builtin_nontemporal_load(colval + p0 + ch + i) - base;
                SpmmEntry e;
                e.val = __builtin_nontemporal_load(nzval + p0 + ch + i);
                e.row = (SPLIT && cc >= n_own) ? B_ghost + (cc - n_own) * bg_rs : B_own + cc * b_rs;
                s_ent[i] = e;
            }
            __syncthreads();
            int j = (int)((lo > ch ? lo : ch) - ch);
            const int e = (int)((hi < ch + n ? hi : ch + n) - ch);
            for (; j + 2 <= e; j += 2) {
                const SpmmEntry e0 = s_ent[j], e1 = s_ent[j + 1];
                const char *s0 = reinterpret_cast<const char *>(e0.row) + lane_bytes;
                const char *s1 = reinterpret_cast<const char *>(e1.row) + lane_bytes;
                const vdouble2 a0 = *(gvec2_ptr)(s0), a1 = *(gvec2_ptr)(s0 + 64);
                const vdouble2 q0 = *(gvec2_ptr)(s1), q1 = *(gvec2_ptr)(s1 + 64);
                acc[0] += e0.val * a0.x; acc[1] += e0.val * a0.y; acc[2] += e0.val * a1.x; acc[3] += e0.val * a1.y;
                acc[0] += e1.val * q0.x; acc[1] += e1.val * q0.y; acc[2] += e1.val * q1.x; acc[3] += e1.val * q1.y;
            }
            if (j < e) {
                const SpmmEntry e0 = s_ent[j];
                const char *s0 = reinterpret_cast<const char *>(e0.row) + lane_bytes;
                const vdouble2 a0 = *(gvec2_ptr)(s0), a1 = *(gvec2_ptr)(s0 + 64);
                acc[0] += e0.val * a0.x; acc[1] += e0.val * a0.y; acc[2] += e0.val * a1.x; acc[3] += e0.val * a1.y;
            }
        }
    }
    if (g < nr) {
        double *dst = C + (r0 + g) * c_rs + c;
        vdouble2 o0, o1;
        o0.x = acc[0]; o0.y = acc[1]; o1.x = acc[2]; o1.y = acc[3];
        *reinterpret_cast<vdouble2 *>(dst) = o0;
        *reinterpret_cast<vdouble2 *>(dst + 8) = o1;
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
                          (split ? reinterpret_cast<uintptr_t>(B_ghost) : 0)) & 15) == 0;
    if (vec_ok) {
        // tuning knobs: entries per inner step (HPCLA_SPMM_VU = 2 | 4) and records per LDS pass
        // (HPCLA_SPMM_CHUNK = 512 | 1536; default by density: short rows fit the small pass)
        static const int vu = [] {
            const char *e = getenv("HPCLA_SPMM_VU");
            return (e && atoi(e) == 4) ? 4 : 2;
        }();
        static const int chunk_env = [] {
            const char *e = getenv("HPCLA_SPMM_CHUNK");
            return e ? atoi(e) : 0;
        }();
        const bool small = chunk_env ? chunk_env == 512 : nnz <= 8 * nrows;
#define HPCLA_SPMM_VEC(SP, VUU, CH, H64)                                                                \
    spmm_rowblock_vec_kernel<I, SP, VUU, CH, H64><<<grid, block, 0, s>>>(                                \
        rowptr, colval, nzval, B_own, b_rs, SP ? B_ghost : nullptr, SP ? bg_rs : 0, SP ? n_own : 0, C, c_rs, \
        nrows, k, index_base, block_list, (uint32_t)launch_blocks)
        // lane->column mapping: 64 contiguous bytes per row per load when k % 16 == 0 (HPCLA_SPMM_HALF64=0: off)
        static const int h64_env = [] {
            const char *e = getenv("HPCLA_SPMM_HALF64");
            return e ? atoi(e) : 1;
        }();
        const bool h64 = h64_env != 0 && (k % 16) == 0;
#define HPCLA_SPMM_VEC2(SP)                                                                             \
    do {                                                                                                \
        if (h64) {                                                                                      \
            if (small) HPCLA_SPMM_VEC(SP, 2, 512, true); else HPCLA_SPMM_VEC(SP, 2, 1536, true);        \
        } else if (small) { if (vu == 4) HPCLA_SPMM_VEC(SP, 4, 512, false); else HPCLA_SPMM_VEC(SP, 2, 512, false); } \
        else { if (vu == 4) HPCLA_SPMM_VEC(SP, 4, 1536, false); else HPCLA_SPMM_VEC(SP, 2, 1536, false); } \
    } while (0)
        // LDS-tiled kernel for short-row matrices at k = 16 (HPCLA_SPMM_TILE=0: off)
        static const int tile_env = [] {
            const char *e = getenv("HPCLA_SPMM_TILE");
            return e ? atoi(e) : 1;
        }();
        if (tile_env != 0 && k == KT && small) {
            if (split)
                spmm_rowblock_tile_kernel<I, true><<<grid, block, 0, s>>>(rowptr, colval, nzval, B_own, b_rs, B_ghost, bg_rs,
                                                                        n_own, C, c_rs, nrows, index_base, block_list,
                                                                        (uint32_t)launch_blocks);
            else
                spmm_rowblock_tile_kernel<I, false><<<grid, block, 0, s>>>(rowptr, colval, nzval, B_own, b_rs, nullptr, 0, 0,
                                                                         C, c_rs, nrows, index_base, block_list,
                                                                         (uint32_t)launch_blocks);
        } else if (split) HPCLA_SPMM_VEC2(true); else HPCLA_SPMM_VEC2(false);
#undef HPCLA_SPMM_VEC2
#undef HPCLA_SPMM_VEC
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
