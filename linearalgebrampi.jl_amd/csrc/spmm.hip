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
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace hpcla {

constexpr int TPB_MM = 256;
constexpr int RPB_MM = 64;                  // rows per workgroup (the granularity of the callers' block lists)
constexpr int CHUNK_MM = 1984;              // entries staged per pass: 1984 * 16 B = 31 744 B of LDS = 5 workgroups per CU.  (2048 =
                                            // 32 768 B looks like five, runs as four: the generic kernel took 1.088 ms on the 5-point
                                            // matrix x 15 columns, 0.946 now; profiles/r05_lds_footprint.log)
constexpr int KT = 16;                      // columns per tile (one per lane of a group)

// The four rows of a lane-group advance TOGETHER, two entries each per step, so a lane has up to 8
// independent B loads in flight.  (A first version walked the rows one after the other with 4 loads
// in flight: a 5-entry stencil row then costs two latency rounds per row, eight per lane-group; the
// interleaved form measured 1.255 -> 1.041 ms on the 5-point matrix x 16 columns and 1.707 -> 1.616 ms
// on config 5's random pattern, profiles/r01_spmm_variants.log.)  Per row the entries are still
// accumulated one after the other in stored order: same bits.
// GRP = lanes per row = columns per tile (16, 8 or 4: the smallest that holds k, so that a narrow B does not leave most
// lanes of a group idle -- k = 3 on 16 lanes per row ran at 0.12 of peak, profiles/r03_spmm_rate_vs_k_before.log);
// a group owns SL = GRP / 4 rows, a workgroup 64 rows whatever GRP is (the block lists' granularity).
template <typename I, bool SPLIT, int GRP>
__global__ __launch_bounds__(TPB_MM) void spmm_rowblock_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ B_own, int64_t b_rs, int64_t b_cs,
    const double *__restrict__ B_ghost, int64_t bg_rs, int64_t n_own, double *__restrict__ C,
    int64_t c_rs, int64_t c_cs, int64_t nrows, int k, int base,
    const int32_t *__restrict__ block_list, uint32_t nblocks, int accumulate)
{
    __shared__ double s_val[CHUNK_MM];
    __shared__ int64_t s_col[CHUNK_MM];   // element offset of the B row (col * row stride), 64-bit

    const int tid = threadIdx.x;
    constexpr int NGR = TPB_MM / GRP, SL = RPB_MM / NGR;        // groups per workgroup, rows per group
    static_assert(NGR * SL == RPB_MM && SL >= 1, "64 rows per workgroup");
    const int g = tid / GRP, l = tid % GRP;
    const uint32_t b = blockIdx.x;
    const int64_t blk = block_list ? (int64_t)block_list[b] : (int64_t)b;
    const int64_t r0 = blk * RPB_MM;
    const int nr = (int)((nrows - r0) < RPB_MM ? (nrows - r0) : RPB_MM);
    const int64_t p0 = (int64_t)rowptr[r0] - base;
    const int64_t p1 = (int64_t)rowptr[r0 + nr] - base;
    const int64_t total = p1 - p0;

    int64_t lo[SL], hi[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int r = g + s * NGR;
        lo[s] = hi[s] = 0;
        if (r < nr) {
            lo[s] = (int64_t)rowptr[r0 + r] - base - p0;
            hi[s] = (int64_t)rowptr[r0 + r + 1] - base - p0;
        }
    }

    for (int kt = 0; kt < k; kt += GRP) {
        const int c = kt + l;
        const bool col_ok = c < k;
        const int64_t c_off = (int64_t)c * b_cs;
        double acc[SL];
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            acc[s] = 0.0;
            // panel order (hpcla_spmm_panel_*): the sums continue from what earlier panels left in C
            const int r = g + s * NGR;
            if (accumulate && r < nr && col_ok) acc[s] = C[(r0 + r) * c_rs + (int64_t)c * c_cs];
        }

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
            int a[SL], len[SL];
            int maxlen = 0;
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int64_t a64 = lo[s] > ch ? lo[s] : ch;
                const int64_t e64 = hi[s] < ch + n ? hi[s] : ch + n;
                a[s] = (int)(a64 - ch);
                len[s] = e64 > a64 ? (int)(e64 - a64) : 0;
                maxlen = len[s] > maxlen ? len[s] : maxlen;
            }
            for (int i = 0; i < maxlen; i += 2) {
                double v[SL][2], bv[SL][2];
                int64_t off[SL][2];
                bool ok[SL][2];
                // LDS reads first (all unconditional, index clamped to a valid entry), then the B
                // loads back to back, then the sums in stored order
#pragma unroll
                for (int s = 0; s < SL; ++s) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        ok[s][u] = i + u < len[s];
                        const int idx = ok[s][u] ? a[s] + i + u : 0;
                        off[s][u] = s_col[idx];
                        v[s][u] = s_val[idx];
                    }
                }
#pragma unroll
                for (int s = 0; s < SL; ++s) {
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
                for (int s = 0; s < SL; ++s) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (ok[s][u]) acc[s] += v[s][u] * bv[s][u];
                }
            }
        }
        // (col_ok is uniform per lane for the whole kt pass: no barrier is skipped by a subset)
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            const int r = g + s * NGR;
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
// the number of gather rounds is not the limit either); an LDS-tiled form that finds a block's distinct B rows
// on the fly (presence bitmap + popcount scan in LDS) and streams them into LDS once: bit-exact, 1.15 ms (five
// barriers and 16 waves per CU per 64-row block; profiles/r02_spmm_lds_tile_on_the_fly.log).  What DID pay is
// the SHAPE of the accesses: the HALF64 lane mapping (+1.5 %) and C written through LDS in whole lines
// (CSTAGE, 0.608 -> 0.573 ms, profiles/r02_spmm_cstage.log) -- for the STORES only: reading B in whole lines
// (eight lanes per row, two rows per lane-group at 256 threads) measured 0.66-0.69 ms
// (profiles/r02_spmm_whole_line_form.log).  A hand-written pure stream of the same byte mix
// at the same 64-row granularity runs in 0.438 ms on the same box (benchmarks/tune/stream_mix.hip).  For
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

// LPR = lanes per row (4, or 2 for k <= 8: a narrow B on four lanes per row leaves half or three quarters of every
// wave idle through the gather loop -- k = 4 ran at 0.38, k = 8 at 0.50 of peak, profiles/r03_spmm_rate_vs_k_before.log);
// the workgroup is 64 * LPR threads and still owns 64 rows, a column tile is 4 * LPR columns.
// CSTAGE: the block's results leave through LDS as whole lines whenever its C rows form ONE contiguous region
// (c_rs == k <= tile width); K16 additionally folds the strides of the device-native shape to constants.
// TAIL2: k is even but not a multiple of 4 (k = 2, 6, 10, 14 ...): the lane that holds the last column pair loads, adds
// and stores only its first 16 bytes -- the reference's own SpMM test has k = 6 (test/test_new_operations.jl:43-59).
// CCOL (round 5): C is COLUMN-major -- the caller's Julia Matrix (src/dense.jl:63) -- while B stays row-major rows: the
// block's 64 x k results leave through LDS anyway (CSTAGE), so they are parked there column by column and written as k
// runs of 64 doubles (512 B) instead of 64 rows of k.  `c_rs` is then the COLUMN stride of C (its leading dimension).
// The unstructured product of a column-major caller drops the conversion of C this way (B is still converted once: an
// unstructured matrix gathers whole B rows).  Not with `accumulate` (the panel order runs on row-major C).
template <typename I, bool SPLIT, int CHUNK_V, bool HALF64, bool CSTAGE, bool K16, int LPR, bool TAIL2, bool CCOL = false>
__global__ __launch_bounds__(64 * LPR) void spmm_rowblock_vec_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ B_own, int64_t b_rs, const double *__restrict__ B_ghost, int64_t bg_rs,
    int64_t n_own, double *__restrict__ C, int64_t c_rs, int64_t nrows, int k, int base,
    const int32_t *__restrict__ block_list, uint32_t nblocks, int accumulate, int group_log2)
{
    constexpr int TPB = 64 * LPR, KTILE = VCPL * LPR, VU = 2;
    static_assert(!HALF64 || LPR == 4, "the 64-contiguous-bytes lane mapping needs four lanes per row");
    static_assert(!K16 || (CSTAGE && HALF64), "K16 is the device-native shape");
    static_assert(!TAIL2 || !HALF64, "the 64-contiguous-bytes mapping needs whole 16-column tiles");
    static_assert(!CCOL || (CSTAGE && LPR == 4), "the column-major store goes through the LDS tile");
    __shared__ SpmmEntry s_ent[CHUNK_V];

    // ODD k (round 6): lanes own column PAIRS, so an odd k is run as k + 1 columns -- the second column of the last pair is the
    // padding double of the (even, > k: checked by the launcher) row pitch: loaded with its pair, summed in a register of its
    // own, never stored (`kr`, the real column count, masks the stores).  Every real column is still one running sum in stored
    // order: same bits.  From here on `k` is even.
    const int kr = k;
    k += k & 1;
    if (K16) {
        // the device-native shape (k = 16 -- or 15 on the pitch 16 --, B / ghost / C rows of exactly 16 doubles: checked by the
        // launcher): strides and the column-tile loop fold to constants (measured against a k = 16-only copy of this kernel in
        // the tuning harness, benchmarks/tune/spmm_variants.hip MODE 0: the generic form ran 3 % behind it)
        k = KT; b_rs = KT; bg_rs = KT;
        if (!CCOL) c_rs = KT;
    }
    const int tid = threadIdx.x;
    const int g = tid / LPR, l = tid % LPR;   // g = row of the block (0..63)
    const uint32_t b = blockIdx.x;
    // XCD-grouped order of a contiguous launch (spmv.hip xcd_group_index), an EXPERIMENT switch (HPCLA_SPMM_XCD_GROUP):
    // on the 5-point matrix every group size from 32 to 1024 blocks loses (B rows of +-64-block neighbours land in other
    // L2s: +2.5 ... +20 %), 8 is neutral; config 5's random pattern gains 2.6 % at 64-256 (profiles/r03_spmm_xcd_group.log).
    // Default: the natural order.
    // Round 4: the order is MEASURED per structure at plan time (hpcla_spmm_tune_block_order_*), natural unless a grouped
    // one is >= 1 % faster, and it applies to the POSITIONS of a block list as well (config 5 at N > 1: every block is a
    // boundary block and arrives through the list).
    const int glog = group_log2;
    int64_t blk = (int64_t)b;
    if (glog > 0) {
        const int64_t span = (int64_t)1 << (3 + glog), nb = (int64_t)nblocks;
        if (blk < nb - (nb & (span - 1))) {
            const int64_t xcd = blk & 7, q = blk >> 3;
            blk = ((((q >> glog) << 3) + xcd) << glog) + (q & (((int64_t)1 << glog) - 1));
        }
    }
    if (block_list) blk = (int64_t)block_list[blk];
    const int64_t r0 = blk * RPB_MM;
    const int nr = (int)((nrows - r0) < RPB_MM ? (nrows - r0) : RPB_MM);
    const int64_t p0 = (int64_t)rowptr[r0] - base;
    const int64_t p1 = (int64_t)rowptr[r0 + nr] - base;
    const int64_t total = p1 - p0;
    // this lane's row bounds stay RAW until their first use behind the staging barrier: subtracting here would put a
    // full memory round trip (1.3 us of a workgroup's 7.4, profiles/r03_spmm_ablations_and_candidates.txt) in front of the
    // staging loads instead of under them (stencil 0.5220 -> 0.5150 ms, profiles/r03_spmm_late_row_bounds.log)
    I rlo = 0, rhi = 0;
    if (g < nr) {
        rlo = rowptr[r0 + g];
        rhi = rowptr[r0 + g + 1];
    }

    // A block whose entries fit ONE pass is staged once for ALL column tiles (k > one tile: the records are read-only
    // across tiles); longer blocks re-stage per tile and pass (the running sums of a tile live in registers).
    const bool single = total <= CHUNK_V;
    if (single) {
        for (int i = tid; i < (int)total; i += TPB) {
            const int64_t col = (int64_t)__builtin_nontemporal_load(colval + p0 + i) - base;
            SpmmEntry e;
            e.val = __builtin_nontemporal_load(nzval + p0 + i);
            e.row = (SPLIT && col >= n_own) ? B_ghost + (col - n_own) * bg_rs : B_own + col * b_rs;
            s_ent[i] = e;
        }
        __syncthreads();
    }
    for (int kt = 0; kt < k; kt += KTILE) {
        // lane -> columns.  Default: four adjacent columns (32 contiguous bytes of the B row, two 16-byte loads).
        // HALF64 (full 16-column tiles only): columns {2l, 2l+1} and {8+2l, 8+2l+1}, i.e. the four lanes of a row
        // read 64 CONTIGUOUS bytes per load instruction (one half line each) instead of every other 16 bytes
        // of the whole line.
        const int c = HALF64 ? kt + 2 * l : kt + VCPL * l;
        const bool col_ok = c < k;
        const bool two = !TAIL2 || c + 2 < k;                              // this lane owns a second column pair
        constexpr int SECOND = HALF64 ? 64 : 16;                           // byte offset of the lane's second load
        const int64_t lane_bytes = (int64_t)c * (int64_t)sizeof(double);   // this lane's slice of every B row
        double acc[VCPL];
#pragma unroll
        for (int q = 0; q < VCPL; ++q) acc[q] = 0.0;
        if (!CCOL && accumulate && g < nr && col_ok) {
            // panel order (hpcla_spmm_panel_*): this lane's four sums continue from what earlier panels left in C
            const double *cur = C + (r0 + g) * c_rs + c;
            const vdouble2 c0 = *reinterpret_cast<const vdouble2 *>(cur);
            vdouble2 c1 = (vdouble2)(0.0);
            if (two) c1 = *reinterpret_cast<const vdouble2 *>(cur + SECOND / 8);
            acc[0] = c0.x; acc[1] = c0.y; acc[2] = c1.x; acc[3] = c1.y;
        }

        for (int64_t ch = 0; ch < total; ch += CHUNK_V) {
            const int n = (int)((total - ch) < CHUNK_V ? (total - ch) : CHUNK_V);
            if (!single) {
                __syncthreads();   // previous pass finished reading LDS
                for (int i = tid; i < n; i += TPB) {
                    const int64_t col = (int64_t)__builtin_nontemporal_load(colval + p0 + ch + i) - base;
                    SpmmEntry e;
                    e.val = __builtin_nontemporal_load(nzval + p0 + ch + i);
                    e.row = (SPLIT && col >= n_own) ? B_ghost + (col - n_own) * bg_rs : B_own + col * b_rs;
                    s_ent[i] = e;
                }
                __syncthreads();
            }
            if (!col_ok) continue;
            const int64_t lo = g < nr ? (int64_t)rlo - base - p0 : 0;
            const int64_t hi = g < nr ? (int64_t)rhi - base - p0 : 0;
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
                    b1[u] = (vdouble2)(0.0);
                    if (two) b1[u] = *(gvec2_ptr)(src + SECOND);       // (never reads past the end of the last B row)
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
                vdouble2 b1 = (vdouble2)(0.0);
                if (two) b1 = *(gvec2_ptr)(src + SECOND);
                acc[0] += en.val * b0.x;
                acc[1] += en.val * b0.y;
                acc[2] += en.val * b1.x;
                acc[3] += en.val * b1.y;
            }
        }
        if (CCOL) {
            // column-major C: park the tile column by column (s_c[c * 64 + g]; the four lanes of a row write four different
            // columns at the same row offset: a 4-way bank conflict on 4 writes per lane, nothing next to the gathers),
            // then every column leaves as one run of nr doubles -- a wave's store instruction covers two columns' 512 bytes
            __syncthreads();                               // every lane has finished reading the records
            double *s_c = reinterpret_cast<double *>(s_ent);
            static_assert(sizeof(SpmmEntry) * CHUNK_V >= sizeof(double) * RPB_MM * KT, "C tile fits the record area");
            if (col_ok) {                                  // (odd k: the padding column k <= 15 lands in the tile and stays there)
                s_c[c * RPB_MM + g] = acc[0];
                s_c[(c + 1) * RPB_MM + g] = acc[1];
                if (two) {
                    s_c[(c + SECOND / 8) * RPB_MM + g] = acc[2];
                    s_c[(c + SECOND / 8 + 1) * RPB_MM + g] = acc[3];
                }
            }
            __syncthreads();
            const vdouble2 *src = reinterpret_cast<const vdouble2 *>(s_c);
            const int count = kr * (RPB_MM / 2);           // (the real columns: an odd k's padding column stays in the tile)
            if ((c_rs & 1) == 0) {
                for (int i = tid; i < count; i += TPB) {
                    const int cc = i / (RPB_MM / 2), gp = (i % (RPB_MM / 2)) * 2;
                    double *dst = C + (int64_t)cc * c_rs + r0 + gp;
                    if (gp + 1 < nr) __builtin_nontemporal_store(src[i], reinterpret_cast<vdouble2 *>(dst));
                    else if (gp < nr) __builtin_nontemporal_store(src[i].x, dst);
                }
            } else {                                       // odd leading dimension: columns start 8-byte aligned only
                for (int i = tid; i < kr * RPB_MM; i += TPB) {
                    const int cc = i / RPB_MM, gg = i % RPB_MM;
                    if (gg < nr) __builtin_nontemporal_store(s_c[i], C + (int64_t)cc * c_rs + r0 + gg);
                }
            }
        } else
        if (CSTAGE) {
            // C through LDS (k <= one column tile, C rows contiguous: c_rs == k): the block's 64 x k results form ONE
            // contiguous region of C (8 KiB at k = 16); written from the accumulators, a store instruction covers 16
            // half lines, written from LDS in linear order it covers 8 whole lines
            __syncthreads();                               // every lane has finished reading the records
            double *s_c = reinterpret_cast<double *>(s_ent);
            static_assert(sizeof(SpmmEntry) * CHUNK_V >= sizeof(double) * RPB_MM * KT, "C tile fits the record area");
            if (col_ok) {
                vdouble2 o0, o1;
                o0.x = acc[0]; o0.y = acc[1]; o1.x = acc[2]; o1.y = acc[3];
                // odd k (the block's C rows are contiguous on the pitch k = kr + 1): the padding column leaves as 0.0 with its
                // pair, so that the region is still written in WHOLE lines -- masking it out cost 0.72 ms against 0.52 on the
                // 5-point matrix x 15 (a partial last sector per 128-byte row; profiles/r06_spmm_odd_k.log)
                if (c + 1 == kr) o0.y = 0.0;
                if (c + SECOND / 8 + 1 == kr) o1.y = 0.0;
                *reinterpret_cast<vdouble2 *>(s_c + g * k + c) = o0;
                if (two) *reinterpret_cast<vdouble2 *>(s_c + g * k + c + SECOND / 8) = o1;
            }
            __syncthreads();
            vdouble2 *dst = reinterpret_cast<vdouble2 *>(C + r0 * c_rs);
            const vdouble2 *src = reinterpret_cast<const vdouble2 *>(s_c);
            const int count = nr * (k / 2);
            for (int i = tid; i < count; i += TPB)
                // non-temporal: C is never re-read by this kernel; measured -1.4 % (stencil) / -1.2 % (config 5's
                // pattern) against plain stores, benchmarks/tune_spmm.py MODE 21 (profiles/r03_spmm_ablations_and_candidates.txt)
                __builtin_nontemporal_store(src[i], dst + i);
        } else
        if (g < nr && col_ok) {
            double *dst = C + (r0 + g) * c_rs + c;
            vdouble2 o0, o1;
            o0.x = acc[0]; o0.y = acc[1]; o1.x = acc[2]; o1.y = acc[3];
            // (c + 1 == k, c + 3 == k: the last pair of an odd k -- its second column does not exist in C)
            if (c + 1 < kr) *reinterpret_cast<vdouble2 *>(dst) = o0;
            else dst[0] = o0.x;
            if (two) {
                if (c + SECOND / 8 + 1 < kr) *reinterpret_cast<vdouble2 *>(dst + SECOND / 8) = o1;
                else dst[SECOND / 8] = o1.x;
            }
        }
    }
}

// ---- RUN TILES (round 4): B rows of a block's contiguous column runs staged into LDS by LDS-DMA ----------------------
// For a banded / stencil matrix the 64 rows of a workgroup touch only a few CONTIGUOUS runs of B rows (the 5-point matrix
// on an nx-wide grid: [r0 - nx, r0 - nx + 64), [r0 - 1, r0 + 65), [r0 + nx, r0 + nx + 64) = 194 rows) although its
// entries name 320 of them; the gather form fetches every entry's row through the vector-memory pipe separately
// (5.24 M gather wave-instructions of 1 KiB: profiles/r03_spmm_ablations_and_candidates.txt -- the gather path adds
// almost its full pipe time).  Here the runs are found ONCE per structure (spmm_runs_build_kernel: sort the block's
// columns, cut where they stop being consecutive, and at the own / ghost boundary), and per product:
//   * the descriptor and the block's row pointers leave together (nothing of the B path depends on A);
//   * the runs' B rows come straight into LDS by LDS-DMA (global_load_lds_dwordx4: one wave-instruction = 8 whole
//     128-byte rows, per-lane SOURCE address, no VGPR destination) next to the block's A entries (coalesced);
//   * every entry is resolved once to the LDS offset of its B row; four lanes per row then multiply out of LDS in
//     stored order -- one sequential sum per C(r, c), the reference's bits (src/sparse.jl:2391-2413);
//   * C leaves through LDS (the tile's space) as whole lines with non-temporal stores.
// Harness (benchmarks/tune_spmm.py MODE 30, profiles/r04_spmm_run_tiles.log): 0.4662 ms against 0.5167 for the gather
// form on the 5-point matrix x 16 = 0.72 of peak; through registers instead of LDS-DMA 0.7165; 32-row blocks 0.4930.
// Shape: k = 16, row-major B / ghost / C of exactly 16 doubles per row, <= RUNS_MAX runs and <= RUNS_TILE_ROWS distinct B
// rows and <= RUNS_EMAX entries per 64-row block; a block that does not fit takes a plain per-entry path (the host layer
// uses this kernel only when >= 99 % of the blocks fit).
constexpr int RUNS_MAX = 4, RUNS_TILE_ROWS = 200, RUNS_EMAX = 512;

struct __attribute__((aligned(16))) SpmmRunDesc {
    int32_t start[RUNS_MAX];     // first split column of run i (0-based)
    int32_t len[RUNS_MAX];       // rows in run i (0: unused); len[0] < 0: the block does not fit
};

template <typename I, bool SPLIT>
__global__ __launch_bounds__(TPB_MM) void spmm_rowblock_runs_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ B_own, const double *__restrict__ B_ghost, int64_t n_own, double *__restrict__ C,
    int64_t nrows, int base, const SpmmRunDesc *__restrict__ runs, const int32_t *__restrict__ block_list,
    uint32_t nblocks, int group_log2)
{
    __shared__ __attribute__((aligned(16))) double s_tile[RUNS_TILE_ROWS * KT];   // the runs' B rows back to back; then the C tile
    __shared__ int32_t s_off[RUNS_EMAX];                                          // per entry: byte offset of its B row in s_tile
    __shared__ double s_val[RUNS_EMAX];

    const int tid = threadIdx.x, g = tid / VG, l = tid % VG;
    int64_t blk = (int64_t)blockIdx.x;
    if (group_log2 > 0) {                                  // XCD-grouped order of the launch's positions (as in the gather kernel)
        const int64_t span = (int64_t)1 << (3 + group_log2), nb = (int64_t)nblocks;
        if (blk < nb - (nb & (span - 1))) {
            const int64_t xcd = blk & 7, q = blk >> 3;
            blk = ((((q >> group_log2) << 3) + xcd) << group_log2) + (q & (((int64_t)1 << group_log2) - 1));
        }
    }
    if (block_list) blk = (int64_t)block_list[blk];
    const int64_t r0 = blk * RPB_MM;
    const int nr = (int)((nrows - r0) < RPB_MM ? (nrows - r0) : RPB_MM);
    const SpmmRunDesc d = runs[blk];
    const int64_t p0 = (int64_t)rowptr[r0] - base, p1 = (int64_t)rowptr[r0 + nr] - base;
    I rlo = 0, rhi = 0;                                    // raw: first used behind the staging barrier
    if (g < nr) { rlo = rowptr[r0 + g]; rhi = rowptr[r0 + g + 1]; }
    const int total = (int)(p1 - p0);
    const int o1 = d.len[0], o2 = o1 + d.len[1], o3 = o2 + d.len[2], T = o3 + d.len[3];      // tile row offsets of the runs

    if (d.len[0] < 0) {
        // the block does not fit the tile (workgroup-uniform; rare by the host layer's choice): per-entry gathers straight
        // from global memory, four lanes per row, same order of the sums
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (g < nr) {
            for (int64_t j = (int64_t)rlo - base; j < (int64_t)rhi - base; ++j) {
                const double v = nzval[j];
                const int64_t c = (int64_t)colval[j] - base;
                const double *row = (SPLIT && c >= n_own) ? B_ghost + (c - n_own) * KT : B_own + c * KT;
                const vdouble2 b0 = *reinterpret_cast<const vdouble2 *>(row + 2 * l);
                const vdouble2 b1 = *reinterpret_cast<const vdouble2 *>(row + 8 + 2 * l);
                acc[0] += v * b0.x; acc[1] += v * b0.y; acc[2] += v * b1.x; acc[3] += v * b1.y;
            }
            vdouble2 q0, q1;
            q0.x = acc[0]; q0.y = acc[1]; q1.x = acc[2]; q1.y = acc[3];
            *reinterpret_cast<vdouble2 *>(C + (r0 + g) * KT + 2 * l) = q0;
            *reinterpret_cast<vdouble2 *>(C + (r0 + g) * KT + 8 + 2 * l) = q1;
        }
        return;
    }

    // B rows -> LDS: piece q = 8 tile rows = 1 KiB = one wave-instruction (lanes 8 r .. 8 r + 7 carry row r of the piece)
    {
        const int wave = tid >> 6, lane = tid & 63;
        const int npieces = (T + 7) >> 3;
        for (int q = wave; q < npieces; q += TPB_MM / 64) {
            int t = q * 8 + (lane >> 3);
            if (t >= T) t = T - 1;                          // the ragged last piece re-reads the last row (inside the tile's capacity)
            const int64_t sc = t < o1 ? (int64_t)d.start[0] + t
                                      : (t < o2 ? (int64_t)d.start[1] + (t - o1) : (t < o3 ? (int64_t)d.start[2] + (t - o2) : (int64_t)d.start[3] + (t - o3)));
            const double *src = ((SPLIT && sc >= n_own) ? B_ghost + (sc - n_own) * KT : B_own + sc * KT) + (lane & 7) * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(s_tile + q * 8 * KT), 16, 0, 0);
        }
    }
    // A entries -> {LDS offset of the B row, value}: both of a thread's entries are requested UNCONDITIONALLY (index
    // clamped to the block's last entry), so the four loads leave back to back -- a predicated load is a branch, and with an
    // LDS-DMA in flight the compiler waits vmcnt(0) at the first use behind it: one full round trip per entry otherwise
    if (total > 0) {                                        // workgroup-uniform
        static_assert(RUNS_EMAX == 2 * TPB_MM, "two entries per thread");
        const int i0 = tid, i1 = tid + TPB_MM;
        const int j0 = i0 < total ? i0 : total - 1, j1 = i1 < total ? i1 : total - 1;
        const I cr0 = __builtin_nontemporal_load(colval + p0 + j0);
        const double v0 = __builtin_nontemporal_load(nzval + p0 + j0);
        const I cr1 = __builtin_nontemporal_load(colval + p0 + j1);
        const double v1 = __builtin_nontemporal_load(nzval + p0 + j1);
        auto tile_row = [&](int64_t c) -> int {
            if (d.len[3] > 0 && c >= d.start[3]) return o3 + (int)(c - d.start[3]);
            if (d.len[2] > 0 && c >= d.start[2]) return o2 + (int)(c - d.start[2]);
            if (d.len[1] > 0 && c >= d.start[1]) return o1 + (int)(c - d.start[1]);
            return (int)(c - d.start[0]);
        };
        const int t0 = tile_row((int64_t)cr0 - base), t1 = tile_row((int64_t)cr1 - base);
        if (i0 < total) { s_off[i0] = t0 * (KT * 8); s_val[i0] = v0; }
        if (i1 < total) { s_off[i1] = t1 * (KT * 8); s_val[i1] = v1; }
    }
    __syncthreads();                                        // (drains the LDS-DMA too: vmcnt(0), then the barrier)

    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (g < nr) {
        const char *tile = reinterpret_cast<const char *>(s_tile) + l * 16;
        int j = (int)((int64_t)rlo - base - p0);
        const int e = (int)((int64_t)rhi - base - p0);
        for (; j + 2 <= e; j += 2) {
            const int oa = s_off[j], ob = s_off[j + 1];
            const double va = s_val[j], vb = s_val[j + 1];
            const vdouble2 a0 = *reinterpret_cast<const vdouble2 *>(tile + oa), a1 = *reinterpret_cast<const vdouble2 *>(tile + oa + 64);
            const vdouble2 b0 = *reinterpret_cast<const vdouble2 *>(tile + ob), b1 = *reinterpret_cast<const vdouble2 *>(tile + ob + 64);
            acc[0] += va * a0.x; acc[1] += va * a0.y; acc[2] += va * a1.x; acc[3] += va * a1.y;
            acc[0] += vb * b0.x; acc[1] += vb * b0.y; acc[2] += vb * b1.x; acc[3] += vb * b1.y;
        }
        for (; j < e; ++j) {
            const int oa = s_off[j];
            const double va = s_val[j];
            const vdouble2 a0 = *reinterpret_cast<const vdouble2 *>(tile + oa), a1 = *reinterpret_cast<const vdouble2 *>(tile + oa + 64);
            acc[0] += va * a0.x; acc[1] += va * a0.y; acc[2] += va * a1.x; acc[3] += va * a1.y;
        }
    }
    __syncthreads();                                        // everybody has finished reading the tile
    double *s_c = s_tile;
    vdouble2 q0, q1;
    q0.x = acc[0]; q0.y = acc[1]; q1.x = acc[2]; q1.y = acc[3];
    *reinterpret_cast<vdouble2 *>(s_c + g * KT + 2 * l) = q0;         // columns {2l, 2l+1} and {8+2l, 8+2l+1} (the HALF64 mapping)
    *reinterpret_cast<vdouble2 *>(s_c + g * KT + 8 + 2 * l) = q1;
    __syncthreads();
    vdouble2 *dst = reinterpret_cast<vdouble2 *>(C + r0 * KT);
    const vdouble2 *srcl = reinterpret_cast<const vdouble2 *>(s_c);
#pragma unroll
    for (int u = 0; u < (RPB_MM * KT / 2) / TPB_MM; ++u) {
        const int i = tid + u * TPB_MM;
        if (i < nr * (KT / 2)) __builtin_nontemporal_store(srcl[i], dst + i);
    }
}

// RUN TILES for a COLUMN-major caller (round 5; Julia's Matrix, src/dense.jl:63): B_own and C column-major, the ghost rows the
// halo plan's ordinary row-major segment.  Same descriptors, same sums in the same order (the reference's bits,
// src/sparse.jl:2391-2413), other shape:
//   * a run of ONE column is a contiguous piece of memory, but it starts on any 8-byte boundary, which left the first form of
//     this kernel global_load_lds's 4-byte variant (144 wave-instructions per workgroup, 1.78 ms on the 5-point matrix x 16:
//     profiles/r04_colmajor.log).  Here every own run is widened to an EVEN first row and an even end (at most two rows more
//     per run: B_own 16-byte aligned, ldb even -- the launcher checks), so the 16-byte LDS-DMA applies: the tile is the
//     block's runs of column 0, then of column 1, ... back to back (Tpad rows each), 64 row pairs per wave-instruction
//     (lanes past the tile's end are masked off: a masked lane neither loads nor writes its LDS slot);
//   * lanes = rows, waves = column quads: a lane reads its entries' four B values out of LDS (64 lanes on consecutive tile
//     rows of one column: no bank conflicts), and C leaves as 512-byte runs straight from the registers;
//   * ghost columns are not staged: their entries (boundary blocks only) read the row-major ghost row from memory.
// A block whose widened runs exceed the tile, or would read past row n_own of a column (the last block, n_own odd), takes
// the per-entry path.  The tile keeps RUNS_TILE_ROWS rows: 31 744 bytes of LDS per workgroup = 5 workgroups per CU; at
// 32 768 the fifth does not fit and the product takes 0.625 ms instead of 0.563 (profiles/r05_colmajor_run_tiles.log).
constexpr int RUNS_CM_TILE_ROWS = RUNS_TILE_ROWS;

template <typename I, bool SPLIT>
__global__ __launch_bounds__(TPB_MM) void spmm_runs_colmajor_kernel(
    const I *__restrict__ rowptr, const I *__restrict__ colval, const double *__restrict__ nzval,
    const double *__restrict__ B_own, int64_t ldb, const double *__restrict__ B_ghost, int64_t ldg, int64_t n_own,
    double *__restrict__ C, int64_t ldc, int64_t nrows, int base, const SpmmRunDesc *__restrict__ runs,
    const int32_t *__restrict__ block_list, uint32_t nblocks, int group_log2)
{
    static_assert(KT == 4 * (TPB_MM / 64), "one wave per column quad");
    static_assert(RPB_MM == 64, "lanes = rows");
    __shared__ __attribute__((aligned(16))) double s_tile[RUNS_CM_TILE_ROWS * KT];   // column c of the widened runs at [c * Tpad, (c + 1) * Tpad)
    __shared__ int32_t s_off[RUNS_EMAX];                                             // per entry: tile row, or -(ghost row) - 1
    __shared__ double s_val[RUNS_EMAX];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar loop control below)
    int64_t blk = (int64_t)blockIdx.x;
    if (group_log2 > 0) {                                  // XCD-grouped order of the launch's positions (as in the row-major kernel)
        const int64_t span = (int64_t)1 << (3 + group_log2), nb = (int64_t)nblocks;
        if (blk < nb - (nb & (span - 1))) {
            const int64_t xcd = blk & 7, q = blk >> 3;
            blk = ((((q >> group_log2) << 3) + xcd) << group_log2) + (q & (((int64_t)1 << group_log2) - 1));
        }
    }
    if (block_list) blk = (int64_t)block_list[blk];
    const int64_t r0 = blk * RPB_MM;
    const int nr = (int)((nrows - r0) < RPB_MM ? (nrows - r0) : RPB_MM);
    const SpmmRunDesc d = runs[blk];
    const int64_t p0 = (int64_t)rowptr[r0] - base, p1 = (int64_t)rowptr[r0 + nr] - base;
    I rlo = 0, rhi = 0;
    if (lane < nr) { rlo = rowptr[r0 + lane]; rhi = rowptr[r0 + lane + 1]; }
    const int total = (int)(p1 - p0);

    // the own runs, widened to even first rows and even ends; o[i]: tile row where run i begins
    int64_t a[RUNS_MAX];
    int o[RUNS_MAX + 1];
    bool fits = d.len[0] >= 0;
    o[0] = 0;
#pragma unroll
    for (int i = 0; i < RUNS_MAX; ++i) {
        const bool own = d.len[i] > 0 && (!SPLIT || (int64_t)d.start[i] < n_own);
        const int64_t ai = (int64_t)d.start[i] & ~(int64_t)1, ei = ((int64_t)d.start[i] + d.len[i] + 1) & ~(int64_t)1;
        a[i] = ai;
        o[i + 1] = o[i] + (own ? (int)(ei - ai) : 0);
        if (own && ei > n_own) fits = false;               // the widened run would read past the column's last row
    }
    const int Tpad = o[RUNS_MAX];
    if (Tpad > RUNS_CM_TILE_ROWS) fits = false;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    double *Cq = C + (int64_t)(4 * wave) * ldc + r0 + lane;

    if (!fits) {
        // workgroup-uniform, rare (the host layers use this kernel when >= 99 % of the blocks fit): per-entry reads from memory
        if (lane < nr) {
            const double *Bq = B_own + (int64_t)(4 * wave) * ldb;
            for (int64_t j = (int64_t)rlo - base; j < (int64_t)rhi - base; ++j) {
                const double v = nzval[j];
                const int64_t c = (int64_t)colval[j] - base;
                double b0, b1, b2, b3;
                if (SPLIT && c >= n_own) {
                    const double *row = B_ghost + (c - n_own) * ldg + 4 * wave;
                    b0 = row[0]; b1 = row[1]; b2 = row[2]; b3 = row[3];
                } else {
                    b0 = Bq[c]; b1 = Bq[ldb + c]; b2 = Bq[2 * ldb + c]; b3 = Bq[3 * ldb + c];
                }
                acc[0] += v * b0; acc[1] += v * b1; acc[2] += v * b2; acc[3] += v * b3;
            }
            Cq[0] = acc[0]; Cq[ldc] = acc[1]; Cq[2 * ldc] = acc[2]; Cq[3 * ldc] = acc[3];
        }
        return;
    }

    // B -> LDS: one wave-instruction = 64 row pairs (1 KiB) of the tile's linear space; pair P = rows {2 pr, 2 pr + 1} of column c
    {
        const unsigned halfT = (unsigned)Tpad >> 1, pairs = halfT * KT;
        const int npieces = (int)((pairs + 63) >> 6);
        for (int q = wave; q < npieces; q += TPB_MM / 64) {
            const unsigned P = (unsigned)q * 64 + lane;
            if (P < pairs) {
                const unsigned c = P / halfT;
                const int t = (int)(2 * (P - c * halfT));
                const int64_t sr = t < o[1] ? a[0] + t : (t < o[2] ? a[1] + (t - o[1]) : (t < o[3] ? a[2] + (t - o[2]) : a[3] + (t - o[3])));
                const double *src = B_own + (int64_t)c * ldb + sr;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(s_tile + q * 128), 16, 0, 0);
            }
        }
    }
    // A entries -> {tile row | ghost row, value}; both of a thread's entries requested unconditionally (see the row-major kernel)
    if (total > 0) {                                        // workgroup-uniform
        static_assert(RUNS_EMAX == 2 * TPB_MM, "two entries per thread");
        const int i0 = tid, i1 = tid + TPB_MM;
        const int j0 = i0 < total ? i0 : total - 1, j1 = i1 < total ? i1 : total - 1;
        const I cr0 = __builtin_nontemporal_load(colval + p0 + j0);
        const double v0 = __builtin_nontemporal_load(nzval + p0 + j0);
        const I cr1 = __builtin_nontemporal_load(colval + p0 + j1);
        const double v1 = __builtin_nontemporal_load(nzval + p0 + j1);
        auto tile_row = [&](int64_t c) -> int {
            if (SPLIT && c >= n_own) return -(int)(c - n_own) - 1;
            if (d.len[3] > 0 && c >= d.start[3]) return o[3] + (int)(c - a[3]);
            if (d.len[2] > 0 && c >= d.start[2]) return o[2] + (int)(c - a[2]);
            if (d.len[1] > 0 && c >= d.start[1]) return o[1] + (int)(c - a[1]);
            return (int)(c - a[0]);
        };
        const int t0 = tile_row((int64_t)cr0 - base), t1 = tile_row((int64_t)cr1 - base);
        if (i0 < total) { s_off[i0] = t0; s_val[i0] = v0; }
        if (i1 < total) { s_off[i1] = t1; s_val[i1] = v1; }
    }
    __syncthreads();                                        // (drains the LDS-DMA too)

    if (lane < nr) {
        const double *q0 = s_tile + (4 * wave) * Tpad, *q1 = q0 + Tpad, *q2 = q1 + Tpad, *q3 = q2 + Tpad;
        auto fetch = [&](int off, double &b0, double &b1, double &b2, double &b3) {
            if (!SPLIT || off >= 0) { b0 = q0[off]; b1 = q1[off]; b2 = q2[off]; b3 = q3[off]; }
            else {
                const double *row = B_ghost + (int64_t)(-off - 1) * ldg + 4 * wave;
                b0 = row[0]; b1 = row[1]; b2 = row[2]; b3 = row[3];
            }
        };
        int j = (int)((int64_t)rlo - base - p0);
        const int e = (int)((int64_t)rhi - base - p0);
        for (; j + 2 <= e; j += 2) {
            const int oa = s_off[j], ob = s_off[j + 1];
            const double va = s_val[j], vb = s_val[j + 1];
            double a0, a1, a2, a3, b0, b1, b2, b3;
            fetch(oa, a0, a1, a2, a3);
            fetch(ob, b0, b1, b2, b3);
            acc[0] += va * a0; acc[1] += va * a1; acc[2] += va * a2; acc[3] += va * a3;
            acc[0] += vb * b0; acc[1] += vb * b1; acc[2] += vb * b2; acc[3] += vb * b3;
        }
        for (; j < e; ++j) {
            const int oa = s_off[j];
            const double va = s_val[j];
            double a0, a1, a2, a3;
            fetch(oa, a0, a1, a2, a3);
            acc[0] += va * a0; acc[1] += va * a1; acc[2] += va * a2; acc[3] += va * a3;
        }
    }
    if (lane < nr) {
        __builtin_nontemporal_store(acc[0], Cq);
        __builtin_nontemporal_store(acc[1], Cq + ldc);
        __builtin_nontemporal_store(acc[2], Cq + 2 * ldc);
        __builtin_nontemporal_store(acc[3], Cq + 3 * ldc);
    }
}

// Plan time, once per structure: the run descriptor of every 64-row block.  One workgroup per block: the block's (split)
// columns are sorted in LDS (bitonic, <= RUNS_EMAX keys), an element starts a run where it is neither equal to nor the
// successor of its predecessor, or where it crosses the own / ghost boundary; runs are ranked by wave ballots.
template <typename I>
__global__ __launch_bounds__(TPB_MM) void spmm_runs_build_kernel(const I *__restrict__ rowptr, const I *__restrict__ colval,
                                                                 int64_t nrows, int base, int64_t n_own,
                                                                 SpmmRunDesc *__restrict__ runs,
                                                                 unsigned long long *__restrict__ n_fit, int banded_max_runs)
{
    // runs == nullptr: COUNT ONLY (hpcla_spmm_banded_blocks_*): n_fit counts the blocks whose <= RUNS_EMAX entries touch at most
    // `banded_max_runs` contiguous runs of columns -- the structures on which a lanes = rows kernel reads a column-major block
    // in contiguous pieces (colmajor.hip); no descriptor is written
    __shared__ int64_t s_key[RUNS_EMAX];
    __shared__ unsigned long long s_mask[RUNS_EMAX / 64];
    __shared__ int s_start_idx[RUNS_MAX + 1];
    const int tid = threadIdx.x;
    const int64_t blk = blockIdx.x, r0 = blk * RPB_MM;
    const int nr = (int)((nrows - r0) < RPB_MM ? (nrows - r0) : RPB_MM);
    const int64_t p0 = (int64_t)rowptr[r0] - base, p1 = (int64_t)rowptr[r0 + nr] - base;
    const int64_t total = p1 - p0;
    SpmmRunDesc d;
    for (int i = 0; i < RUNS_MAX; ++i) { d.start[i] = 0; d.len[i] = 0; }
    if (total > RUNS_EMAX) {                               // workgroup-uniform
        if (tid == 0 && runs) { d.len[0] = -1; runs[blk] = d; }
        return;
    }
    constexpr int64_t BIG = (int64_t)1 << 62;
    for (int i = tid; i < RUNS_EMAX; i += TPB_MM) s_key[i] = i < total ? (int64_t)colval[p0 + i] - base : BIG;
    __syncthreads();
    for (int k = 2; k <= RUNS_EMAX; k <<= 1)                // bitonic sort, ascending
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < RUNS_EMAX; i += TPB_MM) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int64_t a = s_key[i], b = s_key[ixj];
                    const bool up = (i & k) == 0;
                    if (up ? a > b : a < b) { s_key[i] = b; s_key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // run starts (element e = tid + h * TPB_MM; wave w of half h owns mask word h * 4 + w)
    bool st[2];
    int n_starts = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = tid + h * TPB_MM;
        bool is = false;
        if (e < total) {
            const int64_t c = s_key[e];
            if (e == 0) is = true;
            else {
                const int64_t pc = s_key[e - 1];
                is = (c > pc + 1) || (pc < n_own && c >= n_own);
            }
        }
        st[h] = is;
        const unsigned long long m = __ballot(is);
        if ((tid & 63) == 0) s_mask[h * (TPB_MM / 64) + (tid >> 6)] = m;
        n_starts += __syncthreads_count(is);
    }
    if (tid <= RUNS_MAX) s_start_idx[tid] = (int)total;
    __syncthreads();
    if (!runs) {                                           // count only (workgroup-uniform)
        if (tid == 0 && n_starts <= banded_max_runs) atomicAdd(n_fit, 1ULL);
        return;
    }
    if (n_starts > RUNS_MAX || total == 0) {               // workgroup-uniform (an empty block "fits": nothing to stage)
        if (tid == 0) {
            if (total != 0) d.len[0] = -1;
            else atomicAdd(n_fit, 1ULL);
            runs[blk] = d;
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (st[h]) {
            const int word = h * (TPB_MM / 64) + (tid >> 6);
            int rank = __popcll(s_mask[word] & ((1ULL << (tid & 63)) - 1ULL));
            for (int w = 0; w < word; ++w) rank += __popcll(s_mask[w]);
            s_start_idx[rank] = tid + h * TPB_MM;
        }
    __syncthreads();
    if (tid == 0) {
        int rows = 0;
        for (int r = 0; r < n_starts; ++r) {
            const int64_t s0 = s_key[s_start_idx[r]], s1 = s_key[s_start_idx[r + 1] - 1];
            d.start[r] = (int32_t)s0;
            d.len[r] = (int32_t)(s1 - s0 + 1);
            rows += d.len[r];
            if (s1 > 0x7fffffffLL) rows = RUNS_TILE_ROWS + 1;          // descriptors hold Int32 columns
        }
        if (rows > RUNS_TILE_ROWS) { d.len[0] = -1; }
        else atomicAdd(n_fit, 1ULL);
        runs[blk] = d;
    }
}

// tiled transpose / layout conversion: dst(i,c) = src(i,c), arbitrary (row,col) strides
template <typename T>
__global__ __launch_bounds__(256) void relayout_kernel(const T *__restrict__ src,
                                                       int64_t s_rs, int64_t s_cs,
                                                       T *__restrict__ dst, int64_t d_rs,
                                                       int64_t d_cs, int64_t rows, int64_t cols)
{
    __shared__ T tile[32][33];
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

// spmv.hip
int spmv_split_i32(const int32_t *, const int32_t *, const double *, const double *, const double *, int64_t, double *,
                   int64_t, int64_t, int, const int32_t *, int64_t, void *, double *, int64_t);
int spmv_split_i64(const int64_t *, const int64_t *, const double *, const double *, const double *, int64_t, double *,
                   int64_t, int64_t, int, const int32_t *, int64_t, void *, double *, int64_t);
constexpr int64_t nrows_b_unused = 0;     // n_own of an unsplit product: no column is a ghost (x_ghost == NULL)
static inline int spmv_split_any(const int32_t *rp, const int32_t *cv, const double *nz, const double *x, const double *xg,
                                 int64_t n_own, double *y, int64_t nrows, int64_t nnz, int base, void *stream)
{
    return spmv_split_i32(rp, cv, nz, x, xg, n_own, y, nrows, nnz, base, nullptr, 0, stream, nullptr, -1);
}
static inline int spmv_split_any(const int64_t *rp, const int64_t *cv, const double *nz, const double *x, const double *xg,
                                 int64_t n_own, double *y, int64_t nrows, int64_t nnz, int base, void *stream)
{
    return spmv_split_i64(rp, cv, nz, x, xg, n_own, y, nrows, nnz, base, nullptr, 0, stream, nullptr, -1);
}

// ---- per-matrix block order of the SpMM launches (a performance hint: every order is a bijection) -------------------
static std::mutex g_mm_order_mu;
static std::unordered_map<const void *, int> g_mm_order;       // rowptr device pointer -> log2(group)
static std::atomic<int> g_mm_order_count{0};

static void set_spmm_block_order(const void *rowptr, int group_log2)
{
    std::lock_guard<std::mutex> lock(g_mm_order_mu);
    if (group_log2 <= 0) g_mm_order.erase(rowptr);
    else g_mm_order[rowptr] = group_log2;
    g_mm_order_count.store((int)g_mm_order.size(), std::memory_order_relaxed);
}

static int spmm_group_log2(const void *rowptr)
{
    const char *e = getenv("HPCLA_SPMM_XCD_GROUP");          // experiments: every launch; re-read per launch (the tuning harness switches it between launches)
    if (e) {
        int g = atoi(e), l = 0;
        while (g > 1) { g >>= 1; ++l; }
        return l > 12 ? 12 : l;
    }
    if (g_mm_order_count.load(std::memory_order_relaxed) == 0) return 0;
    std::lock_guard<std::mutex> lock(g_mm_order_mu);
    auto it = g_mm_order.find(rowptr);
    return it == g_mm_order.end() ? 0 : it->second;
}

template <typename I>
static int spmm_launch(const I *rowptr, const I *colval, const double *nzval, const double *B_own,
                       int64_t b_rs, int64_t b_cs, const double *B_ghost, int64_t bg_rs,
                       int64_t n_own, bool split, double *C, int64_t c_rs, int64_t c_cs,
                       int64_t nrows, int64_t nnz, int k, int index_base,
                       const int32_t *block_list, int64_t n_blocks, void *stream, int accumulate = 0)
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
    if (k == 1 && !block_list && !accumulate && b_rs == 1 && c_rs == 1 && (!split || !B_ghost || bg_rs == 1)) {
        // one dense column with unit stride IS a vector: the SpMV kernel computes the same row sums in the same
        // order (the reference's A * B is a column loop over A * x, src/sparse.jl:2391-2413) -- 0.88 ms through the
        // 16-lanes-per-row SpMM form against the SpMV's rate on the 5-point matrix (profiles/r03_spmm_rate_vs_k_*.log)
        return spmv_split_any(rowptr, colval, nzval, B_own, split ? B_ghost : nullptr, split ? n_own : nrows_b_unused, C,
                              nrows, nnz, index_base, stream);
    }
    hipStream_t s = as_stream(stream);
    dim3 grid((uint32_t)launch_blocks), block(TPB_MM);
    // device-native layout: row-major B / C, k even (16-byte column pairs), everything 16-byte aligned
    // ... or row-major B with a COLUMN-major C of up to one column tile (round 5: the column-major caller's unstructured
    // product; CCOL in the kernel): c_rs == 1, c_cs = C's leading dimension
    const bool c_colmajor = c_rs == 1 && c_cs != 1 && !accumulate && c_cs >= nrows;
    const bool c_col = c_colmajor && k <= KT;
    // ODD k (round 6) runs as k + 1 columns when every row pitch is even and LARGER than k: the last column pair's second
    // half is the padding double of the row (read, never stored).  The B / ghost buffers must therefore span rows * pitch
    // doubles (include/hpcla_rocm.h); a row-major C needs the same pitch rule for its 16-byte column pairs.
    const bool odd_ok = (k % 2) == 0 || (b_rs > k && (!split || !B_ghost || bg_rs > k) && (c_colmajor || c_rs > k));
    const bool vec_ok = b_cs == 1 && (c_colmajor || (c_cs == 1 && (c_rs % 2) == 0)) && odd_ok && (b_rs % 2) == 0 &&
                        (!split || (bg_rs % 2) == 0) &&
                        ((reinterpret_cast<uintptr_t>(B_own) | reinterpret_cast<uintptr_t>(C) |
                          (split ? reinterpret_cast<uintptr_t>(B_ghost) : 0)) & 15) == 0;
    if (vec_ok && c_colmajor && !c_col) {
        // column-major C wider than one column tile: one CCOL launch per 16-column tile (B rows are read at a 128-byte
        // column offset, C at whole columns; A is re-streamed per tile -- against the generic strided kernel's one column
        // per lane).  Every C(r, c) is the same running sum.
        for (int kt = 0; kt < k; kt += KT) {
            const int kk = k - kt < KT ? k - kt : KT;
            const int rc = spmm_launch<I>(rowptr, colval, nzval, B_own + kt, b_rs, b_cs, B_ghost ? B_ghost + kt : nullptr, bg_rs,
                                          n_own, split, C + (int64_t)kt * c_cs, c_rs, c_cs, nrows, nnz, kk, index_base,
                                          block_list, n_blocks, stream, 0);
            if (rc != HPCLA_OK) return rc;
        }
        return HPCLA_OK;
    }
    if (vec_ok) {
        // records per LDS pass (HPCLA_SPMM_CHUNK = 512 | 1536; default by density: short rows fit the small pass)
        static const int chunk_env = [] {
            const char *e = getenv("HPCLA_SPMM_CHUNK");
            return e ? atoi(e) : 0;
        }();
        const bool small = chunk_env ? chunk_env == 512 : nnz <= 8 * nrows;
        // lane->column mapping: 64 contiguous bytes per row per load when k % 16 == 0 (HPCLA_SPMM_HALF64=0: off)
        static const int h64_env = [] {
            const char *e = getenv("HPCLA_SPMM_HALF64");
            return e ? atoi(e) : 1;
        }();
        // C through LDS when the block's C rows are one contiguous region (HPCLA_SPMM_CSTAGE=0: off)
        static const int cst_env = [] {
            const char *e = getenv("HPCLA_SPMM_CSTAGE");
            return e ? atoi(e) : 1;
        }();
        // two lanes per row for a narrow B (HPCLA_SPMM_LPR=4: always four)
        static const int lpr_env = [] {
            const char *e = getenv("HPCLA_SPMM_LPR");
            return e ? atoi(e) : 0;
        }();
        const int ke = k + (k & 1);                      // columns the lanes run (odd k: the padding column rides along)
        const int lpr = (ke <= 8 && lpr_env != 4 && !c_col) ? 2 : 4;
        const bool h64 = lpr == 4 && h64_env != 0 && (ke % 16) == 0;
        const bool cstage = c_col || (cst_env != 0 && ke <= 4 * lpr && c_rs == ke);
        const bool k16 = h64 && cstage && ke == KT && b_rs == KT && (!split || bg_rs == KT);
        const bool tail2 = (ke % 4) != 0;                // the last column pair of a row is half a lane's share
        const int glog2 = spmm_group_log2(rowptr);
#define HPCLA_SPMM_VEC(SP, CH, H64, CST, K16F, LPRV)                                                     \
    do {                                                                                                \
        if (!H64 && tail2) HPCLA_SPMM_VECT(SP, CH, H64, CST, K16F, LPRV, (!H64));                       \
        else HPCLA_SPMM_VECT(SP, CH, H64, CST, K16F, LPRV, false);                                      \
    } while (0)
#define HPCLA_SPMM_VECT(SP, CH, H64, CST, K16F, LPRV, T2)                                                \
    spmm_rowblock_vec_kernel<I, SP, CH, H64, CST, K16F, LPRV, (T2)><<<grid, dim3(64 * LPRV), 0, s>>>(     \
        rowptr, colval, nzval, B_own, b_rs, SP ? B_ghost : nullptr, SP ? bg_rs : 0, SP ? n_own : 0, C, c_rs, \
        nrows, k, index_base, block_list, (uint32_t)launch_blocks, accumulate, glog2)
#define HPCLA_SPMM_VECC(SP, CH, H64, K16F, T2)                                                           \
    spmm_rowblock_vec_kernel<I, SP, CH, H64, true, K16F, 4, (T2), true><<<grid, dim3(256), 0, s>>>(       \
        rowptr, colval, nzval, B_own, b_rs, SP ? B_ghost : nullptr, SP ? bg_rs : 0, SP ? n_own : 0, C, c_cs, \
        nrows, k, index_base, block_list, (uint32_t)launch_blocks, 0, glog2)
#define HPCLA_SPMM_VEC1(SP, CH)                                                                          \
    do {                                                                                                \
        if (c_col) {                                                                                    \
            if (k16) HPCLA_SPMM_VECC(SP, CH, true, true, false);                                        \
            else if (tail2) HPCLA_SPMM_VECC(SP, CH, false, false, true);                                \
            else HPCLA_SPMM_VECC(SP, CH, false, false, false);                                          \
        } else                                                                                          \
        if (lpr == 2) { if (cstage) HPCLA_SPMM_VEC(SP, CH, false, true, false, 2); else HPCLA_SPMM_VEC(SP, CH, false, false, false, 2); } \
        else if (k16) HPCLA_SPMM_VEC(SP, CH, true, true, true, 4);                                      \
        else if (h64 && cstage) HPCLA_SPMM_VEC(SP, CH, true, true, false, 4);                           \
        else if (h64) HPCLA_SPMM_VEC(SP, CH, true, false, false, 4);                                    \
        else if (cstage) HPCLA_SPMM_VEC(SP, CH, false, true, false, 4);                                 \
        else HPCLA_SPMM_VEC(SP, CH, false, false, false, 4);                                            \
    } while (0)
#define HPCLA_SPMM_VEC2(SP)                                                                             \
    do { if (small) HPCLA_SPMM_VEC1(SP, 512); else HPCLA_SPMM_VEC1(SP, 1536); } while (0)
        if (split) HPCLA_SPMM_VEC2(true); else HPCLA_SPMM_VEC2(false);
#undef HPCLA_SPMM_VEC2
#undef HPCLA_SPMM_VECC
#undef HPCLA_SPMM_VEC1
#undef HPCLA_SPMM_VEC
#undef HPCLA_SPMM_VECT
    } else {
        // generic strides / any k: lanes per row = the smallest of 4, 8, 16 that holds k (wider B: 16-column tiles)
#define HPCLA_SPMM_GEN(SP, GRP)                                                                                  \
    spmm_rowblock_kernel<I, SP, GRP><<<grid, block, 0, s>>>(                                                    \
        rowptr, colval, nzval, B_own, b_rs, b_cs, SP ? B_ghost : nullptr, SP ? bg_rs : 0, SP ? n_own : 0, C, c_rs, \
        c_cs, nrows, k, index_base, block_list, (uint32_t)launch_blocks, accumulate)
        if (split) { if (k <= 4) HPCLA_SPMM_GEN(true, 4); else if (k <= 8) HPCLA_SPMM_GEN(true, 8); else HPCLA_SPMM_GEN(true, 16); }
        else { if (k <= 4) HPCLA_SPMM_GEN(false, 4); else if (k <= 8) HPCLA_SPMM_GEN(false, 8); else HPCLA_SPMM_GEN(false, 16); }
#undef HPCLA_SPMM_GEN
    }
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int hpcla_spmm_rows_per_block(void) { return RPB_MM; }

HPCLA_API int hpcla_spmm_block_order_hint(const void *rowptr, int group)
{
    if (!rowptr) return set_error(HPCLA_ERR_INVALID, "spmm_block_order_hint: null rowptr");
    if (group < 0 || group > 4096 || (group & (group - 1)) != 0)
        return set_error(HPCLA_ERR_INVALID, "spmm_block_order_hint: group must be 0 or a power of two <= 4096");
    int l = 0;
    for (int g = group; g > 1; g >>= 1) ++l;
    set_spmm_block_order(rowptr, l);
    return HPCLA_OK;
}

// The measuring loop of the plan-time tuners: `launch` (the launch the plan will make; every call writes the complete, correct
// product) under each candidate group (log2; 0 = natural order), interleaved, round 0 not counted, median of three; natural
// unless a grouped order is >= 1 % faster.  Leaves the chosen order set for `rowptr`.
template <typename Launch>
static int tune_order_measured(const void *rowptr, const int *cand, int nc, void *stream, int *chosen_group, const char *who,
                               Launch launch)
{
    constexpr int NC_MAX = 8, ROUNDS = 4, REPS = 2;
    if (nc > NC_MAX) nc = NC_MAX;
    float ms[NC_MAX][ROUNDS];
    hipEvent_t e0, e1;
    HPCLA_CHECK_HIP(hipEventCreate(&e0));
    HPCLA_CHECK_HIP(hipEventCreate(&e1));
    hipStream_t s = as_stream(stream);
    int rc = HPCLA_OK;
    for (int r = 0; r < ROUNDS && rc == HPCLA_OK; ++r)
        for (int c = 0; c < nc && rc == HPCLA_OK; ++c) {
            set_spmm_block_order(rowptr, cand[c]);
            if (hipEventRecord(e0, s) != hipSuccess) { rc = set_error(HPCLA_ERR_HIP, "%s: event", who); break; }
            for (int i = 0; i < REPS && rc == HPCLA_OK; ++i) rc = launch();
            if (rc != HPCLA_OK) break;
            if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                hipEventElapsedTime(&ms[c][r], e0, e1) != hipSuccess)
                rc = set_error(HPCLA_ERR_HIP, "%s: timing", who);
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    set_spmm_block_order(rowptr, 0);
    if (rc != HPCLA_OK) return rc;
    float med[NC_MAX];
    for (int c = 0; c < nc; ++c) {                           // median of the three counted rounds
        float a = ms[c][1], b = ms[c][2], d = ms[c][3];
        med[c] = a > b ? (b > d ? b : (a > d ? d : a)) : (a > d ? a : (b > d ? d : b));
    }
    int best = 0;
    for (int c = 1; c < nc; ++c)
        if (med[c] < med[best]) best = c;
    if (best != 0 && med[best] > 0.99f * med[0]) best = 0;
    set_spmm_block_order(rowptr, cand[best]);
    if (chosen_group) *chosen_group = 1 << cand[best];
    return HPCLA_OK;
}

// Plan-time choice of the SpMM block order BY MEASUREMENT, the SpMV tuner's twin (spmv.hip tune_block_order): the launch
// the plan will make -- same arguments, results into the caller's C (every launch writes the complete, correct product) --
// under the natural order and groups of 16 / 64 / 256 row blocks (of 64 rows), interleaved; natural unless a grouped order
// is >= 1 % faster.  What was measured by hand (profiles/r03_spmm_xcd_group.log): the 5-point matrix LOSES with every
// group from 32 up (the B rows of its +-64-block neighbours land in other L2s), config 5's random pattern GAINS 2.6 % at
// 64-256 -- per structure, hence measured.
template <typename I>
static int spmm_tune_block_order(const I *rowptr, const I *colval_split, const double *nzval, const double *B_own,
                                 int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C,
                                 int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base,
                                 const int32_t *block_list, int64_t n_blocks, void *stream, int *chosen_group)
{
    if (chosen_group) *chosen_group = 1;
    if (nrows < 0 || nnz < 0 || k < 0 || !rowptr) return set_error(HPCLA_ERR_INVALID, "spmm_tune_block_order: bad arguments");
    set_spmm_block_order(rowptr, 0);
    const int64_t launch_blocks = block_list ? n_blocks : (nrows + RPB_MM - 1) / RPB_MM;
    // small launches live in the caches; k = 1 is the SpMV; odd k on an odd pitch takes the generic-stride kernel, which keeps
    // the natural order (odd k on an even pitch > k: the vec kernel, round 6 -- measured like every even k)
    if (launch_blocks < 4096 || nnz == 0 || k < 2 || ((k & 1) && ((ldb_own & 1) || (ldc & 1)))) return HPCLA_OK;
    const bool split = B_ghost != nullptr;                   // no ghost segment: the unsplit instantiation, like hpcla_spmm_csr_*
    if (!C || !B_own) return set_error(HPCLA_ERR_INVALID, "spmm_tune_block_order: null B / C");
    const int cand[4] = {0, 4, 6, 8};
    return tune_order_measured(rowptr, cand, 4, stream, chosen_group, "spmm_tune_block_order", [&]() {
        return spmm_launch<I>(rowptr, colval_split, nzval, B_own, ldb_own, 1, B_ghost, ldb_ghost, n_own, split, C, ldc, 1, nrows, nnz, k,
                              index_base, block_list, n_blocks, stream);
    });
}

HPCLA_API int hpcla_spmm_tune_block_order_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                                  const double *B_own, int64_t ldb_own, const double *B_ghost,
                                                  int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc, int64_t nrows,
                                                  int64_t nnz, int k, int index_base, const int32_t *block_list,
                                                  int64_t n_blocks, void *stream, int *chosen_group)
{
    return spmm_tune_block_order<int32_t>(rowptr, colval_split, nzval, B_own, ldb_own, B_ghost, ldb_ghost, n_own, C, ldc,
                                          nrows, nnz, k, index_base, block_list, n_blocks, stream, chosen_group);
}

HPCLA_API int hpcla_spmm_tune_block_order_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                                  const double *B_own, int64_t ldb_own, const double *B_ghost,
                                                  int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc, int64_t nrows,
                                                  int64_t nnz, int k, int index_base, const int32_t *block_list,
                                                  int64_t n_blocks, void *stream, int *chosen_group)
{
    return spmm_tune_block_order<int64_t>(rowptr, colval_split, nzval, B_own, ldb_own, B_ghost, ldb_ghost, n_own, C, ldc,
                                          nrows, nnz, k, index_base, block_list, n_blocks, stream, chosen_group);
}

namespace hpcla {
// colmajor.hip: both operands column-major (Julia's Matrix) -> the lanes = rows kernel, no layout conversion
int spmm_colmajor_i32(const int32_t *rowptr, const int32_t *colval, const double *nzval, const double *B, int64_t ldb, double *C,
                      int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base, void *stream);
int spmm_colmajor_i64(const int64_t *rowptr, const int64_t *colval, const double *nzval, const double *B, int64_t ldb, double *C,
                      int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base, void *stream);
}  // namespace hpcla

HPCLA_API int hpcla_spmm_csr_f64_i32(const int32_t *rowptr, const int32_t *colval,
                                     const double *nzval, const double *B, int64_t ldb,
                                     int b_layout, double *C, int64_t ldc, int c_layout,
                                     int64_t nrows, int64_t nnz, int k, int index_base,
                                     void *stream)
{
    if (b_layout == HPCLA_LAYOUT_COL && c_layout == HPCLA_LAYOUT_COL && k > 1)
        return spmm_colmajor_i32(rowptr, colval, nzval, B, ldb, C, ldc, nrows, nnz, k, index_base, stream);
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
    if (b_layout == HPCLA_LAYOUT_COL && c_layout == HPCLA_LAYOUT_COL && k > 1)
        return spmm_colmajor_i64(rowptr, colval, nzval, B, ldb, C, ldc, nrows, nnz, k, index_base, stream);
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

// row-major B rows (own block + ghost segment), COLUMN-major C (round 5): the unstructured product of a column-major caller
// without the conversion of C (the vec kernel's CCOL store, one launch per 16-column tile; odd k: on an even B pitch > k;
// anything else: the generic strided kernel)
HPCLA_API int hpcla_spmm_split_ccol_f64_i32(const int32_t *rowptr, const int32_t *colval_split,
                                            const double *nzval, const double *B_own, int64_t ldb_own,
                                            const double *B_ghost, int64_t ldb_ghost, int64_t n_own,
                                            double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                            int index_base, const int32_t *block_list,
                                            int64_t n_blocks, void *stream)
{
    if (ldc < nrows) return set_error(HPCLA_ERR_INVALID, "spmm_split_ccol: ldc < nrows");
    return spmm_launch<int32_t>(rowptr, colval_split, nzval, B_own, ldb_own, 1, B_ghost, ldb_ghost,
                                n_own, true, C, 1, ldc, nrows, nnz, k, index_base, block_list,
                                n_blocks, stream);
}

HPCLA_API int hpcla_spmm_split_ccol_f64_i64(const int64_t *rowptr, const int64_t *colval_split,
                                            const double *nzval, const double *B_own, int64_t ldb_own,
                                            const double *B_ghost, int64_t ldb_ghost, int64_t n_own,
                                            double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                            int index_base, const int32_t *block_list,
                                            int64_t n_blocks, void *stream)
{
    if (ldc < nrows) return set_error(HPCLA_ERR_INVALID, "spmm_split_ccol: ldc < nrows");
    return spmm_launch<int64_t>(rowptr, colval_split, nzval, B_own, ldb_own, 1, B_ghost, ldb_ghost,
                                n_own, true, C, 1, ldc, nrows, nnz, k, index_base, block_list,
                                n_blocks, stream);
}

// ---- RUN TILES: plan-time descriptors + the k = 16 kernel (see spmm_rowblock_runs_kernel) ----------------------------
HPCLA_API int64_t hpcla_spmm_runs_desc_bytes(int64_t nrows)
{
    return nrows < 0 ? 0 : (int64_t)sizeof(SpmmRunDesc) * ((nrows + RPB_MM - 1) / RPB_MM);
}

template <typename I>
static int spmm_runs_build(const I *rowptr, const I *colval_split, int64_t nrows, int64_t nnz, int index_base, int64_t n_own,
                           void *desc, int64_t *n_fit_host, void *stream, int banded_max_runs = -1)
{
    const bool count_only = banded_max_runs >= 0;          // desc == nullptr: hpcla_spmm_banded_blocks_*
    if (n_fit_host) *n_fit_host = 0;
    if (nrows < 0 || nnz < 0) return set_error(HPCLA_ERR_INVALID, "spmm_runs_build: negative size");
    if (index_base != 0 && index_base != 1) return set_error(HPCLA_ERR_INVALID, "spmm_runs_build: index_base must be 0 or 1");
    if (nrows == 0) return HPCLA_OK;
    if (!rowptr || (!desc && !count_only) || (nnz > 0 && !colval_split)) return set_error(HPCLA_ERR_INVALID, "spmm_runs_build: null pointer");
    if (n_own > 0x7fffffffLL) return set_error(HPCLA_ERR_UNSUPPORTED, "spmm_runs_build: split columns beyond Int32");
    const int64_t nb = (nrows + RPB_MM - 1) / RPB_MM;
    HPCLA_CHECK_GRID(nb, "spmm_runs_build");
    unsigned long long *d_fit = nullptr;
    HPCLA_CHECK_HIP(hipMalloc((void **)&d_fit, sizeof(unsigned long long)));
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(d_fit, 0, sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        spmm_runs_build_kernel<I><<<(uint32_t)nb, TPB_MM, 0, s>>>(rowptr, colval_split, nrows, index_base, n_own,
                                                                  count_only ? nullptr : reinterpret_cast<SpmmRunDesc *>(desc), d_fit,
                                                                  banded_max_runs);
        e = hipGetLastError();
    }
    unsigned long long fit = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&fit, d_fit, sizeof(fit), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_fit);
    if (e != hipSuccess) return set_error(HPCLA_ERR_HIP, "spmm_runs_build: %s", hipGetErrorString(e));
    if (n_fit_host) *n_fit_host = (int64_t)fit;
    return HPCLA_OK;
}

HPCLA_API int hpcla_spmm_runs_build_i32(const int32_t *rowptr, const int32_t *colval_split, int64_t nrows, int64_t nnz,
                                        int index_base, int64_t n_own, void *desc, int64_t *n_fit_host, void *stream)
{
    return spmm_runs_build<int32_t>(rowptr, colval_split, nrows, nnz, index_base, n_own, desc, n_fit_host, stream);
}

HPCLA_API int hpcla_spmm_runs_build_i64(const int64_t *rowptr, const int64_t *colval_split, int64_t nrows, int64_t nnz,
                                        int index_base, int64_t n_own, void *desc, int64_t *n_fit_host, void *stream)
{
    return spmm_runs_build<int64_t>(rowptr, colval_split, nrows, nnz, index_base, n_own, desc, n_fit_host, stream);
}

// how many 64-row blocks touch at most `max_runs` contiguous runs of (split) columns: the BANDED test of a structure
HPCLA_API int hpcla_spmm_banded_blocks_i32(const int32_t *rowptr, const int32_t *colval_split, int64_t nrows, int64_t nnz,
                                           int index_base, int64_t n_own, int max_runs, int64_t *n_blocks_host, void *stream)
{
    if (max_runs < 1) return set_error(HPCLA_ERR_INVALID, "spmm_banded_blocks: max_runs must be positive");
    return spmm_runs_build<int32_t>(rowptr, colval_split, nrows, nnz, index_base, n_own, nullptr, n_blocks_host, stream, max_runs);
}
HPCLA_API int hpcla_spmm_banded_blocks_i64(const int64_t *rowptr, const int64_t *colval_split, int64_t nrows, int64_t nnz,
                                           int index_base, int64_t n_own, int max_runs, int64_t *n_blocks_host, void *stream)
{
    if (max_runs < 1) return set_error(HPCLA_ERR_INVALID, "spmm_banded_blocks: max_runs must be positive");
    return spmm_runs_build<int64_t>(rowptr, colval_split, nrows, nnz, index_base, n_own, nullptr, n_blocks_host, stream, max_runs);
}

template <typename I>
static int spmm_runs_launch(const I *rowptr, const I *colval_split, const double *nzval, const double *B_own,
                            const double *B_ghost, int64_t n_own, double *C, int64_t nrows, int64_t nnz, int index_base,
                            const void *desc, const int32_t *block_list, int64_t n_blocks, void *stream)
{
    if (nrows < 0 || nnz < 0) return set_error(HPCLA_ERR_INVALID, "spmm_runs: negative size");
    if (index_base != 0 && index_base != 1) return set_error(HPCLA_ERR_INVALID, "spmm_runs: index_base must be 0 or 1");
    if (nrows == 0) return HPCLA_OK;
    if (!rowptr || !C || !desc) return set_error(HPCLA_ERR_INVALID, "spmm_runs: null rowptr / C / descriptors");
    if (nnz > 0 && (!colval_split || !nzval || !B_own)) return set_error(HPCLA_ERR_INVALID, "spmm_runs: null colval / nzval / B with nnz > 0");
    if (((reinterpret_cast<uintptr_t>(B_own) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(B_ghost)) & 15) != 0)
        return set_error(HPCLA_ERR_INVALID, "spmm_runs: B / ghost / C must be 16-byte aligned");
    const int64_t all_blocks = (nrows + RPB_MM - 1) / RPB_MM;
    int64_t launch_blocks = all_blocks;
    if (block_list) {
        if (n_blocks < 0 || n_blocks > all_blocks) return set_error(HPCLA_ERR_INVALID, "spmm_runs: n_blocks out of range");
        launch_blocks = n_blocks;
    }
    if (launch_blocks == 0) return HPCLA_OK;
    HPCLA_CHECK_GRID(launch_blocks, "spmm_runs");
    hipStream_t s = as_stream(stream);
    const SpmmRunDesc *rd = reinterpret_cast<const SpmmRunDesc *>(desc);
    const int glog2 = spmm_group_log2(rowptr);
    if (B_ghost)
        spmm_rowblock_runs_kernel<I, true><<<(uint32_t)launch_blocks, TPB_MM, 0, s>>>(
            rowptr, colval_split, nzval, B_own, B_ghost, n_own, C, nrows, index_base, rd, block_list, (uint32_t)launch_blocks, glog2);
    else
        spmm_rowblock_runs_kernel<I, false><<<(uint32_t)launch_blocks, TPB_MM, 0, s>>>(
            rowptr, colval_split, nzval, B_own, nullptr, 0, C, nrows, index_base, rd, block_list, (uint32_t)launch_blocks, glog2);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_spmm_runs_k16_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                          const double *B_own, const double *B_ghost, int64_t n_own, double *C,
                                          int64_t nrows, int64_t nnz, int index_base, const void *desc,
                                          const int32_t *block_list, int64_t n_blocks, void *stream)
{
    return spmm_runs_launch<int32_t>(rowptr, colval_split, nzval, B_own, B_ghost, n_own, C, nrows, nnz, index_base, desc,
                                     block_list, n_blocks, stream);
}

HPCLA_API int hpcla_spmm_runs_k16_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                          const double *B_own, const double *B_ghost, int64_t n_own, double *C,
                                          int64_t nrows, int64_t nnz, int index_base, const void *desc,
                                          const int32_t *block_list, int64_t n_blocks, void *stream)
{
    return spmm_runs_launch<int64_t>(rowptr, colval_split, nzval, B_own, B_ghost, n_own, C, nrows, nnz, index_base, desc,
                                     block_list, n_blocks, stream);
}

// the run tiles on a column-major caller's blocks (spmm_runs_colmajor_kernel)
template <typename I>
static int spmm_runs_colmajor_launch(const I *rowptr, const I *colval_split, const double *nzval, const double *B_own,
                                     int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C,
                                     int64_t ldc, int64_t nrows, int64_t nnz, int index_base, const void *desc,
                                     const int32_t *block_list, int64_t n_blocks, void *stream)
{
    const char *who = "spmm_runs_colmajor";
    if (nrows < 0 || nnz < 0 || n_own < 0) return set_error(HPCLA_ERR_INVALID, "%s: negative size", who);
    if (index_base != 0 && index_base != 1) return set_error(HPCLA_ERR_INVALID, "%s: index_base must be 0 or 1", who);
    if (nrows == 0) return HPCLA_OK;
    if (!rowptr || !C || !desc) return set_error(HPCLA_ERR_INVALID, "%s: null rowptr / C / descriptors", who);
    if (nnz > 0 && (!colval_split || !nzval || !B_own)) return set_error(HPCLA_ERR_INVALID, "%s: null colval / nzval / B with nnz > 0", who);
    if (ldb_own < n_own || ldc < nrows) return set_error(HPCLA_ERR_INVALID, "%s: leading dimension smaller than the row count", who);
    if (B_ghost && ldb_ghost < KT) return set_error(HPCLA_ERR_INVALID, "%s: ldb_ghost < 16", who);
    if (n_own > 0x7fffffffLL) return set_error(HPCLA_ERR_UNSUPPORTED, "%s: split columns beyond Int32", who);
    if ((reinterpret_cast<uintptr_t>(B_own) & 15) != 0 || (ldb_own & 1) != 0)
        return set_error(HPCLA_ERR_UNSUPPORTED, "%s: B_own must be 16-byte aligned with an even leading dimension "
                                                "(hpcla_spmm_split_colmajor_f64_* takes any)", who);
    const int64_t all_blocks = (nrows + RPB_MM - 1) / RPB_MM;
    int64_t launch_blocks = all_blocks;
    if (block_list) {
        if (n_blocks < 0 || n_blocks > all_blocks) return set_error(HPCLA_ERR_INVALID, "%s: n_blocks out of range", who);
        launch_blocks = n_blocks;
    }
    if (launch_blocks == 0) return HPCLA_OK;
    HPCLA_CHECK_GRID(launch_blocks, who);
    hipStream_t s = as_stream(stream);
    const SpmmRunDesc *rd = reinterpret_cast<const SpmmRunDesc *>(desc);
    const int glog2 = spmm_group_log2(rowptr);
    if (B_ghost)
        spmm_runs_colmajor_kernel<I, true><<<(uint32_t)launch_blocks, TPB_MM, 0, s>>>(
            rowptr, colval_split, nzval, B_own, ldb_own, B_ghost, ldb_ghost, n_own, C, ldc, nrows, index_base, rd, block_list,
            (uint32_t)launch_blocks, glog2);
    else
        spmm_runs_colmajor_kernel<I, false><<<(uint32_t)launch_blocks, TPB_MM, 0, s>>>(
            rowptr, colval_split, nzval, B_own, ldb_own, nullptr, 0, n_own, C, ldc, nrows, index_base, rd, block_list,
            (uint32_t)launch_blocks, glog2);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_spmm_runs_colmajor_k16_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                                   const double *B_own, int64_t ldb_own, const double *B_ghost,
                                                   int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc, int64_t nrows,
                                                   int64_t nnz, int index_base, const void *desc, const int32_t *block_list,
                                                   int64_t n_blocks, void *stream)
{
    return spmm_runs_colmajor_launch<int32_t>(rowptr, colval_split, nzval, B_own, ldb_own, B_ghost, ldb_ghost, n_own, C, ldc,
                                              nrows, nnz, index_base, desc, block_list, n_blocks, stream);
}

HPCLA_API int hpcla_spmm_runs_colmajor_k16_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                                   const double *B_own, int64_t ldb_own, const double *B_ghost,
                                                   int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc, int64_t nrows,
                                                   int64_t nnz, int index_base, const void *desc, const int32_t *block_list,
                                                   int64_t n_blocks, void *stream)
{
    return spmm_runs_colmajor_launch<int64_t>(rowptr, colval_split, nzval, B_own, ldb_own, B_ghost, ldb_ghost, n_own, C, ldc,
                                              nrows, nnz, index_base, desc, block_list, n_blocks, stream);
}

// Plan-time block order of the column-major run tiles, BY MEASUREMENT (tune_order_measured): this kernel's best group depends
// on the grid's width like the others' -- 5-point matrix x 16, 4096 x 2048 rows: natural 0.60-0.64 ms, groups of 8 blocks
// 0.563; 2896 x 2896: natural 0.708, 8: 0.670, 128: 0.570 (profiles/r05_colmajor_run_tiles.log) -- and differs from the
// gather kernel's, which hpcla_spmm_tune_block_order_* measures.  Same arguments as the product; C receives the product.
template <typename I>
static int spmm_runs_colmajor_tune(const I *rowptr, const I *colval_split, const double *nzval, const double *B_own,
                                   int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc,
                                   int64_t nrows, int64_t nnz, int index_base, const void *desc, const int32_t *block_list,
                                   int64_t n_blocks, void *stream, int *chosen_group)
{
    if (chosen_group) *chosen_group = 1;
    if (!rowptr) return set_error(HPCLA_ERR_INVALID, "spmm_runs_colmajor_tune_block_order: null rowptr");
    set_spmm_block_order(rowptr, 0);
    auto launch = [&]() {
        return spmm_runs_colmajor_launch<I>(rowptr, colval_split, nzval, B_own, ldb_own, B_ghost, ldb_ghost, n_own, C, ldc, nrows, nnz,
                                            index_base, desc, block_list, n_blocks, stream);
    };
    const int64_t launch_blocks = block_list ? n_blocks : (nrows + RPB_MM - 1) / RPB_MM;
    if (launch_blocks < 4096 || nnz <= 0) return launch();    // small launches live in the caches: natural order, arguments checked
    const int cand[5] = {0, 3, 5, 7, 9};
    return tune_order_measured(rowptr, cand, 5, stream, chosen_group, "spmm_runs_colmajor_tune_block_order", launch);
}

HPCLA_API int hpcla_spmm_runs_colmajor_tune_block_order_f64_i32(const int32_t *rowptr, const int32_t *colval_split,
                                                                const double *nzval, const double *B_own, int64_t ldb_own,
                                                                const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C,
                                                                int64_t ldc, int64_t nrows, int64_t nnz, int index_base,
                                                                const void *desc, const int32_t *block_list, int64_t n_blocks,
                                                                void *stream, int *chosen_group)
{
    return spmm_runs_colmajor_tune<int32_t>(rowptr, colval_split, nzval, B_own, ldb_own, B_ghost, ldb_ghost, n_own, C, ldc, nrows, nnz,
                                            index_base, desc, block_list, n_blocks, stream, chosen_group);
}

HPCLA_API int hpcla_spmm_runs_colmajor_tune_block_order_f64_i64(const int64_t *rowptr, const int64_t *colval_split,
                                                                const double *nzval, const double *B_own, int64_t ldb_own,
                                                                const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C,
                                                                int64_t ldc, int64_t nrows, int64_t nnz, int index_base,
                                                                const void *desc, const int32_t *block_list, int64_t n_blocks,
                                                                void *stream, int *chosen_group)
{
    return spmm_runs_colmajor_tune<int64_t>(rowptr, colval_split, nzval, B_own, ldb_own, B_ghost, ldb_ghost, n_own, C, ldc, nrows, nnz,
                                            index_base, desc, block_list, n_blocks, stream, chosen_group);
}

// One PANEL of a product in panel order (the opt-in order of the distributed SpMM): a CSR
// matrix that holds the entries of A whose columns lie in one column panel; accumulate = 1 continues every
// C(r, c) from its current value, entry by entry in stored order -- so a product taken panel by panel is the
// reference's sum (src/sparse.jl:2391-2413) in a different ORDER (own columns, then chunk by chunk), never a sum of
// separately rounded partial sums.
HPCLA_API int hpcla_spmm_panel_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                       const double *B_own, int64_t ldb_own, const double *B_ghost,
                                       int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc, int64_t nrows,
                                       int64_t nnz, int k, int index_base, int accumulate, void *stream)
{
    return spmm_launch<int32_t>(rowptr, colval_split, nzval, B_own, ldb_own, 1, B_ghost, ldb_ghost, n_own, true, C,
                                ldc, 1, nrows, nnz, k, index_base, nullptr, 0, stream, accumulate ? 1 : 0);
}

HPCLA_API int hpcla_spmm_panel_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                       const double *B_own, int64_t ldb_own, const double *B_ghost,
                                       int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc, int64_t nrows,
                                       int64_t nnz, int k, int index_base, int accumulate, void *stream)
{
    return spmm_launch<int64_t>(rowptr, colval_split, nzval, B_own, ldb_own, 1, B_ghost, ldb_ghost, n_own, true, C,
                                ldc, 1, nrows, nnz, k, index_base, nullptr, 0, stream, accumulate ? 1 : 0);
}

// NARROW column-major block (cols <= 32: the k columns of an SpMM operand) -> PACKED row-major rows (leading dimension ==
// cols): a workgroup moves 256 rows.  The column-major side is read column by column (256 consecutive rows of one column:
// 2 KiB runs at Float64), the row-major side written linearly (its 256 x cols block is ONE contiguous region); the LDS tile
// has an odd pitch, so neither phase has bank conflicts.  The generic 32 x 32 tile kernel above uses half of its lanes on
// such a block: 0.43 ms on 8.4 M rows x 16, this one 0.36 ms (5.9 TB/s).  The OPPOSITE direction was measured too and stays
// on the generic kernel: 0.72-0.76 ms in this form (sixteen 2-KiB write streams per workgroup) against 0.57 ms
// (profiles/r04_colmajor.log).
constexpr int RELAY_ROWS = 256, RELAY_MAXC = 32;
template <typename T>
__global__ __launch_bounds__(RELAY_ROWS) void relayout_narrow_to_rows_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                                              int64_t ld_col, int64_t rows, int cols)
{
    // tile sized by the launch from the ACTUAL pitch (a static 256 x 33 tile is 67.6 KB whatever `cols` is -- two
    // workgroups per CU at k = 16 where 34.8 KB allow four, and more than a 64 KiB-LDS part can allocate at all)
    extern __shared__ __attribute__((aligned(16))) unsigned char relay_tile_raw[];
    T *tile = reinterpret_cast<T *>(relay_tile_raw);
    const int pitch = cols | 1;                           // odd
    const int64_t r0 = (int64_t)blockIdx.x * RELAY_ROWS;
    const int nr = (int)((rows - r0) < RELAY_ROWS ? (rows - r0) : RELAY_ROWS);
    const int tid = threadIdx.x;
    if (tid < nr)
        for (int c = 0; c < cols; ++c) tile[tid * pitch + c] = src[r0 + tid + (int64_t)c * ld_col];
    __syncthreads();
    const int count = nr * cols;
    T *d = dst + r0 * cols;
    for (int i = tid; i < count; i += RELAY_ROWS) { const int r = i / cols; d[i] = tile[r * pitch + (i - r * cols)]; }
}

template <typename T>
static int transpose_impl(const T *src, int64_t ld_src, int src_layout, T *dst, int64_t ld_dst, int dst_layout,
                          int64_t rows, int64_t cols, void *stream)
{
    if (rows < 0 || cols < 0) return set_error(HPCLA_ERR_INVALID, "transpose: negative size");
    if (rows == 0 || cols == 0) return HPCLA_OK;
    if (!src || !dst) return set_error(HPCLA_ERR_INVALID, "transpose: null pointer");
    if ((src_layout != HPCLA_LAYOUT_ROW && src_layout != HPCLA_LAYOUT_COL) ||
        (dst_layout != HPCLA_LAYOUT_ROW && dst_layout != HPCLA_LAYOUT_COL))
        return set_error(HPCLA_ERR_INVALID, "transpose: layout must be HPCLA_LAYOUT_ROW or HPCLA_LAYOUT_COL");
    // narrow column-major block -> packed row-major rows: the 256-row kernel
    if (cols <= RELAY_MAXC && rows >= RELAY_ROWS && src_layout == HPCLA_LAYOUT_COL && dst_layout == HPCLA_LAYOUT_ROW &&
        ld_dst == cols && ld_src >= rows) {
        const int64_t nb = (rows + RELAY_ROWS - 1) / RELAY_ROWS;
        if (nb <= 0x7fffffffLL) {
            const size_t lds = (size_t)RELAY_ROWS * (size_t)((int)cols | 1) * sizeof(T);
            static_assert((size_t)RELAY_ROWS * (RELAY_MAXC + 1) * sizeof(double) <= 160 * 1024, "tile exceeds gfx950's LDS");
            // more than 64 KiB of dynamic LDS (31-32 Float64 columns) has to be allowed once per kernel
            static const hipError_t allowed = hipFuncSetAttribute(
                reinterpret_cast<const void *>(&relayout_narrow_to_rows_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                (int)((size_t)RELAY_ROWS * (RELAY_MAXC + 1) * sizeof(T)));
            if (allowed == hipSuccess || lds <= 64 * 1024) {
            relayout_narrow_to_rows_kernel<T><<<(uint32_t)nb, RELAY_ROWS, lds, as_stream(stream)>>>(src, dst, ld_src, rows, (int)cols);
            HPCLA_CHECK_LAUNCH();
            return HPCLA_OK;
            }
        }
    }
    int64_t srs, scs, drs, dcs;
    layout_strides(src_layout, ld_src, &srs, &scs);
    layout_strides(dst_layout, ld_dst, &drs, &dcs);
    const int64_t gy = (cols + 31) / 32;
    if (gy > 65535) return set_error(HPCLA_ERR_UNSUPPORTED, "transpose: more than %d columns", 65535 * 32);
    dim3 grid((uint32_t)((rows + 31) / 32), (uint32_t)gy);
    relayout_kernel<T><<<grid, 256, 0, as_stream(stream)>>>(src, srs, scs, dst, drs, dcs, rows, cols);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

HPCLA_API int hpcla_transpose_f64(const double *src, int64_t ld_src, int src_layout, double *dst,
                                  int64_t ld_dst, int dst_layout, int64_t rows, int64_t cols,
                                  void *stream)
{
    return transpose_impl<double>(src, ld_src, src_layout, dst, ld_dst, dst_layout, rows, cols, stream);
}

HPCLA_API int hpcla_transpose_f32(const float *src, int64_t ld_src, int src_layout, float *dst, int64_t ld_dst,
                                  int dst_layout, int64_t rows, int64_t cols, void *stream)
{
    return transpose_impl<float>(src, ld_src, src_layout, dst, ld_dst, dst_layout, rows, cols, stream);
}
