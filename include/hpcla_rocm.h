/*
 * hpcla_rocm.h -- C ABI of libhpcla_rocm.so, the MI355X (gfx950) device library behind the
 * `DeviceROCm` backend of HPCLinearAlgebra.jl (sloisel/LinearAlgebraMPI.jl).
 *
 * Boundary.  The reference has no FFI for this path: its device work is Julia
 * KernelAbstractions kernels plus CPU-staged MPI.  The entry points below are what a
 * `ext/HPCLinearAlgebraROCmExt.jl` binds with `@ccall`, one per reference
 * function it replaces; each declaration cites that function (paths relative to the reference
 * repository root).  The pattern -- `ccall` returning an int status that the Julia wrapper checks --
 * is the one the reference already uses for cuDSS/NCCL (ext/HPCLinearAlgebraCUDAExt.jl:247-251,
 * 388-402).
 *
 * Conventions
 *  - plain pointers and sizes only; every data buffer is a DEVICE pointer owned by the caller
 *    (ROCArray / torch tensor / hipMalloc) unless the parameter name ends in `_host`;
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); all work is
 *    enqueued asynchronously, nothing synchronises the host unless documented;
 *  - return value: HPCLA_OK (0) or a negative status; hpcla_last_error() gives the text;
 *  - index arrays are int32 (`_i32`) or int64 (`_i64`, the reference default Ti=Int,
 *    src/backends.jl:348); `index_base` is 1 for arrays passed untouched from Julia
 *    (rowptr[1]==1, src/sparse.jl:2060) and 0 for C/Python callers;
 *  - handles (comm, halo plan) are library-owned, explicitly destroyed, never finalised with a
 *    collective (the reference's rule: ext/HPCLinearAlgebraCUDAExt.jl:9-10, 384-386);
 *  - not thread-safe per handle; re-entrant across handles.
 *
 * Arithmetic contract (fp64): every row sum is accumulated sequentially in stored order with a
 * separately rounded multiply and add (no FMA contraction), i.e. bit-identical to the reference
 * loop `acc += nzval[j] * x[colval[j]]` (src/sparse.jl:2055-2066) on IEEE hardware.
 */
#ifndef HPCLA_ROCM_H
#define HPCLA_ROCM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPCLA_VERSION 100 /* 0.1.0 */

/* status codes */
#define HPCLA_OK 0
#define HPCLA_ERR_INVALID -1     /* bad argument (null pointer, negative size, bad base/layout) */
#define HPCLA_ERR_HIP -2         /* a HIP runtime call failed */
#define HPCLA_ERR_RCCL -3        /* an RCCL call failed or librccl could not be loaded */
#define HPCLA_ERR_NO_DEVICE -4   /* no gfx950 device visible */
#define HPCLA_ERR_UNSUPPORTED -5 /* valid request this build does not implement */
#define HPCLA_ERR_ALLOC -6       /* device or host allocation failed */

/* dense layouts for SpMM operands: element (i,c) at ptr[i*ld + c] (ROW) or ptr[i + c*ld] (COL).
 * COL is Julia's Matrix (src/dense.jl:63); ROW is the device-native layout (one 128-byte line per
 * row at k=16). */
#define HPCLA_LAYOUT_ROW 0
#define HPCLA_LAYOUT_COL 1

typedef struct hpcla_comm hpcla_comm_t;           /* one RCCL communicator (or a serial stub) */
typedef struct hpcla_halo_plan hpcla_halo_plan_t; /* device half of a VectorPlan */

/* ---- library ------------------------------------------------------------------------------ */
int hpcla_version(void);
const char *hpcla_last_error(void);
/* Number of visible devices / select device for the calling thread (hipSetDevice).
 * Mirrors the rank->device assignment of ext/HPCLinearAlgebraCUDAExt.jl:611-613. */
int hpcla_device_count(int *count);
int hpcla_set_device(int device);
/* Device properties the host layer and bench report (CU count, arch name up to 63 chars). */
int hpcla_device_info(int device, int *num_cus, char *arch_name_host, int arch_name_len);

/* ---- SpMV:  replaces _spmv_kernel!/_spmv! (src/sparse.jl:2055-2084) ----------------------------
 * y[r] = sum_{j=rowptr[r]}^{rowptr[r+1]-1} nzval[j] * x[colval[j]],  r in [0,nrows).
 * `x` is the gathered vector x[col_indices] (length ncols_compressed), exactly the operand the
 * reference passes (`plan.gathered`, src/sparse.jl:2114-2119).  nnz = rowptr[nrows]-index_base is
 * passed by the caller (Julia: length(A.nzval)) so the launch needs no device->host read. */
int hpcla_spmv_csr_f64_i32(const int32_t *rowptr, const int32_t *colval, const double *nzval,
                           const double *x, double *y, int64_t nrows, int64_t nnz, int index_base,
                           void *stream);
int hpcla_spmv_csr_f64_i64(const int64_t *rowptr, const int64_t *colval, const double *nzval,
                           const double *x, double *y, int64_t nrows, int64_t nnz, int index_base,
                           void *stream);

/* Split-column SpMV used by the distributed path: a column index c < n_own reads x_own[c] (the
 * caller's HPCVector storage, x.v, untouched), c >= n_own reads x_ghost[c - n_own] (the halo
 * plan's receive segment).  This removes the reference's local gather loop
 * (src/vectors.jl:426-428, _gather_kernel! :174-177) from the hot path altogether.
 * `block_list` (device, int32, may be NULL) restricts the launch to the listed row blocks of
 * hpcla_spmv_rows_per_block() rows each -- the interior/boundary split.  colval is 0-based. */
int hpcla_spmv_split_f64_i32(const int32_t *rowptr, const int32_t *colval_split,
                             const double *nzval, const double *x_own, const double *x_ghost,
                             int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base,
                             const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmv_split_f64_i64(const int64_t *rowptr, const int64_t *colval_split,
                             const double *nzval, const double *x_own, const double *x_ghost,
                             int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base,
                             const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmv_rows_per_block(void);
/* The CSR SpMV kernel of every aligned launch (every entry point above and below; no reference counterpart -- the
 * reference has one work-item per row, src/sparse.jl:2055-2066) is the "row gather" (round 4): the block's A entries are
 * streamed into wave-private LDS unmultiplied and every lane walks its own row, so one gather instruction reads ONE x
 * stream for banded matrices (profiles/r04_spmv_2d_vs_3d_counters.txt).  The product-parking quad kernel of rounds 1-3 and
 * its switch (hpcla_set_spmv_kernel / hpcla_get_spmv_kernel, HPCLA_SPMV_KERNEL) were retired in round 6: it lost on every
 * measured matrix (profiles/r04_spmv_rowg.log).  Unaligned colval / nzval take a narrow fallback kernel, same bits. */
/* OPT-IN long rows (round 5; NOT the default, NOT bit-identical to the reference).  The default kernels sum every row
 * sequentially in stored order on one lane -- the reference's bits (acc += nzval[j] * x[colval[j]], src/sparse.jl:2059-2064)
 * and the reference's cliff: its _spmv_kernel! is one work-item per row too, so an "arrow" matrix (one dense row of n
 * entries) costs a whole sequential pass over that row.  This entry computes y = A*x with the rows listed in `long_rows`
 * (device array of 0-based row indices, exactly the rows that hold >= long_min >= 928 entries; at most 65535) summed in TREE
 * order -- 1024 contiguous pieces per row, wave shuffles inside a piece (north_star's "__shfl / segmented-scan row
 * reductions") -- and every other row exactly as hpcla_spmv_split_f64_* does, bit for bit.  A long row's result is within
 * 1e-12 * (|A||x|)_r of the sequential sum (tests: tests/test_gpu_parity.py::test_spmv_long_rows_opt_in).  x_ghost == NULL:
 * unsplit column space.  work: hpcla_spmv_longrows_work_bytes(n_long) bytes of device scratch.
 * A row of >= long_min entries that `long_rows` omits is not summed by anybody: its y is set to NaN (never left at a stale
 * value).  colval / nzval that are not 16- / 32-byte aligned take the default entry's fallback kernel (every row sequential). */
int64_t hpcla_spmv_longrows_work_bytes(int64_t n_long);
int hpcla_spmv_longrows_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                const double *x_own, const double *x_ghost, int64_t n_own, double *y,
                                int64_t nrows, int64_t nnz, int index_base, const int64_t *long_rows,
                                int64_t n_long, int64_t long_min, double *work, void *stream);
int hpcla_spmv_longrows_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                const double *x_own, const double *x_ghost, int64_t n_own, double *y,
                                int64_t nrows, int64_t nnz, int index_base, const int64_t *long_rows,
                                int64_t n_long, int64_t long_min, double *work, void *stream);
int hpcla_spmm_rows_per_block(void);
/* SpMM twins of the two block-order entries below (no reference counterpart: src/sparse.jl:2391-2413 is a column loop
 * over A*x in index order): the row blocks (hpcla_spmm_rows_per_block() rows each) of every SpMM launch over `rowptr`
 * -- a contiguous launch, or the POSITIONS of a block list -- are walked in XCD groups of `group` blocks.  Measured per
 * structure at plan time: the 5-point matrix loses with every group from 32 up, config 5's random pattern gains 2.6 %
 * at 64-256 (profiles/r03_spmm_xcd_group.log); the tuner times the plan's own launch (results go to the caller's C:
 * every launch writes the complete product) under natural / 16 / 64 / 256, keeps natural unless a group is >= 1 %
 * faster, and skips launches below 4096 row blocks, k < 2 and odd k on odd pitches.  B_ghost == NULL: the unsplit kernel. */
int hpcla_spmm_block_order_hint(const void *rowptr, int group);
int hpcla_spmm_tune_block_order_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                        const double *B_own, int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost,
                                        int64_t n_own, double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                        int index_base, const int32_t *block_list, int64_t n_blocks, void *stream,
                                        int *chosen_group);
int hpcla_spmm_tune_block_order_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                        const double *B_own, int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost,
                                        int64_t n_own, double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                        int index_base, const int32_t *block_list, int64_t n_blocks, void *stream,
                                        int *chosen_group);

/* Optional plan-time performance hint (no reference counterpart: the reference launches one work-item per row in
 * index order, src/sparse.jl:2055-2066, 2081-2082): SpMV launches over the matrix whose `rowptr` device pointer is
 * given walk their contiguous runs of row blocks in an XCD-grouped order -- groups of `group` consecutive row blocks
 * per XCD, groups dealt round-robin -- so that row blocks whose columns overlap share one of the 8 L2s (config 4's
 * 512 x 512 planes: neighbours two blocks apart; -2 to -5 % per SpMV on every stencil matrix measured).  `group` = 0 or 1 removes the hint (natural order);
 * otherwise a power of two <= 1024.  Results are bit-identical in every order (each row is still one sequential sum);
 * launches restricted to a block LIST keep the list's order.  Keyed by the pointer: remove the hint before the
 * array is freed (a stale entry can only cost performance).  Thread-safe. */
int hpcla_spmv_block_order_hint(const void *rowptr, int group);

/* Plan-time choice of that order BY MEASUREMENT (run once per structure, next to the VectorPlan build of
 * src/sparse.jl:1875-1984, never inside a timed or captured region: it synchronises): times the split-column SpMV of
 * this matrix -- the arguments of hpcla_spmv_split_f64_*, results discarded into `y_scratch` (nrows doubles) -- under the
 * natural order and groups of 8 / 32 / 64 row blocks, interleaved over four rounds of four launches each, keeps the
 * fastest registered for `rowptr` (the natural order unless a grouped one is >= 1 % faster) and returns it in
 * *chosen_group (1 = natural).  Matrices below 4096 row blocks keep the natural order unmeasured. */
int hpcla_spmv_tune_block_order_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                        const double *x_own, const double *x_ghost, int64_t n_own, double *y_scratch,
                                        int64_t nrows, int64_t nnz, int index_base, void *stream, int *chosen_group);
int hpcla_spmv_tune_block_order_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                        const double *x_own, const double *x_ghost, int64_t n_own, double *y_scratch,
                                        int64_t nrows, int64_t nnz, int index_base, void *stream, int *chosen_group);

/* Plan-time helpers (run once per (A, x.partition), cached with the VectorPlan like
 * _vector_plan_cache, src/HPCLinearAlgebra.jl:133).
 * remap: out[j] = map[in[j] - index_base]   (compressed column -> split column; out is 0-based)
 * classify: flags[b] = 1 if row block b (rows_per_block rows: hpcla_spmv_rows_per_block() or
 *           hpcla_spmm_rows_per_block()) references any column >= n_own, else 0. */
int hpcla_remap_i32(const int32_t *in, const int32_t *map, int32_t *out, int64_t n, int index_base,
                    void *stream);
int hpcla_remap_i64(const int64_t *in, const int64_t *map, int64_t *out, int64_t n, int index_base,
                    void *stream);
/* Plan-time NARROWING of an Int64 structure.  The reference's default index type is Ti = Int (src/backends.jl:348,
 * 369: backend_cpu_serial / backend_cpu_mpi default to Int), so a caller who follows the defaults hands over Int64
 * rowptr / colval although every BASELINE configuration fits Int32 per GPU.  Indices are never results: when
 * nnz + index_base and n_own + n_ghost + index_base fit Int32, the plan keeps Int32 copies (remap_i64_to_i32: the
 * split colval straight from the Int64 compressed colval through an Int32 map; narrow_i64_to_i32: rowptr) and every
 * launch over that plan takes the _i32 kernels -- 12 instead of 16 index+value bytes per stored entry, same bits.
 * narrow: if `overflow_dev` is not NULL, *overflow_dev |= 1 when a value does not fit (the caller zeroes it). */
int hpcla_remap_i64_to_i32(const int64_t *in, const int32_t *map, int32_t *out, int64_t n, int index_base,
                           void *stream);
int hpcla_narrow_i64_to_i32(const int64_t *in, int32_t *out, int64_t n, uint32_t *overflow_dev, void *stream);
int hpcla_classify_blocks_i32(const int32_t *rowptr, const int32_t *colval_split, int64_t nrows,
                              int index_base, int64_t n_own, int rows_per_block, int32_t *flags,
                              void *stream);
int hpcla_classify_blocks_i64(const int64_t *rowptr, const int64_t *colval_split, int64_t nrows,
                              int index_base, int64_t n_own, int rows_per_block, int32_t *flags,
                              void *stream);

/* ---- device-side construction: replaces the host sort + per-nonzero binary search of
 * HPCSparseMatrix_local (col_indices = unique!(sort(rowval)), compress_AT; src/sparse.jl:501-509,
 * 137-144) by a presence bitmap + exclusive scan + emit pass over a column window.
 * colidx_global: this rank's nonzeros' GLOBAL 0-based columns (device int64); every id must lie in
 * [col_lo, col_lo+window).  Outputs: colval_out (local index + index_base per nonzero),
 * col_indices_out (device int64, capacity `window`; first *ncols_compressed_host entries valid,
 * ascending).  `work`: hpcla_colspace_work_bytes(window) device bytes.  Synchronises the stream. */
int64_t hpcla_colspace_work_bytes(int64_t window);
/* 256-bit order-sensitive digest of a device index array (the local pass of compute_structural_hash,
 * src/sparse.jl:97-121, without a D2H copy of colval): out_host[k] = sum_i mix(a[i]*P1 + (i+1)*P2 + S_k)
 * mod 2^64, k = 0..3.  Memoization keys only -- compared for equality, like the reference's Blake3
 * digests.  Synchronises the stream. */
int hpcla_digest_i32(const int32_t *a, int64_t n, uint64_t *out_host /* 4 words */, void *stream);
int hpcla_digest_i64(const int64_t *a, int64_t n, uint64_t *out_host /* 4 words */, void *stream);
int hpcla_compress_columns_i32(const int64_t *colidx_global, int64_t nnz, int64_t col_lo, int64_t window,
                               int32_t *colval_out, int index_base, int64_t *col_indices_out,
                               int64_t *ncols_compressed_host, void *work, void *stream);
int hpcla_compress_columns_i64(const int64_t *colidx_global, int64_t nnz, int64_t col_lo, int64_t window,
                               int64_t *colval_out, int index_base, int64_t *col_indices_out,
                               int64_t *ncols_compressed_host, void *work, void *stream);
/* On-device generator of the BASELINE stencil workload (create_2d_laplacian,
 * test/test_factorization.jl:60-102): rows [row_start,row_end) of the nx*ny 5-point Laplacian as
 * CSR with GLOBAL 0-based columns (rowptr_out: row_end-row_start+1 int64, 0-based). */
int64_t hpcla_poisson2d_nnz(int64_t nx, int64_t ny, int64_t row_start, int64_t row_end);
int hpcla_gen_poisson2d(int64_t nx, int64_t ny, int64_t row_start, int64_t row_end, int64_t *rowptr_out,
                        int64_t *colidx_out, double *vals_out, void *stream);
/* the 7-point analogue of BASELINE configs[3] (idx = (k*ny + j)*nx + i, diagonal 6) */
int64_t hpcla_poisson3d_nnz(int64_t nx, int64_t ny, int64_t nz, int64_t row_start, int64_t row_end);
int hpcla_gen_poisson3d(int64_t nx, int64_t ny, int64_t nz, int64_t row_start, int64_t row_end,
                        int64_t *rowptr_out, int64_t *colidx_out, double *vals_out, void *stream);

/* ---- SpGEMM local product (sparse x sparse): replaces the CPU SparseArrays multiply inside
 * Base.:*(A::HPCSparseMatrix, B::HPCSparseMatrix) (`CT = plan.AT * A_csc`, src/sparse.jl:991-1059).
 * G = the rows of B named by A.col_indices, gathered by the MatrixPlan (src/sparse.jl:554-978), as
 * CSR with int64 rowptr and GLOBAL int64 columns; A's colval indexes G's rows.  Each C(i,j) is the sum
 * over k ascending of separately rounded G(k,j)*A(i,k), first product assigned -- the reference order.
 *  ub:       ub_out[i] = sum of the lengths of the G rows that row i references
 *  numeric:  rows `row_list` (all with ub <= hpcla_spgemm_bin_cap(bin); bins 0,1,... until the cap is
 *            -1; G's column ids must be < 2^58) -> sorted (col,val) runs at c_*_tmp[ub_prefix[i] ...]
 *            and their lengths cnt[i]
 *  compact:  c_rowptr (exclusive scan of cnt) -> final CSR arrays */
int64_t hpcla_spgemm_bin_cap(int bin);
int hpcla_spgemm_ub_i32(const int32_t *a_rowptr, const int32_t *a_col, int64_t nrows, int index_base,
                        const int64_t *g_rowptr, int64_t *ub_out, void *stream);
int hpcla_spgemm_ub_i64(const int64_t *a_rowptr, const int64_t *a_col, int64_t nrows, int index_base,
                        const int64_t *g_rowptr, int64_t *ub_out, void *stream);
int hpcla_spgemm_numeric_i32(int bin, const int32_t *a_rowptr, const int32_t *a_col, const double *a_val,
                             int index_base, const int64_t *g_rowptr, const int64_t *g_col,
                             const double *g_val, const int32_t *row_list, int64_t n_list,
                             const int64_t *ub_prefix, int64_t *c_col_tmp, double *c_val_tmp,
                             int64_t *cnt, void *stream);
int hpcla_spgemm_numeric_i64(int bin, const int64_t *a_rowptr, const int64_t *a_col, const double *a_val,
                             int index_base, const int64_t *g_rowptr, const int64_t *g_col,
                             const double *g_val, const int32_t *row_list, int64_t n_list,
                             const int64_t *ub_prefix, int64_t *c_col_tmp, double *c_val_tmp,
                             int64_t *cnt, void *stream);
int hpcla_spgemm_compact(const int64_t *c_rowptr, const int64_t *ub_prefix, int64_t nrows,
                         const int64_t *c_col_tmp, const double *c_val_tmp, int64_t *c_col, double *c_val,
                         void *stream);
/* Repeated product on a cached structure (the reference recomputes `plan.AT * A_csc` with SparseArrays every time,
 * src/sparse.jl:1011; its MatrixPlan caches the gathering only): result entry e is the sum, in list order, of
 * g_val[pairs[2t + 1]] * a_val[pairs[2t]] for t in [pair_ptr[e], pair_ptr[e + 1]) -- the first product, then the others
 * added one by one (ascending k when the lists are built that way: the bits of hpcla_spgemm_numeric_*).  `pair_ptr` has
 * nnz_c + 1 entries, int32 or int64 (`ptr_is_i64`); `pairs` is 8-byte aligned.  One streaming pass: 8 B per product. */
int hpcla_spgemm_numeric_mapped_f64(const void *pair_ptr, int ptr_is_i64, const int32_t *pairs,
                                    const double *a_val, const double *g_val, double *c_val, int64_t nnz_c,
                                    void *stream);

/* ---- SpMM:  replaces A*B column loop (src/sparse.jl:2391-2413) -------------------------------
 * C[r,c] = sum_j nzval[j] * B[colval[j], c], c in [0,k): one pass over A for all k columns, each
 * (r,c) accumulated sequentially in stored order (bit-identical to k reference SpMVs).
 * B has ncols_compressed rows (or, split form, n_own rows in B_own and ghosts in B_ghost).
 * Fast (16-byte vector) path: row-major B with an EVEN pitch, C row-major on an even pitch or column-major, 16-byte
 * aligned bases.  ODD k takes it too when every row-major pitch is even AND larger than k (ldb = ldc = k + 1 is what the
 * converters of this boundary allocate): the kernel then reads the padding double B[i * ldb + k] of every row it gathers
 * (never stores it), so B / B_ghost must span rows * ldb doubles -- a last row cut off behind its k-th column is not
 * allowed there.  C's padding: with ldc == k + 1 <= 16 the block's C rows leave as whole lines and C[i * ldc + k] is set
 * to 0.0 (a masked last sector per row cost 40 % of the product); any other ldc leaves the padding untouched.  Any other stride combination runs on the generic one-column-per-lane kernel (5-point matrix x 15
 * columns: 0.946 ms on it against 0.475 for 16, profiles/r05_lds_footprint.log). */
int hpcla_spmm_csr_f64_i32(const int32_t *rowptr, const int32_t *colval, const double *nzval,
                           const double *B, int64_t ldb, int b_layout, double *C, int64_t ldc,
                           int c_layout, int64_t nrows, int64_t nnz, int k, int index_base,
                           void *stream);
int hpcla_spmm_csr_f64_i64(const int64_t *rowptr, const int64_t *colval, const double *nzval,
                           const double *B, int64_t ldb, int b_layout, double *C, int64_t ldc,
                           int c_layout, int64_t nrows, int64_t nnz, int k, int index_base,
                           void *stream);
int hpcla_spmm_split_f64_i32(const int32_t *rowptr, const int32_t *colval_split,
                             const double *nzval, const double *B_own, int64_t ldb_own,
                             const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C,
                             int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base,
                             const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmm_split_f64_i64(const int64_t *rowptr, const int64_t *colval_split,
                             const double *nzval, const double *B_own, int64_t ldb_own,
                             const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C,
                             int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base,
                             const int32_t *block_list, int64_t n_blocks, void *stream);
/* hpcla_spmm_split_f64_* with a COLUMN-major result (element (r, c) at C[r + c * ldc], ldc >= nrows; the reference's dense
 * block is a Julia Matrix, src/dense.jl:63) while B_own / B_ghost stay row-major rows: the product an UNSTRUCTURED matrix
 * gets from a column-major caller -- B converted once (hpcla_transpose_f64; an unstructured matrix gathers whole B rows, so
 * the row-major layout is the fast one), C written in the caller's layout by the product itself (the block's results leave
 * through LDS as runs of 64 doubles per column, one launch per 16-column tile; odd k: B pitches even and > k, see above;
 * other pitches: the strided kernel).  Same sums, same order, same bits.
 * hpcla_spmm_csr_f64_* with b_layout = HPCLA_LAYOUT_ROW, c_layout = HPCLA_LAYOUT_COL takes the same kernel. */
int hpcla_spmm_split_ccol_f64_i32(const int32_t *rowptr, const int32_t *colval_split,
                                  const double *nzval, const double *B_own, int64_t ldb_own,
                                  const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C,
                                  int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base,
                                  const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmm_split_ccol_f64_i64(const int64_t *rowptr, const int64_t *colval_split,
                                  const double *nzval, const double *B_own, int64_t ldb_own,
                                  const double *B_ghost, int64_t ldb_ghost, int64_t n_own, double *C,
                                  int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base,
                                  const int32_t *block_list, int64_t n_blocks, void *stream);
/* RUN TILES (round 4; no reference counterpart -- the reference's A * B is a column loop over A * x,
 * src/sparse.jl:2391-2413): for banded / stencil matrices the 64 rows of an SpMM row block touch a few CONTIGUOUS runs of
 * B rows (5-point matrix: 194 rows in 3 runs, where its entries name 320).  Plan time, once per structure:
 * hpcla_spmm_runs_build_* writes one 32-byte descriptor per block of hpcla_spmm_rows_per_block() rows into `desc`
 * (hpcla_spmm_runs_desc_bytes(nrows) device bytes) -- up to 4 runs {start, length} in the SPLIT column space, cut at the
 * own / ghost boundary; blocks with more runs, more than 200 distinct B rows or more than 512 entries are marked as
 * not fitting -- and returns the number of fitting blocks in *n_fit_host (synchronises the stream).  Per product:
 * hpcla_spmm_runs_k16_f64_* = hpcla_spmm_split_f64_* for k = 16 with ldb_own = ldb_ghost = ldc = 16 (row-major, 16-byte
 * aligned): the runs' B rows are staged into LDS by LDS-DMA next to the block's A entries and multiplied out of LDS in
 * stored order -- bit-identical results, 0.466 ms against 0.517 ms on the 5-point matrix x 16 (0.72 of the HBM peak).
 * Blocks marked as not fitting take a slow per-entry path: use these entries when (nearly) all blocks fit, else the
 * gather kernel.  B_ghost == NULL: every column is owned.  `block_list` as in hpcla_spmm_split_f64_*. */
int64_t hpcla_spmm_runs_desc_bytes(int64_t nrows);
int hpcla_spmm_runs_build_i32(const int32_t *rowptr, const int32_t *colval_split, int64_t nrows, int64_t nnz,
                              int index_base, int64_t n_own, void *desc, int64_t *n_fit_host, void *stream);
int hpcla_spmm_runs_build_i64(const int64_t *rowptr, const int64_t *colval_split, int64_t nrows, int64_t nnz,
                              int index_base, int64_t n_own, void *desc, int64_t *n_fit_host, void *stream);
/* BANDED test of a structure (plan time, one pass, synchronises the stream): the number of hpcla_spmm_rows_per_block()-row
 * blocks whose entries (at most 512) touch at most `max_runs` contiguous runs of split columns, in *n_blocks_host.  Every
 * stencil qualifies (5-point: 3 runs, 7-point: 5); a random pattern has about as many runs as entries.  Column-major callers
 * use it to choose between the product on their blocks as they are (hpcla_spmm_split_colmajor_*: contiguous pieces per gather
 * on a banded structure) and the layout conversions around the row-major kernels. */
int hpcla_spmm_banded_blocks_i32(const int32_t *rowptr, const int32_t *colval_split, int64_t nrows, int64_t nnz,
                                 int index_base, int64_t n_own, int max_runs, int64_t *n_blocks_host, void *stream);
int hpcla_spmm_banded_blocks_i64(const int64_t *rowptr, const int64_t *colval_split, int64_t nrows, int64_t nnz,
                                 int index_base, int64_t n_own, int max_runs, int64_t *n_blocks_host, void *stream);
int hpcla_spmm_runs_k16_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                const double *B_own, const double *B_ghost, int64_t n_own, double *C, int64_t nrows,
                                int64_t nnz, int index_base, const void *desc, const int32_t *block_list,
                                int64_t n_blocks, void *stream);
int hpcla_spmm_runs_k16_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                const double *B_own, const double *B_ghost, int64_t n_own, double *C, int64_t nrows,
                                int64_t nnz, int index_base, const void *desc, const int32_t *block_list,
                                int64_t n_blocks, void *stream);
/* The run tiles on a COLUMN-major caller's blocks (Julia's Matrix, src/dense.jl:63; round 5): hpcla_spmm_split_colmajor_f64_*
 * for k = 16 with the descriptors of hpcla_spmm_runs_build_* -- B_own (n_own rows, ALWAYS the row count of B_own here, also
 * with B_ghost == NULL) and C column-major with leading dimensions ldb_own / ldc, B_ghost the plan's row-major ghost segment
 * (ldb_ghost >= 16) or NULL.  Every own run is widened to an even first row and an even end so that the 16-byte LDS-DMA
 * applies to a column's run: B_own must be 16-byte aligned and ldb_own even (HPCLA_ERR_UNSUPPORTED otherwise: take
 * hpcla_spmm_split_colmajor_f64_*).  Same sums in the same order, same bits.  `block_list` over blocks of
 * hpcla_spmm_rows_per_block() rows, as in hpcla_spmm_runs_k16_f64_*. */
int hpcla_spmm_runs_colmajor_k16_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                         const double *B_own, int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost,
                                         int64_t n_own, double *C, int64_t ldc, int64_t nrows, int64_t nnz, int index_base,
                                         const void *desc, const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmm_runs_colmajor_k16_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                         const double *B_own, int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost,
                                         int64_t n_own, double *C, int64_t ldc, int64_t nrows, int64_t nnz, int index_base,
                                         const void *desc, const int32_t *block_list, int64_t n_blocks, void *stream);
/* Plan-time block order of hpcla_spmm_runs_colmajor_k16_f64_*, by measurement (the twin of hpcla_spmm_tune_block_order_f64_*,
 * whose launch is the row-major gather kernel): the product's own arguments; the launch is timed under the natural order and
 * groups of 8 / 32 / 128 / 512 row blocks, the fastest stays set for `rowptr` (hpcla_spmm_block_order_hint overrides it),
 * *chosen_group receives it (1 = natural).  C receives the product.  Synchronises the stream; launches of fewer than 4096
 * blocks are run once and keep the natural order. */
int hpcla_spmm_runs_colmajor_tune_block_order_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                                      const double *B_own, int64_t ldb_own, const double *B_ghost,
                                                      int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc, int64_t nrows,
                                                      int64_t nnz, int index_base, const void *desc, const int32_t *block_list,
                                                      int64_t n_blocks, void *stream, int *chosen_group);
int hpcla_spmm_runs_colmajor_tune_block_order_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                                      const double *B_own, int64_t ldb_own, const double *B_ghost,
                                                      int64_t ldb_ghost, int64_t n_own, double *C, int64_t ldc, int64_t nrows,
                                                      int64_t nnz, int index_base, const void *desc, const int32_t *block_list,
                                                      int64_t n_blocks, void *stream, int *chosen_group);
/* A * B on the CALLER's column-major blocks (Julia's Matrix, src/dense.jl:63), no layout conversion (csrc/colmajor.hip):
 * lanes = rows, so the 64 rows of a wave read one contiguous run of a column per gather instruction wherever the matrix is
 * banded; same bits as the row-major kernels.  hpcla_spmm_csr_f64_* takes this path by itself when both layouts are
 * HPCLA_LAYOUT_COL.  Split form: B_own (n_own rows) and C (nrows rows) column-major with leading dimensions ldb_own / ldc;
 * B_ghost is the halo plan's ordinary ROW-major ghost segment (ldb_ghost >= k doubles per ghost row) or NULL; block_list
 * over blocks of hpcla_spmv_rows_per_block() rows.  For unstructured matrices column-major costs a line per (entry, column)
 * pair: convert to row-major rows there (hpcla_transpose_f64) and use hpcla_spmm_split_f64_*. */
int hpcla_spmm_split_colmajor_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                                      const double *B_own, int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost,
                                      int64_t n_own, double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                      int index_base, const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmm_split_colmajor_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                                      const double *B_own, int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost,
                                      int64_t n_own, double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                      int index_base, const int32_t *block_list, int64_t n_blocks, void *stream);
/* hpcla_halo_begin for an operand in any layout (element (row i, column c) at x[i * x_rs + c * x_cs]; a column-major
 * block: x_rs = 1, x_cs = ld): the rows the plan SENDS are staged into `stage` (row-major like the exchange: at least
 * (largest send index + 1) * width doubles, the caller's, reused from call to call) on `stream`, then the ordinary
 * exchange is posted from `stage`.  hpcla_halo_end / status / ghost pointer as usual. */
int hpcla_halo_begin_strided_f64(hpcla_halo_plan_t *plan, const double *x, int64_t x_rs, int64_t x_cs, double *stage,
                                 void *stream);
/* layout conversion for column-major callers (Julia Matrix): dst(row-major, ld=k) <- src */
/* One column PANEL of a product in panel order (opt-in order of the distributed SpMM; the reference's column loop
 * src/sparse.jl:2391-2413 runs one exchange + SpMV per column instead).  (rowptr, colval_split, nzval) hold the
 * entries of A whose columns lie in the panel, in stored order; columns < n_own index B_own, the others B_ghost
 * (n_own = 0: every column is a ghost position).  accumulate = 1: every C(r, c) continues from its current value,
 * entry by entry -- a product taken panel by panel is the reference's sum in a different ORDER (1e-12 relative,
 * not bit-identical).  Row-major B / C (ldb, ldc = row strides). */
int hpcla_spmm_panel_f64_i32(const int32_t *rowptr, const int32_t *colval_split, const double *nzval,
                             const double *B_own, int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost,
                             int64_t n_own, double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                             int index_base, int accumulate, void *stream);
int hpcla_spmm_panel_f64_i64(const int64_t *rowptr, const int64_t *colval_split, const double *nzval,
                             const double *B_own, int64_t ldb_own, const double *B_ghost, int64_t ldb_ghost,
                             int64_t n_own, double *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                             int index_base, int accumulate, void *stream);
int hpcla_transpose_f64(const double *src, int64_t ld_src, int src_layout, double *dst,
                        int64_t ld_dst, int dst_layout, int64_t rows, int64_t cols, void *stream);

/* ---- dense mat-vec: replaces `LinearAlgebra.mul!(y.v, A.A, plan.gathered)` of
 * Base.:*(A::HPCMatrix, x::HPCVector) / mul! (src/dense.jl:614-658).  A: this rank's rows, ROW-major
 * with leading dimension lda >= ncols; the full x is read in place as three segments (slices owned by
 * lower ranks | own slice | slices of higher ranks), as delivered by a halo plan in which every rank
 * sends its whole slice to every other rank.  Summation order is a tree (BLAS order is
 * implementation-defined): parity by tolerance. */
int hpcla_gemv_rowmajor_f64(const double *A, int64_t lda, int64_t nrows, const double *x_lo, int64_t n_lo,
                            const double *x_own, int64_t n_own, const double *x_hi, int64_t n_hi,
                            double *y, void *stream);
/* transpose(A) * x for the dense row-partitioned A (Base.:*(At::Transpose{T,HPCMatrix}, x),
 * src/dense.jl:1210-1261): y_full[j] = sum_i A[i*lda + j] * x[i] over the LOCAL rows, all ncols
 * columns (the per-rank partial the reference all-reduces: follow with hpcla_allreduce_f64 and keep the
 * own column slice).  x is the slice of the vector on A's ROW partition.  `work`: device scratch of
 * hpcla_gemv_t_work_bytes(nrows, ncols) bytes.  Deterministic two-stage column sums. */
int64_t hpcla_gemv_t_work_bytes(int64_t nrows, int64_t ncols);
int hpcla_gemv_t_rowmajor_f64(const double *A, int64_t lda, int64_t nrows, int64_t ncols, const double *x,
                              double *y_full, void *work, void *stream);


/* ---- gather: replaces _gather_kernel! (src/vectors.jl:174-194) --------------------------------
 * out[dst[i]] = x[src[i]] (dst may be NULL = identity).  The SpMV hot path does not need it (split
 * column space); used for execute_plan! returning `gathered`, the SpGEMM value gather and the value
 * permutations of the TransposePlan (src/sparse.jl:1702-1845). */
int hpcla_gather_f64_i32(const double *x, const int32_t *src, const int32_t *dst, double *out,
                         int64_t n, int index_base, void *stream);
int hpcla_gather_f64_i64(const double *x, const int64_t *src, const int64_t *dst, double *out,
                         int64_t n, int index_base, void *stream);

/* ---- communicator: RCCL over xGMI, replaces comm_* on the hot path (src/backends.jl:264-327)
 * Bootstrap follows ext/HPCLinearAlgebraCUDAExt.jl:388-443: rank 0 gets a 128-byte unique id,
 * the host runtime (MPI.Bcast! / torch.distributed) broadcasts it, every rank calls init_rank.
 * nranks==1 creates a serial communicator that never loads librccl (CommSerial,
 * src/backends.jl:63). */
#define HPCLA_UNIQUE_ID_BYTES 128
int hpcla_comm_get_unique_id(uint8_t *id_host /* 128 bytes */);
int hpcla_comm_init_rank(hpcla_comm_t **comm, const uint8_t *id_host, int nranks, int rank);
/* flags: HPCLA_COMM_NO_RCCL = no RCCL communicator is created (id_host may be NULL); every data-path
 * collective of this communicator then goes through the peer windows below.  Needed where RCCL cannot
 * start -- it refuses two ranks on one GPU ("Duplicate GPU detected"), which is how the multi-rank GPU
 * tests run on a single-GPU box. */
#define HPCLA_COMM_NO_RCCL 1
int hpcla_comm_init_rank_ex(hpcla_comm_t **comm, const uint8_t *id_host, int nranks, int rank, int flags);
int hpcla_comm_rank(const hpcla_comm_t *comm, int *rank);
int hpcla_comm_size(const hpcla_comm_t *comm, int *nranks);
int hpcla_comm_destroy(hpcla_comm_t *comm);
/* in-place all-reduce of `count` doubles on `stream`: op 0 = sum, 1 = max, 2 = product
 * (comm_allreduce(comm, x, + | max | *), src/backends.jl:264-277). */
int hpcla_allreduce_f64(hpcla_comm_t *comm, double *buf, int64_t count, int op, void *stream);

/* ---- peer windows: one-sided transport over xGMI for the ranks of ONE node (csrc/window.hip) --------
 * Replaces, on the hot path, the MPI Isend/Irecv of execute_plan! (src/vectors.jl:431-455) and the scalar
 * comm_allreduce (src/backends.jl:264-277) -- and the RCCL group/all-reduce this library uses otherwise.
 * A window is a fine-grained device allocation that the other ranks map (hipIpcOpenMemHandle) and store
 * into directly; flags/acks in its control lines order producer and consumer, all spins are bounded.
 * Bootstrap is the unique-id pattern again (ext/HPCLinearAlgebraCUDAExt.jl:411-443): every rank EXPORTS a
 * HPCLA_WINDOW_DESC_BYTES descriptor, the host runtime all-gathers them (MPI.Allgather /
 * torch.distributed), every rank ATTACHES.  All ranks must be on one node (descriptor bytes 64..71 hold a
 * node identity the host layer compares before attaching; bytes 112..119 the identity of the GPU that holds
 * the window: attach refuses a peer whose device the runtime reports as not peer-accessible, instead of
 * mapping memory that would fault on the first store).
 *   communicator window: all-reduce of <= 8 doubles in ONE kernel (each rank stores its partial into its
 *     slot of every window, then sums its own window's slots in rank order: the same bits on every rank);
 *   halo plan window:    see hpcla_halo_plan_export below. */
#define HPCLA_WINDOW_DESC_BYTES 128
int hpcla_comm_window_export(hpcla_comm_t *comm, uint8_t *desc_host /* 128 bytes out */);
int hpcla_comm_window_attach(hpcla_comm_t *comm, const uint8_t *all_descs_host /* nranks x 128 bytes */);
/* Connection test after attach: all-reduce of (rank+1) through the windows with its own timeout;
 * *ok = 1 iff it arrived complete.  The host layer all-gathers `ok`; unless every rank passed, every rank
 * calls hpcla_comm_window_detach and the data path stays on RCCL.  Synchronises the device. */
int hpcla_comm_window_selftest(hpcla_comm_t *comm, double timeout_s, int *ok);
int hpcla_comm_window_detach(hpcla_comm_t *comm);
/* 1 in *timed_out if a window all-reduce gave up waiting for a peer (HPCLA_PUSH_TIMEOUT_S, default 300 s: a
 * slow peer is waited for like a blocking MPI_Allreduce, src/backends.jl:264-277; the bound only lets the grid
 * drain when a peer has died).  The result of an expired all-reduce is NaN on this rank.  Synchronises a
 * 4-byte device read. */
int hpcla_comm_status(hpcla_comm_t *comm, int *timed_out);
/* 64-bit identity of (node, physical device): two ranks with equal identities share one GPU. */
int hpcla_device_identity(int device, uint64_t *id);

/* Contiguous-range exchange: device form of execute_plan!(::VectorRepartitionPlan)
 * (src/vectors.jl:624-671; plan lists :511-620), of the dense row repartition (src/dense.jl:1711-1760)
 * and of the nzval leg of the sparse repartition (src/sparse.jl:4443-4535).  For each i < n_send the
 * range src[send_offsets[i] .. +send_counts[i]) goes to send_ranks[i]; for each i < n_recv,
 * recv_counts[i] units from recv_ranks[i] land at dst + recv_offsets[i]; the part that stays
 * (local_src_range / local_dst_offset of the reference plan) is one device copy.  All offsets and
 * counts are 0-based and in units of `width` doubles (1 for vectors and nzval, ncols for row-major
 * dense rows).  Lists are HOST arrays; src/dst are device pointers; one ncclGroup on `stream`. */
int hpcla_exchange_ranges_f64(hpcla_comm_t *comm, const double *src, double *dst, int n_send,
                              const int *send_ranks, const int64_t *send_offsets,
                              const int64_t *send_counts, int n_recv, const int *recv_ranks,
                              const int64_t *recv_offsets, const int64_t *recv_counts,
                              int64_t local_src_offset, int64_t local_dst_offset, int64_t local_count,
                              int width, void *stream);

/* ---- halo plan: device half of VectorPlan / execute_plan! (src/vectors.jl:229-251, 394-463)
 * Inputs are the reference plan's own lists (0-based here):
 *   send_ranks_host[i], send_counts_host[i], send_idx (device, int32 or int64, concatenated in
 *   send_ranks order; local indices into x.v  == plan.send_indices, src/sparse.jl:1939-1944);
 *   recv_ranks_host[i], recv_counts_host[i]: ghost segment i (plan.recv_perm[i]) has
 *   recv_counts[i] entries; segments are stored back to back in recv_ranks order in the ghost
 *   buffer.  Because col_indices is sorted (src/sparse.jl:501) and owners are contiguous rank
 *   ranges (src/sparse.jl:1890), that order IS ascending global column order.
 * `width` = values per index (1 for vectors, k for SpMM ghost rows, row-major).
 * The plan owns: packed send buffer, ghost buffer, one side stream, two events. */
int hpcla_halo_plan_create(hpcla_halo_plan_t **plan, hpcla_comm_t *comm, int n_send,
                           const int32_t *send_ranks_host, const int64_t *send_counts_host,
                           const void *send_idx, int idx_is_i64, int n_recv,
                           const int32_t *recv_ranks_host, const int64_t *recv_counts_host,
                           int width);
/* flags: HPCLA_HALO_SINGLE_BUFFER -- never double-buffer the ghost window.  Required for every plan that is
 * driven through hpcla_halo_begin / hpcla_halo_end with the ghost pointer taken from the host in between
 * (SpMM ghost rows: the interior launch is enqueued before the exchange has completed, so "the buffer of the
 * exchange completed last" is not yet this exchange's buffer); width > 1 plans are single-buffered anyway, the
 * flag matters for a one-column B (width 1).  Vector plans of the fused SpMV keep the default (0). */
#define HPCLA_HALO_SINGLE_BUFFER 1
int hpcla_halo_plan_create_ex(hpcla_halo_plan_t **plan, hpcla_comm_t *comm, int n_send,
                              const int32_t *send_ranks_host, const int64_t *send_counts_host,
                              const void *send_idx, int idx_is_i64, int n_recv,
                              const int32_t *recv_ranks_host, const int64_t *recv_counts_host,
                              int width, int flags);
/* Chain `plan` behind `leader`: both exchange on the LEADER's side stream, so exchanges begun one after the other
 * (hpcla_halo_begin(leader), hpcla_halo_begin(plan), ...) also RUN one after the other -- the chunk-sets of a
 * panel-ordered SpMM arrive in order instead of all at once.  The stream is shared by reference count and goes with
 * the LAST plan of the chain, so the plans may be destroyed in any order. */
int hpcla_halo_plan_chain(hpcla_halo_plan_t *plan, hpcla_halo_plan_t *leader);
int hpcla_halo_plan_destroy(hpcla_halo_plan_t *plan);
/* Push transport for this plan.  When the communicator's window is attached, create() places the ghost
 * segment (double-buffered up to 64 MiB) inside a peer-mappable window.  export: this rank's descriptor
 * plus a table of HPCLA_WINDOW_TABLE_ROWS x nranks int64 (row 0: flag line of source rank r, row 1: offset
 * of r's segment in my ghost, row 2: ack line of consumer rank r, row 3: entries expected from r; -1 =
 * none).  The host runtime all-gathers descriptors and tables; attach maps the neighbours' windows and
 * builds the push lists.  Collective: ranks without neighbours pass a zeroed descriptor / -1 table.
 * After attach, HPCLA_HALO_MODE unset or "push" selects the push transport for this plan. */
#define HPCLA_WINDOW_TABLE_ROWS 4
int hpcla_halo_plan_export(hpcla_halo_plan_t *plan, uint8_t *desc_host, int64_t *table_host);
int hpcla_halo_plan_attach(hpcla_halo_plan_t *plan, const uint8_t *all_descs_host,
                           const int64_t *all_tables_host);
/* give up the push transport of this plan (some rank failed to attach): RCCL receives into the ghost */
int hpcla_halo_plan_detach(hpcla_halo_plan_t *plan);
/* Connection test of one plan on the real topology, run by the host runtime right after attach (collective
 * among the ranks that hold a plan: each makes the same TWO exchanges, one per ghost buffer of a
 * double-buffered plan).  The exchanged vector is x[i*width + j] = rank * 2^40 + i*width + j (+ 0.25 in
 * round two; `n_local_rows` rows); afterwards ghost slot check_slots[t] must hold row check_rows[t] of the
 * rank whose segment the slot lies in (the host knows both from the reference plan: recv_perm and the
 * x partition, src/sparse.jl:1938-1953).  *ok = 1: every checked value arrived and nothing timed out;
 * *ok = 0: hpcla_last_error() says what failed -- the runtime then all-gathers the verdicts and detaches
 * the plan on every rank (hpcla_halo_plan_detach), leaving it on RCCL.  Works for RCCL plans too. */
int hpcla_halo_plan_probe(hpcla_halo_plan_t *plan, int64_t n_local_rows, const int64_t *check_slots_host,
                          const int64_t *check_rows_host, int64_t n_check, void *stream, int *ok);
/* 1 in *timed_out if a push or wait of this plan gave up; synchronising 4-byte read.  An expired WAIT poisons
 * what it would have fed: the waiting row blocks' y entries and dot partials (fused SpMV) or the whole ghost
 * buffer (hpcla_halo_end) become NaN, so the failure cannot pass as a result; an expired ack wait of a PUSH
 * stores nothing and publishes nothing (the consumer may still be reading the buffer).  The status is sticky:
 * the plan is dead afterwards. */
int hpcla_halo_status(hpcla_halo_plan_t *plan, int *timed_out);
/* device pointer of the ghost buffer of the exchange completed last (n_ghost*width doubles) and its length in
 * indices.  Constant for RCCL plans, for dense (width > 1) push plans and for HPCLA_HALO_SINGLE_BUFFER plans;
 * a double-buffered vector push plan reads its device step counter here, i.e. the call SYNCHRONISES the device
 * and is only meaningful AFTER hpcla_halo_end of the exchange in question -- only the API-parity path
 * (execute_plan!'s `gathered`) asks, the fused SpMV finds its buffer in the kernel.  NEVER inside a collective when
 * ranks are threads of one process (a device-wide wait then covers the other ranks' waiting kernels: round 6 found
 * hpcla_halo_plan_probe stalling that way); plans driven through halo_begin / halo_end should be created with
 * HPCLA_HALO_SINGLE_BUFFER, whose pointer is a constant fetched once at plan time. */
int hpcla_halo_ghost_ptr(hpcla_halo_plan_t *plan, double **ghost, int64_t *n_ghost);
/* begin: after everything already enqueued on `stream`, pack x[send_idx] and post the
 * ncclSend/ncclRecv group on the plan's side stream (tag-21 exchange, src/vectors.jl:431-446).
 * end: make `stream` wait for the exchange.  Work enqueued on `stream` between begin and end
 * (the interior SpMV) overlaps the exchange.  x has leading dimension `width` (row-major). */
int hpcla_halo_begin(hpcla_halo_plan_t *plan, const double *x, void *stream);
int hpcla_halo_end(hpcla_halo_plan_t *plan, void *stream);

/* Fused distributed SpMV  y = A*x  (Base.:*(A,x), src/sparse.jl:2096-2128; mul!, :2019-2037).
 * Three orderings, chosen by the environment variable HPCLA_HALO_MODE when the library is first used:
 *   "push"   (default when the plan's peer windows are attached) a push kernel stores the boundary
 *                       values into the neighbours' ghost windows, then ONE launch = interior blocks
 *                       followed by the boundary blocks, which wait in-kernel for the neighbours' flags;
 *                       interior_blocks + boundary_blocks must cover every row block;
 *   "serial" (default otherwise) the RCCL send/recv group on `stream`, then ONE launch over all row blocks;
 *   "overlap"           exchange + boundary blocks on the plan's side stream, interior blocks on
 *                       `stream`, joined by events (block lists from hpcla_classify_blocks_*).
 * On MI355X the SpMV kernel fills every wave slot of every CU, so a side-stream RCCL kernel only runs
 * once the interior grid drains; measured "overlap" = +29..32 us, "serial" = +13..14 us per 4096^2 step
 * (profiles/r01_halo_mode_experiments.log).  Both give the same bits.  With plan==NULL or no
 * neighbours it is a plain split SpMV. */
/* Run-time override of HPCLA_HALO_MODE for plans used from now on: -1 automatic (push where attached,
 * else serial), 0 serial, 1 overlap, 2 push.  Must be called with the same value on every rank, between
 * steps (used by bench.py to time every ordering in one run). */
int hpcla_set_halo_mode(int mode);
int hpcla_spmv_dist_f64_i32(hpcla_halo_plan_t *plan, const int32_t *rowptr,
                            const int32_t *colval_split, const double *nzval, const double *x,
                            int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base,
                            const int32_t *interior_blocks, int64_t n_interior,
                            const int32_t *boundary_blocks, int64_t n_boundary, void *stream);
int hpcla_spmv_dist_f64_i64(hpcla_halo_plan_t *plan, const int64_t *rowptr,
                            const int64_t *colval_split, const double *nzval, const double *x,
                            int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base,
                            const int32_t *interior_blocks, int64_t n_interior,
                            const int32_t *boundary_blocks, int64_t n_boundary, void *stream);

/* Fused y = A*x and out = x.y (CG's p.Ap; SURVEY section 7 step 6 "SpMV+dot fusion"): the SpMV
 * workgroups leave one partial per row block in `work` (hpcla_spmv_dot_work_bytes(nrows) bytes),
 * summed in index order (deterministic) and all-reduced over `comm` into out_dev[0].
 * Requires x partitioned like A's rows (n_own == nrows). */
int64_t hpcla_spmv_dot_work_bytes(int64_t nrows);
int hpcla_spmv_dist_dot_f64_i32(hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const int32_t *rowptr,
                                const int32_t *colval_split, const double *nzval, const double *x,
                                int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base,
                                const int32_t *interior_blocks, int64_t n_interior,
                                const int32_t *boundary_blocks, int64_t n_boundary,
                                double *dot_out_dev, void *work, void *stream);
int hpcla_spmv_dist_dot_f64_i64(hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const int64_t *rowptr,
                                const int64_t *colval_split, const double *nzval, const double *x,
                                int64_t n_own, double *y, int64_t nrows, int64_t nnz, int index_base,
                                const int32_t *interior_blocks, int64_t n_interior,
                                const int32_t *boundary_blocks, int64_t n_boundary,
                                double *dot_out_dev, void *work, void *stream);

/* ---- OPT-IN packed copy of the matrix for SpMV (no reference counterpart) ----------------------------
 * For matrices with <= 256 distinct values whose row blocks reach only columns within +-32 K of the
 * block's first row (stencils, graph Laplacians ...), a plan-time copy with 16-bit block-relative
 * columns and 8-bit value codes moves 3 B per stored entry instead of 12.  The products are the same
 * fp64 numbers summed in the same order, so results stay bit-identical to the CSR path.
 * create returns HPCLA_ERR_UNSUPPORTED when the matrix (restricted to `block_list`, NULL = all row
 * blocks) is not packable.  Int32 indices only.  Reported separately from the CSR numbers. */
typedef struct hpcla_packed hpcla_packed_t;
int hpcla_packed_create_i32(hpcla_packed_t **out, const int32_t *rowptr, const int32_t *colval_split,
                            const double *nzval, int64_t nrows, int64_t nnz, int64_t n_own,
                            int index_base, const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_packed_destroy(hpcla_packed_t *p);
int hpcla_packed_info(const hpcla_packed_t *p, int64_t *bytes, int *ndict);
int hpcla_spmv_packed_f64_i32(const hpcla_packed_t *p, const int32_t *rowptr, const double *x, double *y,
                              int index_base, const int32_t *block_list, int64_t n_blocks,
                              double *dot_partial, void *stream);
/* distributed form of hpcla_spmv_dist_* / hpcla_spmv_dist_dot_* (dot_out_dev may be NULL): interior
 * row blocks from the packed copy, boundary row blocks (ghost columns) from the CSR arrays. */
int hpcla_spmv_dist_packed_f64_i32(hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const hpcla_packed_t *p,
                                   const int32_t *rowptr, const int32_t *colval_split,
                                   const double *nzval, const double *x, int64_t n_own, double *y,
                                   int64_t nrows, int64_t nnz, int index_base,
                                   const int32_t *interior_blocks, int64_t n_interior,
                                   const int32_t *boundary_blocks, int64_t n_boundary,
                                   double *dot_out_dev, void *work, void *stream);

/* ---- reductions: replace dot / norm (src/vectors.jl:798-812, 758-780) --------------------------
 * Local deterministic two-stage reduction into out_dev[0] (device double), then, if comm has
 * more than one rank, an in-place RCCL all-reduce on the same stream.  No host sync: CG keeps the
 * scalar on the device.  `work` is a caller-provided device scratch of hpcla_reduce_work_bytes().
 *  dot:   sum x[i]*y[i]
 *  nrm2sq: sum x[i]^2 (the reference squares the local 2-norm before summing, :764-765; the
 *          caller takes sqrt), asum: sum |x[i]| (p=1), amax: max |x[i]| (p=Inf). */
int64_t hpcla_reduce_work_bytes(void);
int hpcla_dot_f64(hpcla_comm_t *comm, const double *x, const double *y, int64_t n, double *out_dev,
                  void *work, void *stream);
int hpcla_nrm2sq_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev, void *work,
                     void *stream);
int hpcla_asum_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev, void *work,
                   void *stream);
int hpcla_amax_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev, void *work,
                   void *stream);
/* maximum(v) (negate = 0) / minimum(v) (negate = 1: out = max(-x), the caller flips the sign)
 * (src/vectors.jl:815-836); an empty vector contributes -inf like typemin in the reference */
int hpcla_maxval_f64(hpcla_comm_t *comm, const double *x, int64_t n, int negate, double *out_dev, void *work,
                     void *stream);
/* sum(v) (src/vectors.jl:838-845): out = sum of all elements over all ranks */
int hpcla_sum_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev, void *work, void *stream);
/* prod(v), src/vectors.jl:853-858 (an empty local part contributes 1) */
int hpcla_prod_f64(hpcla_comm_t *comm, const double *x, int64_t n, double *out_dev, void *work, void *stream);
/* general p-norm, p > 0 finite: out = sum |x_i|^p over all ranks; the caller takes the 1/p power
 * (norm(v, p), src/vectors.jl:774-779) */
int hpcla_powsum_f64(hpcla_comm_t *comm, const double *x, int64_t n, double p, double *out_dev, void *work,
                     void *stream);

/* ---- vector updates: replace u+v, a*v and fused broadcast (src/vectors.jl:868-903, 944-964,
 * 1203-1226).  The scalar is alpha_host * (num_dev ? *num_dev : 1) / (den_dev ? *den_dev : 1),
 * so CG coefficients computed by the reductions above never visit the host.
 *  axpy: y[i] = y[i] + a*x[i]      xpay: y[i] = x[i] + a*y[i]
 *  scale: y[i] = a*x[i]            axpby: z[i] = a*x[i] + b*y[i] (host scalars)
 *  divide: y[i] = x[i] / a  (v / a, src/vectors.jl:960-964; a true division, not a reciprocal) */
int hpcla_axpy_f64(double alpha_host, const double *num_dev, const double *den_dev, const double *x,
                   double *y, int64_t n, void *stream);
int hpcla_xpay_f64(const double *x, double alpha_host, const double *num_dev, const double *den_dev,
                   double *y, int64_t n, void *stream);
int hpcla_scale_f64(double alpha_host, const double *x, double *y, int64_t n, void *stream);
/* Fused CG update (two broadcasts + one norm of the reference, src/vectors.jl:1203-1226, 758-765):
 * a = alpha_host * *num_dev / *den_dev;  x += a*p;  r -= a*Ap;  rr_out_dev[0] = allreduce(sum r^2).
 * `work` as for the reductions (hpcla_reduce_work_bytes()). */
int hpcla_cg_update_f64(hpcla_comm_t *comm, double alpha_host, const double *num_dev,
                        const double *den_dev, const double *p, const double *Ap, double *x, double *r,
                        int64_t n, double *rr_out_dev, void *work, void *stream);
/* The same iteration with the x update deferred to the direction update (p is then read once per iteration
 * instead of twice; identical operations per element, identical bits):
 *   residual:  a = alpha_host * *num_dev / *den_dev;  r -= a*Ap;  rr_out_dev[0] = allreduce(sum r^2)
 *   direction: a = alpha_host * *a_num_dev / *a_den_dev, b = beta_host * *b_num_dev / *b_den_dev;
 *              x += a*p;  p = r + b*p      (null num/den pointers count as 1) */
int hpcla_cg_residual_f64(hpcla_comm_t *comm, double alpha_host, const double *num_dev, const double *den_dev,
                          const double *Ap, double *r, int64_t n, double *rr_out_dev, void *work, void *stream);
int hpcla_cg_direction_f64(double alpha_host, const double *a_num_dev, const double *a_den_dev, double beta_host,
                           const double *b_num_dev, const double *b_den_dev, const double *r, double *x,
                           double *p, int64_t n, void *stream);
/* `iters` fused CG iterations enqueued by ONE host call (no reference counterpart: the reference has no
 * Krylov solver, a caller composes the iteration from A*p src/sparse.jl:2096-2128, dot src/vectors.jl:798-812,
 * the broadcasts :1203-1226 and norm :758-765 -- SURVEY 3.4).  Per iteration exactly the launches of
 * hpcla_spmv_dist_dot_* (Ap = A*p, pAp), hpcla_cg_residual_f64 and hpcla_cg_direction_f64, with the same
 * arguments, so the same bits as the three separate calls; neither Python nor Julia sits in the loop.
 * rr_hist_dev: iters + 1 doubles, [0] = sum r_0^2 on entry, [j] = sum r_j^2 on return (device memory; the
 * caller takes the square roots).  pAp_dev: 1 double.  dot_work: hpcla_spmv_dot_work_bytes(nrows) bytes,
 * reduce_work: hpcla_reduce_work_bytes() bytes.  x, r, p, Ap partitioned like A's rows (nrows entries,
 * 16-byte aligned); plan / comm / block lists as for hpcla_spmv_dist_dot_*.  Only enqueues: capturable. */
int hpcla_cg_iterations_f64_i32(hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const int32_t *rowptr,
                                const int32_t *colval_split, const double *nzval, int64_t nrows, int64_t nnz,
                                int index_base, const int32_t *interior_blocks, int64_t n_interior,
                                const int32_t *boundary_blocks, int64_t n_boundary, double *x, double *r,
                                double *p, double *Ap, double *rr_hist_dev, double *pAp_dev, void *dot_work,
                                void *reduce_work, int iters, void *stream);
int hpcla_cg_iterations_f64_i64(hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const int64_t *rowptr,
                                const int64_t *colval_split, const double *nzval, int64_t nrows, int64_t nnz,
                                int index_base, const int32_t *interior_blocks, int64_t n_interior,
                                const int32_t *boundary_blocks, int64_t n_boundary, double *x, double *r,
                                double *p, double *Ap, double *rr_hist_dev, double *pAp_dev, void *dot_work,
                                void *reduce_work, int iters, void *stream);
int hpcla_divide_f64(const double *x, double a_host, double *y, int64_t n, void *stream);
int hpcla_axpby_f64(double a, const double *x, double b, const double *y, double *z, int64_t n,
                    void *stream);

/* ---- sparse A +/- B value pass: the reference's five index-mapped kernels (_copy_a_only_/
 * _copy_b_only_/_negate_b_only_/_add_both_/_sub_both_kernel!, src/sparse.jl:1258-1303) as ONE pass
 * over the merged pattern: out[i] = a[ia[i]] (+|-) b[ib[i]] where both indices are >= 0, a copy of
 * a[ia[i]] or of (+|-) b[ib[i]] where only one is (never an addition to zero), for i in [0, n).
 * ia/ib (0-based positions in the operands' nzval, -1 = absent) come from the host AdditionPlan
 * (union of the two sparsity patterns, src/sparse.jl:1112-1245).  subtract: 0 = A+B, 1 = A-B. */
int hpcla_merge_combine_f64_i32(double *out, const double *a, const int32_t *ia, const double *b,
                                const int32_t *ib, int64_t n, int subtract, void *stream);
int hpcla_merge_combine_f64_i64(double *out, const double *a, const int64_t *ia, const double *b,
                                const int64_t *ib, int64_t n, int subtract, void *stream);

/* ---- synthetic inputs on the device (bench/tests): the SURVEY section 8d counter-based
 * generator, v[i] = u01(seed, start+i). */
int hpcla_fill_uniform_f64(double *v, int64_t start, int64_t count, uint64_t seed, void *stream);

/* ==== Float32 element type on the hot path (csrc/f32.hip) ======================================================
 * The reference is generic in T; its GPU test configurations are CUDA x {Float32, Float64} and Metal x Float32
 * (test/test_utils.jl:62-80), all through the same _spmv_kernel! (src/sparse.jl:2055-2066: acc = zero(T),
 * acc += nzval[j] * x[colval[j]] in T), the same column loop for A * HPCMatrix (:2391-2413) and the same local
 * dot / nrm2 + scalar all-reduce (src/vectors.jl:758-812).  Float64 is the graded type; these entries give the same
 * path to a Float32 backend: row sums in float, stored order, separate multiply and add (the reference's bits).
 *
 * spmv_csr / spmv_split / spmm_csr / spmm_split: arguments as for the _f64 entries of the same name, with float values,
 * operands and results.  GHOSTS ARE DOUBLES: the halo transports move 8-byte words, so a Float32 exchange widens what it
 * sends (float -> double is exact) and the split kernels narrow the ghost values they gather (exact again); x_ghost_wide
 * / B_ghost_wide are the plan's ordinary ghost segment (hpcla_halo_ghost_ptr).  spmm_split is row-major like its _f64
 * twin; spmm_csr takes either layout for B and C (Julia's column-major Matrix goes in untouched). */
int hpcla_spmv_csr_f32_i32(const int32_t *rowptr, const int32_t *colval, const float *nzval, const float *x, float *y,
                           int64_t nrows, int64_t nnz, int index_base, void *stream);
int hpcla_spmv_csr_f32_i64(const int64_t *rowptr, const int64_t *colval, const float *nzval, const float *x, float *y,
                           int64_t nrows, int64_t nnz, int index_base, void *stream);
int hpcla_spmv_split_f32_i32(const int32_t *rowptr, const int32_t *colval_split, const float *nzval, const float *x_own,
                             const double *x_ghost_wide, int64_t n_own, float *y, int64_t nrows, int64_t nnz,
                             int index_base, const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmv_split_f32_i64(const int64_t *rowptr, const int64_t *colval_split, const float *nzval, const float *x_own,
                             const double *x_ghost_wide, int64_t n_own, float *y, int64_t nrows, int64_t nnz,
                             int index_base, const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmm_csr_f32_i32(const int32_t *rowptr, const int32_t *colval, const float *nzval, const float *B, int64_t ldb,
                           int b_layout, float *C, int64_t ldc, int c_layout, int64_t nrows, int64_t nnz, int k,
                           int index_base, void *stream);
int hpcla_spmm_csr_f32_i64(const int64_t *rowptr, const int64_t *colval, const float *nzval, const float *B, int64_t ldb,
                           int b_layout, float *C, int64_t ldc, int c_layout, int64_t nrows, int64_t nnz, int k,
                           int index_base, void *stream);
int hpcla_spmm_split_f32_i32(const int32_t *rowptr, const int32_t *colval_split, const float *nzval, const float *B_own,
                             int64_t ldb_own, const double *B_ghost_wide, int64_t ldb_ghost, int64_t n_own, float *C,
                             int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base, const int32_t *block_list,
                             int64_t n_blocks, void *stream);
int hpcla_spmm_split_f32_i64(const int64_t *rowptr, const int64_t *colval_split, const float *nzval, const float *B_own,
                             int64_t ldb_own, const double *B_ghost_wide, int64_t ldb_ghost, int64_t n_own, float *C,
                             int64_t ldc, int64_t nrows, int64_t nnz, int k, int index_base, const int32_t *block_list,
                             int64_t n_blocks, void *stream);
/* column-major own block and result (Julia's Matrix), the plan's row-major ghost segment: hpcla_spmm_split_colmajor_f64_*'s
 * twin for Float32; hpcla_halo_begin_strided_f32 is hpcla_halo_begin_strided_f64's (the staged values are widened) */
int hpcla_spmm_split_colmajor_f32_i32(const int32_t *rowptr, const int32_t *colval_split, const float *nzval,
                                      const float *B_own, int64_t ldb_own, const double *B_ghost_wide, int64_t ldb_ghost,
                                      int64_t n_own, float *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                      int index_base, const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_spmm_split_colmajor_f32_i64(const int64_t *rowptr, const int64_t *colval_split, const float *nzval,
                                      const float *B_own, int64_t ldb_own, const double *B_ghost_wide, int64_t ldb_ghost,
                                      int64_t n_own, float *C, int64_t ldc, int64_t nrows, int64_t nnz, int k,
                                      int index_base, const int32_t *block_list, int64_t n_blocks, void *stream);
int hpcla_halo_begin_strided_f32(hpcla_halo_plan_t *plan, const float *x, int64_t x_rs, int64_t x_cs, double *stage,
                                 void *stream);
/* Distributed y = A*x for Float32 (Base.:*(A, x), src/sparse.jl:2096-2128) in one call: hpcla_halo_begin_f32, the interior
 * blocks (they overlap the exchange), hpcla_halo_end, the boundary blocks with the plan's ghost segment.  Arguments as for
 * hpcla_spmv_dist_f64_* plus `stage` (hpcla_halo_begin_f32).  A plan with attached peer windows must have been created with
 * HPCLA_HALO_SINGLE_BUFFER.  plan == NULL or no neighbours: a plain product over all row blocks. */
int hpcla_spmv_dist_f32_i32(hpcla_halo_plan_t *plan, const int32_t *rowptr, const int32_t *colval_split, const float *nzval,
                            const float *x, int64_t n_own, float *y, int64_t nrows, int64_t nnz, int index_base,
                            const int32_t *interior_blocks, int64_t n_interior, const int32_t *boundary_blocks,
                            int64_t n_boundary, double *stage, void *stream);
int hpcla_spmv_dist_f32_i64(hpcla_halo_plan_t *plan, const int64_t *rowptr, const int64_t *colval_split, const float *nzval,
                            const float *x, int64_t n_own, float *y, int64_t nrows, int64_t nnz, int index_base,
                            const int32_t *interior_blocks, int64_t n_interior, const int32_t *boundary_blocks,
                            int64_t n_boundary, double *stage, void *stream);
/* hpcla_halo_begin for a Float32 operand (execute_plan!, src/vectors.jl:394-463): widens x at the plan's send positions
 * into `stage` (doubles, laid out like x: at least (largest send index + 1) * width entries; the caller's, reused from
 * call to call) on `stream`, then posts the ordinary exchange from `stage`.  hpcla_halo_end / hpcla_halo_status /
 * hpcla_halo_ghost_ptr as usual: every transport is the Float64 one. */
int hpcla_halo_begin_f32(hpcla_halo_plan_t *plan, const float *x, double *stage, void *stream);
/* dot / norm pieces (src/vectors.jl:758-812): products and squares are formed and summed in DOUBLE (exact products, two
 * deterministic stages, scalar all-reduce in double); out_dev is a device double, the caller rounds to Float32 once.
 * work: hpcla_reduce_work_bytes().  16-byte aligned operands. */
int hpcla_dot_f32(hpcla_comm_t *comm, const float *x, const float *y, int64_t n, double *out_dev, void *work, void *stream);
int hpcla_nrm2sq_f32(hpcla_comm_t *comm, const float *x, int64_t n, double *out_dev, void *work, void *stream);
int hpcla_asum_f32(hpcla_comm_t *comm, const float *x, int64_t n, double *out_dev, void *work, void *stream);
int hpcla_amax_f32(hpcla_comm_t *comm, const float *x, int64_t n, double *out_dev, void *work, void *stream);
int hpcla_sum_f32(hpcla_comm_t *comm, const float *x, int64_t n, double *out_dev, void *work, void *stream);
int hpcla_maxval_f32(hpcla_comm_t *comm, const float *x, int64_t n, int negate, double *out_dev, void *work, void *stream);
/* z = a*x + b*y, y = a*x, y = x / a in float (src/vectors.jl:868-903, 944-964), separate multiply and add; 16-byte
 * aligned operands. */
/* layout conversion of a Float32 block (hpcla_transpose_f64's twin): column-major Julia Matrix <-> the library's row-major
 * rows, whose ghost rows travel as contiguous k-value pieces in ONE exchange */
int hpcla_transpose_f32(const float *src, int64_t ld_src, int src_layout, float *dst, int64_t ld_dst, int dst_layout,
                        int64_t rows, int64_t cols, void *stream);
int hpcla_axpby_f32(float a, const float *x, float b, const float *y, float *z, int64_t n, void *stream);
int hpcla_scale_f32(float a, const float *x, float *y, int64_t n, void *stream);
int hpcla_divide_f32(const float *x, float a, float *y, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HPCLA_ROCM_H */
