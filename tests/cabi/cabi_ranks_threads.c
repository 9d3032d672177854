/* cabi_ranks_threads.c -- R ranks (default 8, the rank count of BASELINE's multi-GPU configurations) hosted by ONE
 * process, one host thread per rank, all on device 0: the peer-window transport of libhpcla_rocm.so at the target's
 * rank count and neighbour count, through include/hpcla_rocm.h alone.
 *
 * Why threads: the GPU boxes of this build allow at most 6 processes on a card, so 8 one-process-per-rank workers cannot
 * be started there (tests/test_gpu_multirank.py stops at 5 ranks + the test runner); a window exported by a process is
 * reachable by the other ranks of that process through the exporter's own pointer (csrc/window.hip, window_open), so
 * everything rank-count dependent on the DEVICE side -- the communicator's 2 x nranks all-reduce slots, 7 flag lines and
 * 7 ack lines per plan, 7 push targets with their chunk maps, the epoch counters with 8 ranks arriving at different
 * times -- runs exactly as it would with 8 processes.  Every rank drives its own non-blocking HIP stream (ranks on one
 * null stream would queue behind each other's waiting kernels).
 *
 * Two matrices, the two exchange shapes of the reference's plans (src/sparse.jl:1875-1984):
 *   "alltoall": every row has one entry in EVERY rank's column slice (config 5's shape: each rank has R-1 send and R-1
 *               recv neighbours, scattered send lists);
 *   "slab":     2-D 5-point Poisson, row slabs (configs 3/4: the ranks next door, contiguous sends).
 * Checked per rank: the plan's connection test (hpcla_halo_plan_probe, every ghost slot), 6 fused distributed SpMVs with
 * x changing from step to step bit-exact against a scalar CPU loop in stored order (the loop of src/sparse.jl:2055-2066),
 * dot(x, y) through the window all-reduce with identical bits on all ranks, A*B with k = 16 through a width-16 plan
 * (begin / interior blocks / end / boundary blocks), no spin timed out.
 * Exit code 0 = every rank passed.  Build/run: tests/test_cabi_from_c.py. */
#define _GNU_SOURCE
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hpcla_rocm.h"

#define MAXR 16
#define MAXROW 24

static int R = 8;
static pthread_barrier_t g_bar;
static int g_fail = 0;                                   /* set by any rank; read after a barrier */

/* "all-gather" boards: a rank writes its row, barrier, everybody reads */
static uint8_t g_desc[MAXR * HPCLA_WINDOW_DESC_BYTES];
static int64_t g_tab[MAXR * HPCLA_WINDOW_TABLE_ROWS * MAXR];
static int g_ok[MAXR];
static double g_dots[MAXR][8];

static void barrier(void) { pthread_barrier_wait(&g_bar); }
/* where every rank is (kind * 1000 + phase * 10 + detail): printed by a rank whose connection test fails, so that a timeout
 * names the rank the others were waiting for and the call it was in */
static volatile int g_stage[MAXR];
static int g_kind_now[MAXR];
#define STAGE(phase) (g_stage[rank] = g_kind_now[rank] * 1000 + (phase))
static void print_stages(int rank, const char *what)
{
    char line[512];
    int n = snprintf(line, sizeof(line), "rank %d: %s; stages:", rank, what);
    for (int r = 0; r < R && n < (int)sizeof(line) - 16; ++r) n += snprintf(line + n, sizeof(line) - (size_t)n, " r%d=%d", r, g_stage[r]);
    fprintf(stderr, "%s\n", line);
}

static uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double u01(uint64_t seed, int64_t i)               /* hpcla_fill_uniform_f64's generator */
{
    return (double)(splitmix64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1)) >> 11) * 0x1.0p-53;
}

/* ---- the two matrices, row by row, global columns ascending ------------------------------------------------ */
enum { ALLTOALL = 0, SLAB = 1 };
static const int64_t M = 1024;                           /* rows per rank: 4 SpMV row blocks, 16 SpMM row blocks */
static const int64_t NX = 256;                           /* slab: grid line length; M / NX = 4 lines per rank */

static int gen_row(int kind, int64_t g, int64_t *cols, double *vals)
{
    int n = 0;
    if (kind == ALLTOALL) {
        const int64_t own = g / M;
        for (int64_t q = 0; q < R; ++q) {
            int64_t c = q * M + (int64_t)((uint64_t)(g * g * 3 + g + q * 17) % (uint64_t)M);
            if (q == own) {                              /* plus the diagonal, in ascending order, no duplicate */
                const int64_t lo = c < g ? c : g, hi = c < g ? g : c;
                cols[n] = lo; vals[n++] = 4.0 + (double)(lo % 3);
                if (hi != lo) { cols[n] = hi; vals[n++] = 0.5 + (double)((hi + g) % 11) * 0.125; }
            } else {
                cols[n] = c; vals[n++] = 0.5 + (double)((c + g) % 11) * 0.125;
            }
        }
    } else {
        const int64_t ny = (M / NX) * R, i = g % NX, j = g / NX;
        if (j > 0) { cols[n] = g - NX; vals[n++] = -1.0; }
        if (i > 0) { cols[n] = g - 1; vals[n++] = -1.0; }
        cols[n] = g; vals[n++] = 4.0;
        if (i < NX - 1) { cols[n] = g + 1; vals[n++] = -1.0; }
        if (j < ny - 1) { cols[n] = g + NX; vals[n++] = -1.0; }
    }
    return n;
}

static int cmp_i64(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : x > y;
}

/* sorted distinct columns outside [lo, hi) that the rows of `rank` reference: its ghost columns */
static int64_t ghost_columns(int kind, int rank, int64_t **out)
{
    const int64_t lo = rank * M, hi = lo + M;
    int64_t *buf = (int64_t *)malloc((size_t)M * MAXROW * 8), n = 0, cols[MAXROW];
    double vals[MAXROW];
    for (int64_t g = lo; g < hi; ++g) {
        const int k = gen_row(kind, g, cols, vals);
        for (int t = 0; t < k; ++t)
            if (cols[t] < lo || cols[t] >= hi) buf[n++] = cols[t];
    }
    qsort(buf, (size_t)n, 8, cmp_i64);
    int64_t m = 0;
    for (int64_t t = 0; t < n; ++t)
        if (m == 0 || buf[m - 1] != buf[t]) buf[m++] = buf[t];
    *out = buf;
    return m;
}

#define CHECK(call)                                                                               \
    do {                                                                                          \
        int _s = (call);                                                                          \
        if (_s != 0) {                                                                            \
            fprintf(stderr, "rank %d: %s failed with status %d: %s\n", rank, #call, _s, hpcla_last_error()); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
#define HIPCHECK(call)                                                                            \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", rank, #call, hipGetErrorString(_e)); return 1; } \
    } while (0)

/* every rank reaches every barrier of a phase even when it has failed: a failed rank keeps going with `bad` set */
static int all_ok(int rank, int mine)
{
    g_ok[rank] = mine;
    barrier();
    int ok = 1;
    for (int r = 0; r < R; ++r) ok &= g_ok[r];
    barrier();
    return ok;
}

static int upload(int rank, void **dev, const void *host, size_t bytes, hipStream_t s)
{
    HIPCHECK(hipMalloc(dev, bytes ? bytes : 8));
    if (bytes) HIPCHECK(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, s));
    return 0;
}

static int block_lists(int rank, const int32_t *d_rp, const int32_t *d_split, int rpb, int32_t **d_int, int64_t *n_int,
                       int32_t **d_bnd, int64_t *n_bnd, hipStream_t s)
{
    const int64_t nblk = (M + rpb - 1) / rpb;
    int32_t *d_flags, *h_flags = (int32_t *)malloc((size_t)nblk * 4), *h_i = (int32_t *)malloc((size_t)nblk * 4),
            *h_b = (int32_t *)malloc((size_t)nblk * 4);
    HIPCHECK(hipMalloc((void **)&d_flags, (size_t)nblk * 4));
    CHECK(hpcla_classify_blocks_i32(d_rp, d_split, M, 0, M, rpb, d_flags, s));
    HIPCHECK(hipMemcpyAsync(h_flags, d_flags, (size_t)nblk * 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    *n_int = *n_bnd = 0;
    for (int64_t b = 0; b < nblk; ++b) { if (h_flags[b]) h_b[(*n_bnd)++] = (int32_t)b; else h_i[(*n_int)++] = (int32_t)b; }
    if (upload(rank, (void **)d_int, h_i, (size_t)*n_int * 4, s)) return 1;
    if (upload(rank, (void **)d_bnd, h_b, (size_t)*n_bnd * 4, s)) return 1;
    HIPCHECK(hipStreamSynchronize(s));
    HIPCHECK(hipFree(d_flags));
    free(h_flags); free(h_i); free(h_b);
    return 0;
}

/* export / all-gather / attach / probe of one plan; `width` values per index */
static int connect_plan(int rank, hpcla_halo_plan_t *plan, const int64_t *ghost, int64_t n_ghost, hipStream_t s, int phase0)
{
    STAGE(phase0 + 1);
    CHECK(hpcla_halo_plan_export(plan, g_desc + rank * HPCLA_WINDOW_DESC_BYTES, g_tab + (size_t)rank * HPCLA_WINDOW_TABLE_ROWS * R));
    STAGE(phase0 + 2);
    barrier();
    STAGE(phase0 + 3);
    int rc = hpcla_halo_plan_attach(plan, g_desc, g_tab);
    STAGE(phase0 + 4);
    if (rc) fprintf(stderr, "rank %d: attach: %s\n", rank, hpcla_last_error());
    if (!all_ok(rank, rc == 0)) return 1;
    STAGE(phase0 + 5);
    /* connection test over EVERY ghost slot: slot t holds local row (column - owner's first row) of its owner */
    int64_t *slots = (int64_t *)malloc((size_t)(n_ghost ? n_ghost : 1) * 8), *rows = (int64_t *)malloc((size_t)(n_ghost ? n_ghost : 1) * 8);
    for (int64_t t = 0; t < n_ghost; ++t) { slots[t] = t; rows[t] = ghost[t] % M; }
    int ok = 0;
    rc = hpcla_halo_plan_probe(plan, M, slots, rows, n_ghost, s, &ok);
    STAGE(phase0 + 6);
    if (rc || !ok) {
        fprintf(stderr, "rank %d: plan probe failed: %s\n", rank, hpcla_last_error());
        print_stages(rank, "probe failed");
    }
    free(slots); free(rows);
    return all_ok(rank, rc == 0 && ok) ? 0 : 1;
}

static int run_matrix(int rank, int kind, hpcla_comm_t *comm, hipStream_t s)
{
    const char *name = kind == ALLTOALL ? "alltoall" : "slab";
    g_kind_now[rank] = kind + 1;
    STAGE(1);
    const int64_t lo = rank * M, hi = lo + M, n = M * R;
    /* ---- local CSR in the split column space (own -> offset in x.v, ghost -> M + position in the ghost segment) ---- */
    int64_t *ghost = NULL;
    const int64_t n_ghost = ghost_columns(kind, rank, &ghost);
    int32_t *h_rp = (int32_t *)malloc((size_t)(M + 1) * 4), *h_split = (int32_t *)malloc((size_t)M * MAXROW * 4);
    int64_t *h_colg = (int64_t *)malloc((size_t)M * MAXROW * 8), cols[MAXROW];
    double *h_vals = (double *)malloc((size_t)M * MAXROW * 8), vals[MAXROW];
    int64_t nnz = 0;
    h_rp[0] = 0;
    for (int64_t g = lo; g < hi; ++g) {
        const int k = gen_row(kind, g, cols, vals);
        for (int t = 0; t < k; ++t) {
            const int64_t c = cols[t];
            int64_t sc;
            if (c >= lo && c < hi) sc = c - lo;
            else sc = M + ((int64_t *)bsearch(&c, ghost, (size_t)n_ghost, 8, cmp_i64) - ghost);
            h_colg[nnz] = c; h_split[nnz] = (int32_t)sc; h_vals[nnz++] = vals[t];
        }
        h_rp[g - lo + 1] = (int32_t)nnz;
    }
    /* recv lists: ghost columns by owner (ascending rank = ascending column); send lists: what rank q's ghosts name of mine */
    int32_t recv_ranks[MAXR], send_ranks[MAXR];
    int64_t recv_counts[MAXR], send_counts[MAXR];
    int n_recv = 0, n_send = 0;
    for (int q = 0; q < R; ++q) {
        int64_t cnt = 0;
        for (int64_t t = 0; t < n_ghost; ++t) cnt += ghost[t] / M == q;
        if (cnt) { recv_ranks[n_recv] = q; recv_counts[n_recv++] = cnt; }
    }
    int32_t *h_send = (int32_t *)malloc((size_t)M * R * 4);
    int64_t n_send_total = 0;
    for (int q = 0; q < R; ++q) {
        if (q == rank) continue;
        int64_t *gq = NULL, cnt = 0;
        const int64_t nq = ghost_columns(kind, q, &gq);
        for (int64_t t = 0; t < nq; ++t)
            if (gq[t] >= lo && gq[t] < hi) { h_send[n_send_total + cnt] = (int32_t)(gq[t] - lo); ++cnt; }
        free(gq);
        if (cnt) { send_ranks[n_send] = q; send_counts[n_send++] = cnt; n_send_total += cnt; }
    }
    const int want_nb = kind == ALLTOALL ? R - 1 : (rank > 0) + (rank < R - 1);
    if (n_recv != want_nb || n_send != want_nb) {
        fprintf(stderr, "rank %d %s: %d recv / %d send neighbours, expected %d\n", rank, name, n_recv, n_send, want_nb);
        return 1;
    }
    int32_t *d_rp, *d_split, *d_send;
    double *d_vals;
    if (upload(rank, (void **)&d_rp, h_rp, (size_t)(M + 1) * 4, s) || upload(rank, (void **)&d_split, h_split, (size_t)nnz * 4, s) ||
        upload(rank, (void **)&d_vals, h_vals, (size_t)nnz * 8, s) || upload(rank, (void **)&d_send, h_send, (size_t)n_send_total * 4, s))
        return 1;
    HIPCHECK(hipStreamSynchronize(s));

    /* ---- vector plan (double-buffered window, fused SpMV) -------------------------------------------------------- */
    hpcla_halo_plan_t *plan = NULL, *plan16 = NULL;
    STAGE(90);
    int rc = hpcla_halo_plan_create(&plan, comm, n_send, send_ranks, send_counts, d_send, 0, n_recv, recv_ranks, recv_counts, 1);
    if (rc) fprintf(stderr, "rank %d %s: plan create: %s\n", rank, name, hpcla_last_error());
    if (!all_ok(rank, rc == 0)) return 1;
    if (connect_plan(rank, plan, ghost, n_ghost, s, 100)) return 1;
    int32_t *d_int, *d_bnd;
    int64_t n_int, n_bnd;
    if (block_lists(rank, d_rp, d_split, hpcla_spmv_rows_per_block(), &d_int, &n_int, &d_bnd, &n_bnd, s)) return 1;

    double *d_x, *d_y, *d_out, *d_work, *h_y = (double *)malloc((size_t)M * 8), *xg = (double *)malloc((size_t)n * 8);
    HIPCHECK(hipMalloc((void **)&d_x, (size_t)M * 8));
    HIPCHECK(hipMalloc((void **)&d_y, (size_t)M * 8));
    HIPCHECK(hipMalloc((void **)&d_out, 8));
    HIPCHECK(hipMalloc((void **)&d_work, (size_t)hpcla_reduce_work_bytes()));
    int bad = 0;
    barrier();
    for (int step = 0; step < 6; ++step) {
        STAGE(200 + step);
        const uint64_t seed = 0xC0FFEEULL + 977ULL * (uint64_t)step + 31ULL * (uint64_t)kind;
        CHECK(hpcla_fill_uniform_f64(d_x, lo, M, seed, s));
        CHECK(hpcla_spmv_dist_f64_i32(plan, d_rp, d_split, d_vals, d_x, M, d_y, M, nnz, 0, d_int, n_int, d_bnd, n_bnd, s));
        CHECK(hpcla_dot_f64(comm, d_x, d_y, M, d_out, d_work, s));
        /* read-backs BEHIND the synchronisation: nothing of the host's is parked behind the waiting kernels (the rank threads
         * of this process share the runtime's locks; the stall this test showed in round 6 was a device-wide wait inside the
         * library's probe, comm.hip hpcla_halo_plan_probe) */
        HIPCHECK(hipStreamSynchronize(s));
        HIPCHECK(hipMemcpy(h_y, d_y, (size_t)M * 8, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy((void *)&g_dots[rank][step], d_out, 8, hipMemcpyDeviceToHost));
        for (int64_t g = 0; g < n; ++g) xg[g] = u01(seed, g);
        for (int64_t r = 0; r < M; ++r) {                 /* stored order, multiply then add: the kernel's order */
            double acc = 0.0;
            for (int32_t j = h_rp[r]; j < h_rp[r + 1]; ++j) acc += h_vals[j] * xg[h_colg[j]];
            if (memcmp(&acc, &h_y[r], 8) != 0 && bad++ < 5)
                fprintf(stderr, "rank %d %s step %d row %lld: got %.17g want %.17g\n", rank, name, step, (long long)(lo + r), h_y[r], acc);
        }
    }
    barrier();
    for (int step = 0; step < 6; ++step)
        for (int r = 0; r < R; ++r)
            if (memcmp(&g_dots[rank][step], &g_dots[r][step], 8) != 0 && bad++ < 5)
                fprintf(stderr, "rank %d %s: dot of step %d differs from rank %d's\n", rank, name, step, r);
    /* the dot itself: every rank can afford the global vectors at this size (tolerance 1e-12 relative, tree vs sequential) */
    {
        const uint64_t seed = 0xC0FFEEULL + 977ULL * 5 + 31ULL * (uint64_t)kind;
        long double ref = 0.0L, mag = 0.0L;
        for (int64_t g = 0; g < n; ++g) {
            const int k = gen_row(kind, g, cols, vals);
            double acc = 0.0;
            for (int t = 0; t < k; ++t) acc += vals[t] * u01(seed, cols[t]);
            ref += (long double)u01(seed, g) * (long double)acc;
            mag += (long double)u01(seed, g) * (long double)(acc < 0 ? -acc : acc);
        }
        const long double err = (long double)g_dots[rank][5] - ref;
        if (!((err < 0 ? -err : err) <= 1e-12L * mag)) { fprintf(stderr, "rank %d %s: dot %.17g, reference %.17Lg\n", rank, name, g_dots[rank][5], ref); ++bad; }
    }

    /* ---- A*B, k = 16: ghost ROWS through a width-16 single-buffered plan ---------------------------------------- */
    const int K = 16;
    STAGE(280);
    rc = hpcla_halo_plan_create_ex(&plan16, comm, n_send, send_ranks, send_counts, d_send, 0, n_recv, recv_ranks, recv_counts, K,
                                   HPCLA_HALO_SINGLE_BUFFER);
    if (rc) fprintf(stderr, "rank %d %s: width-16 plan create: %s\n", rank, name, hpcla_last_error());
    if (!all_ok(rank, rc == 0)) return 1;
    STAGE(290);
    if (connect_plan(rank, plan16, ghost, n_ghost, s, 300)) return 1;
    int32_t *d_int16, *d_bnd16;
    int64_t n_int16, n_bnd16;
    if (block_lists(rank, d_rp, d_split, hpcla_spmm_rows_per_block(), &d_int16, &n_int16, &d_bnd16, &n_bnd16, s)) return 1;
    double *d_B, *d_C, *d_ghost = NULL, *h_C = (double *)malloc((size_t)M * K * 8);
    int64_t ng = 0;
    HIPCHECK(hipMalloc((void **)&d_B, (size_t)M * K * 8));
    HIPCHECK(hipMalloc((void **)&d_C, (size_t)M * K * 8));
    CHECK(hpcla_halo_ghost_ptr(plan16, &d_ghost, &ng));
    if (ng != n_ghost) { fprintf(stderr, "rank %d %s: ghost of %lld rows, expected %lld\n", rank, name, (long long)ng, (long long)n_ghost); ++bad; }
    barrier();
    for (int rep = 0; rep < 3; ++rep) {                  /* three products, other B each time: a stale ghost row cannot pass */
        STAGE(400 + rep);
        const uint64_t seed = 4711ULL + 13ULL * (uint64_t)rep;
        CHECK(hpcla_fill_uniform_f64(d_B, lo * K, M * K, seed, s));      /* B[g, c] = u01(seed, g*K + c), row-major */
        CHECK(hpcla_halo_begin(plan16, d_B, s));
        if (n_int16)
            CHECK(hpcla_spmm_split_f64_i32(d_rp, d_split, d_vals, d_B, K, d_ghost, K, M, d_C, K, M, nnz, K, 0, d_int16, n_int16, s));
        CHECK(hpcla_halo_end(plan16, s));
        if (n_bnd16)
            CHECK(hpcla_spmm_split_f64_i32(d_rp, d_split, d_vals, d_B, K, d_ghost, K, M, d_C, K, M, nnz, K, 0, d_bnd16, n_bnd16, s));
        HIPCHECK(hipStreamSynchronize(s));
        HIPCHECK(hipMemcpy(h_C, d_C, (size_t)M * K * 8, hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < M; ++r)
            for (int c = 0; c < K; ++c) {
                double acc = 0.0;
                for (int32_t j = h_rp[r]; j < h_rp[r + 1]; ++j) acc += h_vals[j] * u01(seed, h_colg[j] * K + c);
                if (memcmp(&acc, &h_C[r * K + c], 8) != 0 && bad++ < 5)
                    fprintf(stderr, "rank %d %s product %d C(%lld,%d): got %.17g want %.17g\n", rank, name, rep, (long long)(lo + r), c, h_C[r * K + c], acc);
            }
    }
    int t1 = 0, t2 = 0, t3 = 0;
    CHECK(hpcla_halo_status(plan, &t1));
    CHECK(hpcla_halo_status(plan16, &t2));
    CHECK(hpcla_comm_status(comm, &t3));
    if (t1 || t2 || t3) { fprintf(stderr, "rank %d %s: a spin timed out (%d %d %d)\n", rank, name, t1, t2, t3); ++bad; }
    HIPCHECK(hipStreamSynchronize(s));
    STAGE(500);
    const int ok = all_ok(rank, bad == 0);                /* everybody done before anybody unmaps */
    STAGE(510);
    CHECK(hpcla_halo_plan_destroy(plan16));
    STAGE(520);
    CHECK(hpcla_halo_plan_destroy(plan));
    STAGE(530);
    if (ok && rank == 0)
        printf("%s: %d ranks, %d neighbours each way on rank 0, %lld ghost values: 6 SpMVs + 3 products (k = 16) bit-exact, dots identical on all ranks\n",
               name, R, n_recv, (long long)n_ghost);
    hipFree(d_rp); hipFree(d_split); hipFree(d_vals); hipFree(d_send); hipFree(d_int); hipFree(d_bnd); hipFree(d_int16); hipFree(d_bnd16);
    hipFree(d_x); hipFree(d_y); hipFree(d_out); hipFree(d_work); hipFree(d_B); hipFree(d_C);
    free(ghost); free(h_rp); free(h_split); free(h_colg); free(h_vals); free(h_send); free(h_y); free(xg); free(h_C);
    return ok ? 0 : 1;
}

static int run_rank(int rank)
{
    hipStream_t s;
    HIPCHECK(hipSetDevice(0));
    HIPCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hpcla_comm_t *comm = NULL;
    int rc = hpcla_comm_init_rank_ex(&comm, NULL, R, rank, HPCLA_COMM_NO_RCCL);
    if (rc == 0) rc = hpcla_comm_window_export(comm, g_desc + rank * HPCLA_WINDOW_DESC_BYTES);
    if (rc) fprintf(stderr, "rank %d: communicator: %s\n", rank, hpcla_last_error());
    if (!all_ok(rank, rc == 0)) return 1;
    rc = hpcla_comm_window_attach(comm, g_desc);
    if (rc) fprintf(stderr, "rank %d: window attach: %s\n", rank, hpcla_last_error());
    if (!all_ok(rank, rc == 0)) return 1;
    int ok = 0;
    rc = hpcla_comm_window_selftest(comm, 20.0, &ok);
    if (rc || !ok) fprintf(stderr, "rank %d: window connection test failed (%s)\n", rank, hpcla_last_error());
    if (!all_ok(rank, rc == 0 && ok)) return 1;
    int bad = 0;
    for (int kind = 0; kind < 2 && !bad; ++kind) {
        bad = run_matrix(rank, kind, comm, s);
        if (!all_ok(rank, !bad)) bad = 1;
    }
    CHECK(hpcla_comm_destroy(comm));
    HIPCHECK(hipStreamDestroy(s));
    return bad;
}

static void *thread_main(void *arg)
{
    const int rank = (int)(intptr_t)arg;
    const int rc = run_rank(rank);
    if (rc) {
        __atomic_store_n(&g_fail, 1, __ATOMIC_SEQ_CST);
        /* a rank that left early would leave the others in a barrier: no collective cleanup, end the process */
        fprintf(stderr, "rank %d failed -- exiting\n", rank);
        fflush(stderr);
        _Exit(1);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc > 1) R = atoi(argv[1]);
    if (R < 2 || R > MAXR) { fprintf(stderr, "usage: %s [ranks 2..%d]\n", argv[0], MAXR); return 2; }
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    setenv("HPCLA_PUSH_TIMEOUT_S", "20", 0);
    /* every rank's streams need hardware queues of their own: a waiting kernel must never sit in front of the kernel it
     * waits for (the runtime's default is 4 queues per process, shared round-robin) */
    setenv("GPU_MAX_HW_QUEUES", "32", 0);
    pthread_barrier_init(&g_bar, NULL, (unsigned)R);
    pthread_t th[MAXR];
    for (int r = 0; r < R; ++r) pthread_create(&th[r], NULL, thread_main, (void *)(intptr_t)r);
    for (int r = 0; r < R; ++r) pthread_join(th[r], NULL);
    if (!g_fail) printf("C-ABI %d ranks in one process PASS\n", R);
    return g_fail;
}
