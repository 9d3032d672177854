/* cabi_window_pair.c -- TWO processes, plain C, no Python, no torch, no MPI: the peer-window transport of
 * libhpcla_rocm.so bootstrapped and driven through include/hpcla_rocm.h alone, the way a Julia extension would
 * with MPI.Allgather in place of the socket pair used here.
 *
 * The parent forks BEFORE anything touches the GPU; the two ranks (both on device 0 -- the windows allow ranks
 * to share a GPU, RCCL does not) then
 *   1. create a window-only communicator (HPCLA_COMM_NO_RCCL), export / exchange / attach its window, run
 *      the connection test;
 *   2. each own one slab of a 2-D 5-point Poisson matrix (generated on the device), build the split column
 *      space and the halo lists of a two-slab partition by hand, create the halo plan, export / exchange /
 *      attach its window;
 *   3. run y = A*x several times through hpcla_spmv_dist_f64_i32 (fused push + in-kernel wait), with x
 *      changing between steps, and compare every row with a scalar CPU loop over the closed-form stencil:
 *      bit-exact;
 *   4. compute dot(x, y) through hpcla_dot_f64 (window all-reduce): both ranks must hold the same bits.
 * Exit code 0 = both ranks passed.  Build/run: tests/test_cabi_from_c.py. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include "hpcla_rocm.h"

static int g_rank = 0;
#define CHECK(call)                                                                               \
    do {                                                                                          \
        int _s = (call);                                                                          \
        if (_s != 0) {                                                                            \
            fprintf(stderr, "rank %d: %s failed with status %d: %s\n", g_rank, #call, _s, hpcla_last_error()); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
#define HIPCHECK(call)                                                                            \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #call, hipGetErrorString(_e)); return 1; } \
    } while (0)

static int xfer(int fd, void *mine, void *theirs, size_t n)      /* the "all-gather" of a two-rank job */
{
    if (write(fd, mine, n) != (ssize_t)n) return 1;
    size_t got = 0;
    while (got < n) {
        ssize_t r = read(fd, (char *)theirs + got, n - got);
        if (r <= 0) return 1;
        got += (size_t)r;
    }
    return 0;
}

static uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double u01(uint64_t seed, int64_t i)                    /* hpcla_fill_uniform_f64's generator */
{
    return (double)(splitmix64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1)) >> 11) * 0x1.0p-53;
}

static int run_rank(int rank, int fd)
{
    g_rank = rank;
    const int peer = 1 - rank;
    const int64_t nx = 512, ny = 96, n = nx * ny, lo = rank * (n / 2), hi = lo + n / 2, n_own = hi - lo;
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    setenv("HPCLA_PUSH_TIMEOUT_S", "30", 0);
    HIPCHECK(hipSetDevice(0));

    /* ---- 1. communicator + its window ------------------------------------------------------------- */
    hpcla_comm_t *comm = NULL;
    CHECK(hpcla_comm_init_rank_ex(&comm, NULL, 2, rank, HPCLA_COMM_NO_RCCL));
    uint8_t descs[2 * HPCLA_WINDOW_DESC_BYTES];
    CHECK(hpcla_comm_window_export(comm, descs + rank * HPCLA_WINDOW_DESC_BYTES));
    if (xfer(fd, descs + rank * HPCLA_WINDOW_DESC_BYTES, descs + peer * HPCLA_WINDOW_DESC_BYTES, HPCLA_WINDOW_DESC_BYTES)) return 1;
    CHECK(hpcla_comm_window_attach(comm, descs));
    int ok = 0;
    CHECK(hpcla_comm_window_selftest(comm, 10.0, &ok));
    if (!ok) { fprintf(stderr, "rank %d: window connection test failed\n", rank); return 1; }

    /* ---- 2. this rank's slab, split column space, halo plan ------------------------------------------- */
    const int64_t nnz = hpcla_poisson2d_nnz(nx, ny, lo, hi);
    int64_t *d_rp64, *d_colg;
    double *d_vals;
    HIPCHECK(hipMalloc((void **)&d_rp64, (n_own + 1) * 8));
    HIPCHECK(hipMalloc((void **)&d_colg, nnz * 8));
    HIPCHECK(hipMalloc((void **)&d_vals, nnz * 8));
    CHECK(hpcla_gen_poisson2d(nx, ny, lo, hi, d_rp64, d_colg, d_vals, NULL));
    int64_t *h_rp64 = (int64_t *)malloc((n_own + 1) * 8), *h_colg = (int64_t *)malloc(nnz * 8);
    HIPCHECK(hipMemcpy(h_rp64, d_rp64, (n_own + 1) * 8, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(h_colg, d_colg, nnz * 8, hipMemcpyDeviceToHost));
    int32_t *h_rp = (int32_t *)malloc((n_own + 1) * 4), *h_split = (int32_t *)malloc(nnz * 4);
    for (int64_t i = 0; i <= n_own; ++i) h_rp[i] = (int32_t)h_rp64[i];
    /* split columns: own -> offset in x.v; the one ghost line -> n_own + position (ascending global column) */
    const int64_t ghost_lo = rank == 0 ? hi : lo - nx;
    for (int64_t j = 0; j < nnz; ++j) {
        const int64_t c = h_colg[j];
        h_split[j] = (int32_t)((c >= lo && c < hi) ? c - lo : n_own + (c - ghost_lo));
    }
    int32_t *d_rp, *d_split;
    HIPCHECK(hipMalloc((void **)&d_rp, (n_own + 1) * 4));
    HIPCHECK(hipMalloc((void **)&d_split, nnz * 4));
    HIPCHECK(hipMemcpy(d_rp, h_rp, (n_own + 1) * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_split, h_split, nnz * 4, hipMemcpyHostToDevice));
    /* what the neighbour needs from me: my last grid line (rank 0) / my first grid line (rank 1) */
    int32_t *h_send = (int32_t *)malloc(nx * 4), *d_send;
    for (int64_t i = 0; i < nx; ++i) h_send[i] = (int32_t)(rank == 0 ? n_own - nx + i : i);
    HIPCHECK(hipMalloc((void **)&d_send, nx * 4));
    HIPCHECK(hipMemcpy(d_send, h_send, nx * 4, hipMemcpyHostToDevice));
    const int32_t nb_rank[1] = {peer};
    const int64_t nb_count[1] = {nx};
    hpcla_halo_plan_t *plan = NULL;
    HIPCHECK(hipDeviceSynchronize());
    CHECK(hpcla_halo_plan_create(&plan, comm, 1, nb_rank, nb_count, d_send, 0, 1, nb_rank, nb_count, 1));
    uint8_t pdesc[2 * HPCLA_WINDOW_DESC_BYTES];
    int64_t ptab[2 * HPCLA_WINDOW_TABLE_ROWS * 2];
    CHECK(hpcla_halo_plan_export(plan, pdesc + rank * HPCLA_WINDOW_DESC_BYTES, ptab + rank * HPCLA_WINDOW_TABLE_ROWS * 2));
    if (xfer(fd, pdesc + rank * HPCLA_WINDOW_DESC_BYTES, pdesc + peer * HPCLA_WINDOW_DESC_BYTES, HPCLA_WINDOW_DESC_BYTES)) return 1;
    if (xfer(fd, ptab + rank * HPCLA_WINDOW_TABLE_ROWS * 2, ptab + peer * HPCLA_WINDOW_TABLE_ROWS * 2,
             sizeof(int64_t) * HPCLA_WINDOW_TABLE_ROWS * 2)) return 1;
    CHECK(hpcla_halo_plan_attach(plan, pdesc, ptab));
    /* the plan's connection test: my ghost slot i must hold the peer's row h (its last grid line for rank 1's
     * lower ghosts, its first grid line for rank 0's upper ghosts); both ranks run the same two exchanges */
    {
        int64_t *slots = (int64_t *)malloc(nx * 8), *prow = (int64_t *)malloc(nx * 8);
        for (int64_t i = 0; i < nx; ++i) { slots[i] = i; prow[i] = rank == 0 ? i : n_own - nx + i; }
        int probe_ok = 0, peer_ok = 0;
        CHECK(hpcla_halo_plan_probe(plan, n_own, slots, prow, nx, NULL, &probe_ok));
        if (xfer(fd, &probe_ok, &peer_ok, sizeof(int))) return 1;
        if (!probe_ok || !peer_ok) { fprintf(stderr, "rank %d: plan probe failed: %s\n", rank, hpcla_last_error()); return 1; }
        /* a wrong expectation must be caught (and costs two more exchanges on BOTH ranks) */
        prow[nx / 2] += 1;
        CHECK(hpcla_halo_plan_probe(plan, n_own, slots, prow, nx, NULL, &probe_ok));
        if (probe_ok) { fprintf(stderr, "rank %d: plan probe accepted a wrong row\n", rank); return 1; }
        free(slots); free(prow);
    }
    /* interior / boundary row blocks */
    const int rpb = hpcla_spmv_rows_per_block();
    const int64_t nblk = (n_own + rpb - 1) / rpb;
    int32_t *d_flags, *h_flags = (int32_t *)malloc(nblk * 4);
    HIPCHECK(hipMalloc((void **)&d_flags, nblk * 4));
    CHECK(hpcla_classify_blocks_i32(d_rp, d_split, n_own, 0, n_own, rpb, d_flags, NULL));
    HIPCHECK(hipMemcpy(h_flags, d_flags, nblk * 4, hipMemcpyDeviceToHost));
    int32_t *h_int = (int32_t *)malloc(nblk * 4), *h_bnd = (int32_t *)malloc(nblk * 4), *d_int, *d_bnd;
    int64_t n_int = 0, n_bnd = 0;
    for (int64_t b = 0; b < nblk; ++b) { if (h_flags[b]) h_bnd[n_bnd++] = (int32_t)b; else h_int[n_int++] = (int32_t)b; }
    HIPCHECK(hipMalloc((void **)&d_int, (n_int ? n_int : 1) * 4));
    HIPCHECK(hipMalloc((void **)&d_bnd, (n_bnd ? n_bnd : 1) * 4));
    HIPCHECK(hipMemcpy(d_int, h_int, n_int * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_bnd, h_bnd, n_bnd * 4, hipMemcpyHostToDevice));
    if (n_bnd != nx / rpb) { fprintf(stderr, "rank %d: %lld boundary blocks, expected %lld\n", rank, (long long)n_bnd, (long long)(nx / rpb)); return 1; }

    /* ---- 3. distributed SpMV, x changing from step to step, every row against the closed form --------------- */
    double *d_x, *d_y, *h_y = (double *)malloc(n_own * 8), *d_out, *d_work;
    HIPCHECK(hipMalloc((void **)&d_x, n_own * 8));
    HIPCHECK(hipMalloc((void **)&d_y, n_own * 8));
    HIPCHECK(hipMalloc((void **)&d_out, 8));
    HIPCHECK(hipMalloc((void **)&d_work, hpcla_reduce_work_bytes()));
    int bad = 0;
    double dots[6];
    for (int step = 0; step < 6; ++step) {
        const uint64_t seed = 0xC0FFEEULL + 977ULL * (uint64_t)step;
        CHECK(hpcla_fill_uniform_f64(d_x, lo, n_own, seed, NULL));
        CHECK(hpcla_spmv_dist_f64_i32(plan, d_rp, d_split, d_vals, d_x, n_own, d_y, n_own, nnz, 0, d_int, n_int, d_bnd,
                                      n_bnd, NULL));
        CHECK(hpcla_dot_f64(comm, d_x, d_y, n_own, d_out, d_work, NULL));
        HIPCHECK(hipMemcpy(h_y, d_y, n_own * 8, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(&dots[step], d_out, 8, hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < n_own; ++r) {           /* ascending column order, multiply then add: the kernel's order */
            const int64_t g = lo + r, i = g % nx, j = g / nx;
            double acc = 0.0;
            if (j > 0) acc += -1.0 * u01(seed, g - nx);
            if (i > 0) acc += -1.0 * u01(seed, g - 1);
            acc += 4.0 * u01(seed, g);
            if (i < nx - 1) acc += -1.0 * u01(seed, g + 1);
            if (j < ny - 1) acc += -1.0 * u01(seed, g + nx);
            if (acc != h_y[r] && bad++ < 5)
                fprintf(stderr, "rank %d step %d row %lld: got %.17g want %.17g\n", rank, step, (long long)g, h_y[r], acc);
        }
    }
    /* ---- 4. both ranks hold the same dot bits --------------------------------------------------------------- */
    double theirs[6];
    if (xfer(fd, dots, theirs, sizeof(dots))) return 1;
    for (int step = 0; step < 6; ++step)
        if (memcmp(&dots[step], &theirs[step], 8) != 0) { fprintf(stderr, "rank %d: dot of step %d differs between the ranks\n", rank, step); ++bad; }
    int t1 = 0, t2 = 0;
    CHECK(hpcla_halo_status(plan, &t1));
    CHECK(hpcla_comm_status(comm, &t2));
    if (t1 || t2) { fprintf(stderr, "rank %d: a spin timed out (%d %d)\n", rank, t1, t2); ++bad; }
    HIPCHECK(hipDeviceSynchronize());
    char c = 1, d = 0;
    if (xfer(fd, &c, &d, 1)) return 1;                    /* both done before either unmaps */
    CHECK(hpcla_halo_plan_destroy(plan));
    CHECK(hpcla_comm_destroy(comm));
    if (!bad) printf("rank %d: 6 distributed SpMVs bit-exact, dot %.15g identical on both ranks\n", rank, dots[5]);
    return bad ? 1 : 0;
}

int main(void)
{
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) { perror("socketpair"); return 1; }
    const pid_t pid = fork();                              /* before ANY HIP call */
    if (pid < 0) { perror("fork"); return 1; }
    if (pid == 0) {
        close(sv[0]);
        _exit(run_rank(1, sv[1]));
    }
    close(sv[1]);
    const int rc0 = run_rank(0, sv[0]);
    int st = 0;
    waitpid(pid, &st, 0);
    const int rc1 = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
    if (rc0 == 0 && rc1 == 0) printf("C-ABI window pair PASS\n");
    return rc0 || rc1;
}
