/* cabi_smoke.c -- a host with NO Python, NO torch drives libhpcla_rocm.so through include/hpcla_rocm.h
 * only (plain C + the HIP runtime for memory), the way the Julia extension's @ccall stubs do:
 * generate the 2-D Poisson rows on the device, build the compressed column space on the device,
 * run y = A*x through the plain CSR entry point AND the distributed entry point (serial communicator,
 * no neighbours), a dot product, a fused CG update and 12 CG iterations enqueued by ONE call, and compare with
 * scalar CPU loops.
 * Build/run: tests/test_cabi_from_c.py (gcc + libamdhip64 for hipMalloc/hipMemcpy only). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hpcla_rocm.h"

#define CHECK(call)                                                                               \
    do {                                                                                          \
        int _s = (call);                                                                          \
        if (_s != 0) {                                                                            \
            fprintf(stderr, "%s failed with status %d: %s\n", #call, _s, hpcla_last_error());     \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
#define HIPCHECK(call)                                                                            \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(_e)); return 1; } \
    } while (0)

/* `cabi_smoke N` : time y = A*x on the N x N 5-point matrix through the C ABI alone (HIP events), the
 * headline workload without any Python in the process. */
static int bench(int64_t N)
{
    const int64_t n = N * N, nnz = hpcla_poisson2d_nnz(N, N, 0, n);
    int64_t *rp64, *colg, *col_indices;
    int32_t *rowptr, *colval;
    double *vals, *x, *y;
    void *work;
    HIPCHECK(hipMalloc((void **)&rp64, (n + 1) * 8));
    HIPCHECK(hipMalloc((void **)&colg, nnz * 8));
    HIPCHECK(hipMalloc((void **)&vals, nnz * 8));
    CHECK(hpcla_gen_poisson2d(N, N, 0, n, rp64, colg, vals, NULL));
    HIPCHECK(hipMalloc(&work, hpcla_colspace_work_bytes(n)));
    HIPCHECK(hipMalloc((void **)&colval, nnz * 4));
    HIPCHECK(hipMalloc((void **)&col_indices, n * 8));
    int64_t ncomp = 0;
    CHECK(hpcla_compress_columns_i32(colg, nnz, 0, n, colval, 0, col_indices, &ncomp, work, NULL));
    int64_t *h_rp = (int64_t *)malloc((n + 1) * 8);
    int32_t *h_rp32 = (int32_t *)malloc((n + 1) * 4);
    HIPCHECK(hipMemcpy(h_rp, rp64, (n + 1) * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i <= n; ++i) h_rp32[i] = (int32_t)h_rp[i];
    HIPCHECK(hipMalloc((void **)&rowptr, (n + 1) * 4));
    HIPCHECK(hipMemcpy(rowptr, h_rp32, (n + 1) * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipFree(rp64)); HIPCHECK(hipFree(colg)); HIPCHECK(hipFree(work)); HIPCHECK(hipFree(col_indices));
    HIPCHECK(hipMalloc((void **)&x, n * 8));
    HIPCHECK(hipMalloc((void **)&y, n * 8));
    CHECK(hpcla_fill_uniform_f64(x, 0, n, 0xC0FFEEULL, NULL));
    /* plan time: the order in which the launches walk the row blocks, measured once for this matrix (y as scratch) */
    int group = 0;
    CHECK(hpcla_spmv_tune_block_order_f64_i32(rowptr, colval, vals, x, NULL, n, y, n, nnz, 0, NULL, &group));
    printf("C-ABI bench: block order chosen by measurement: %s%d\n", group > 1 ? "XCD groups of " : "natural, group = ", group);
    for (int i = 0; i < 20; ++i) CHECK(hpcla_spmv_csr_f64_i32(rowptr, colval, vals, x, y, n, nnz, 0, NULL));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    const int steps = 200;
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0, NULL));
    for (int i = 0; i < steps; ++i) CHECK(hpcla_spmv_csr_f64_i32(rowptr, colval, vals, x, y, n, nnz, 0, NULL));
    HIPCHECK(hipEventRecord(e1, NULL));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms / steps * 1e-3;
    const double b_alg = 12.0 * nnz + 4.0 * (n + 1) + 8.0 * n + 8.0 * n;
    printf("C-ABI bench: N=%lld n=%lld nnz=%lld  %.4f ms/SpMV  %.1f GFLOP/s  %.1f GB/s algorithmic = %.3f of 8 TB/s\n",
           (long long)N, (long long)n, (long long)nnz, t * 1e3, 2.0 * nnz / t / 1e9, b_alg / t / 1e9,
           b_alg / t / 1e9 / 8000.0);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1) {
        CHECK(hpcla_set_device(0));
        return bench(atoll(argv[1]));
    }
    const int64_t nx = 300, ny = 200, n = nx * ny;
    int ndev = 0;
    CHECK(hpcla_device_count(&ndev));
    CHECK(hpcla_set_device(0));
    char arch[64];
    int cus = 0;
    CHECK(hpcla_device_info(0, &cus, arch, sizeof(arch)));
    printf("device 0: %s, %d CUs, library version %d\n", arch, cus, hpcla_version());

    /* matrix on the device */
    const int64_t nnz = hpcla_poisson2d_nnz(nx, ny, 0, n);
    int64_t *rp64, *colg, *col_indices;
    int32_t *rowptr, *colval;
    double *vals, *x, *y, *y2, *z, *scal;
    void *work;
    HIPCHECK(hipMalloc((void **)&rp64, (n + 1) * 8));
    HIPCHECK(hipMalloc((void **)&colg, nnz * 8));
    HIPCHECK(hipMalloc((void **)&vals, nnz * 8));
    CHECK(hpcla_gen_poisson2d(nx, ny, 0, n, rp64, colg, vals, NULL));
    HIPCHECK(hipMalloc(&work, hpcla_colspace_work_bytes(n)));
    HIPCHECK(hipMalloc((void **)&colval, nnz * 4));
    HIPCHECK(hipMalloc((void **)&col_indices, n * 8));
    int64_t ncomp = 0;
    CHECK(hpcla_compress_columns_i32(colg, nnz, 0, n, colval, 0, col_indices, &ncomp, work, NULL));
    if (ncomp != n) { fprintf(stderr, "ncols_compressed %lld != %lld\n", (long long)ncomp, (long long)n); return 1; }

    /* host copies for the reference loop; rowptr as int32 */
    int64_t *h_rp = (int64_t *)malloc((n + 1) * 8), *h_col = (int64_t *)malloc(nnz * 8);
    double *h_val = (double *)malloc(nnz * 8), *h_x = (double *)malloc(n * 8), *h_y = (double *)malloc(n * 8);
    int32_t *h_rp32 = (int32_t *)malloc((n + 1) * 4);
    HIPCHECK(hipMemcpy(h_rp, rp64, (n + 1) * 8, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(h_col, colg, nnz * 8, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(h_val, vals, nnz * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i <= n; ++i) h_rp32[i] = (int32_t)h_rp[i];
    HIPCHECK(hipMalloc((void **)&rowptr, (n + 1) * 4));
    HIPCHECK(hipMemcpy(rowptr, h_rp32, (n + 1) * 4, hipMemcpyHostToDevice));

    HIPCHECK(hipMalloc((void **)&x, n * 8));
    HIPCHECK(hipMalloc((void **)&y, n * 8));
    HIPCHECK(hipMalloc((void **)&y2, n * 8));
    HIPCHECK(hipMalloc((void **)&z, n * 8));
    HIPCHECK(hipMemset(z, 0, n * 8));
    HIPCHECK(hipMalloc((void **)&scal, 4 * 8));
    CHECK(hpcla_fill_uniform_f64(x, 0, n, 0xC0FFEEULL, NULL));
    HIPCHECK(hipMemcpy(h_x, x, n * 8, hipMemcpyDeviceToHost));

    /* (1) plain CSR entry point */
    CHECK(hpcla_spmv_csr_f64_i32(rowptr, colval, vals, x, y, n, nnz, 0, NULL));
    /* (2) distributed entry point, serial communicator, no halo plan */
    hpcla_comm_t *comm = NULL;
    CHECK(hpcla_comm_init_rank(&comm, NULL, 1, 0));
    CHECK(hpcla_spmv_dist_f64_i32(NULL, rowptr, colval, vals, x, n, y2, n, nnz, 0, NULL, 0, NULL, 0, NULL));
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(h_y, y, n * 8, hipMemcpyDeviceToHost));
    double *h_y2 = (double *)malloc(n * 8);
    HIPCHECK(hipMemcpy(h_y2, y2, n * 8, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    double dot_ref = 0.0;
    for (int64_t r = 0; r < n; ++r) {
        double acc = 0.0;                                  /* the reference loop, src/sparse.jl:2055-2066 */
        for (int64_t j = h_rp[r]; j < h_rp[r + 1]; ++j) acc += h_val[j] * h_x[h_col[j]];
        if (memcmp(&acc, &h_y[r], 8) != 0 || memcmp(&acc, &h_y2[r], 8) != 0) ++bad;
        dot_ref += h_x[r] * acc;
    }
    if (bad) { fprintf(stderr, "SpMV mismatch in %lld rows\n", (long long)bad); return 1; }

    /* (3) dot with the device-resident scalar, (4) fused CG update */
    void *rwork;
    HIPCHECK(hipMalloc(&rwork, hpcla_reduce_work_bytes()));
    CHECK(hpcla_dot_f64(comm, x, y, n, scal, rwork, NULL));
    double h_s[2];
    HIPCHECK(hipMemcpy(h_s, scal, 8, hipMemcpyDeviceToHost));
    if (fabs(h_s[0] - dot_ref) > 1e-12 * fabs(dot_ref) + 1e-9) { fprintf(stderr, "dot %g vs %g\n", h_s[0], dot_ref); return 1; }
    CHECK(hpcla_cg_update_f64(comm, 0.5, NULL, NULL, x, y2, z, y, n, scal + 1, rwork, NULL)); /* z += .5x ; y -= .5*y2 */
    HIPCHECK(hipMemcpy(h_s + 1, scal + 1, 8, hipMemcpyDeviceToHost));
    double rr_ref = 0.0;
    for (int64_t r = 0; r < n; ++r) { const double rn = h_y[r] - 0.5 * h_y[r]; rr_ref += rn * rn; }
    if (fabs(h_s[1] - rr_ref) > 1e-12 * rr_ref) { fprintf(stderr, "cg_update rr %g vs %g\n", h_s[1], rr_ref); return 1; }

    /* (5) 12 CG iterations enqueued by ONE call (hpcla_cg_iterations_f64_i32: no host language in the loop), b = x,
     * against the textbook recurrence on the host: row-sequential SpMV like the reference loop; the dot products are
     * tree sums on the device, so the residual history is compared at 1e-12 relative (BASELINE's tolerance) */
    {
        const int iters = 12;
        double *cx, *cr, *cp, *cAp, *hist, *pAp;
        void *dot_work;
        HIPCHECK(hipMalloc((void **)&cx, n * 8)); HIPCHECK(hipMalloc((void **)&cr, n * 8));
        HIPCHECK(hipMalloc((void **)&cp, n * 8)); HIPCHECK(hipMalloc((void **)&cAp, n * 8));
        HIPCHECK(hipMalloc((void **)&hist, (iters + 1) * 8)); HIPCHECK(hipMalloc((void **)&pAp, 8));
        HIPCHECK(hipMalloc(&dot_work, hpcla_spmv_dot_work_bytes(n)));
        HIPCHECK(hipMemset(cx, 0, n * 8));
        HIPCHECK(hipMemcpy(cr, x, n * 8, hipMemcpyDeviceToDevice));
        HIPCHECK(hipMemcpy(cp, x, n * 8, hipMemcpyDeviceToDevice));
        HIPCHECK(hipMemset(hist, 0, (iters + 1) * 8));
        CHECK(hpcla_nrm2sq_f64(comm, cr, n, hist, rwork, NULL));                       /* hist[0] = sum r0^2 */
        CHECK(hpcla_cg_iterations_f64_i32(NULL, comm, rowptr, colval, vals, n, nnz, 0, NULL, 0, NULL, 0, cx, cr, cp, cAp,
                                          hist, pAp, dot_work, rwork, iters, NULL));
        double *h_hist = (double *)malloc((iters + 1) * 8), *h_cx = (double *)malloc(n * 8);
        HIPCHECK(hipMemcpy(h_hist, hist, (iters + 1) * 8, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(h_cx, cx, n * 8, hipMemcpyDeviceToHost));
        double *rx = (double *)calloc(n, 8), *rr_ = (double *)malloc(n * 8), *rp = (double *)malloc(n * 8), *rAp = (double *)malloc(n * 8);
        double rr = 0.0;
        for (int64_t i = 0; i < n; ++i) { rr_[i] = rp[i] = h_x[i]; rr += h_x[i] * h_x[i]; }
        if (fabs(h_hist[0] - rr) > 1e-12 * rr) { fprintf(stderr, "cg: r0.r0 %g vs %g\n", h_hist[0], rr); return 1; }
        for (int it = 0; it < iters; ++it) {
            double pap = 0.0;
            for (int64_t r = 0; r < n; ++r) {
                double acc = 0.0;
                for (int64_t j = h_rp[r]; j < h_rp[r + 1]; ++j) acc += h_val[j] * rp[h_col[j]];
                rAp[r] = acc; pap += rp[r] * acc;
            }
            const double a = rr / pap;
            double rr_new = 0.0;
            for (int64_t i = 0; i < n; ++i) { rr_[i] -= a * rAp[i]; rr_new += rr_[i] * rr_[i]; }
            const double bta = rr_new / rr;
            for (int64_t i = 0; i < n; ++i) { rx[i] += a * rp[i]; rp[i] = rr_[i] + bta * rp[i]; }
            rr = rr_new;
            if (fabs(h_hist[it + 1] - rr) > 1e-12 * rr) {
                fprintf(stderr, "cg: iteration %d: sum r^2 %.17g (device) vs %.17g (host)\n", it + 1, h_hist[it + 1], rr);
                return 1;
            }
        }
        double xmax = 0.0, dmax = 0.0;
        for (int64_t i = 0; i < n; ++i) { xmax = fmax(xmax, fabs(rx[i])); dmax = fmax(dmax, fabs(rx[i] - h_cx[i])); }
        if (dmax > 1e-12 * xmax) { fprintf(stderr, "cg: x differs by %g of max |x| %g\n", dmax, xmax); return 1; }
        printf("C-ABI CG: %d iterations in one call, residual history within 1e-12 of the host recurrence, |dx|/|x| = %.1e\n",
               iters, dmax / xmax);
    }

    /* error convention: negative status + message, never abort */
    /* block-order hint: any power of two is a valid order (same y), anything else is refused */
    {
        double *yn, *yo;
        HIPCHECK(hipMalloc((void **)&yn, n * 8));
        HIPCHECK(hipMalloc((void **)&yo, n * 8));
        CHECK(hpcla_spmv_csr_f64_i32(rowptr, colval, vals, x, yn, n, nnz, 0, NULL));
        CHECK(hpcla_spmv_block_order_hint(rowptr, 4));
        CHECK(hpcla_spmv_csr_f64_i32(rowptr, colval, vals, x, yo, n, nnz, 0, NULL));
        CHECK(hpcla_spmv_block_order_hint(rowptr, 0));
        double *hy = (double *)malloc(n * 8), *hyo = (double *)malloc(n * 8);
        HIPCHECK(hipMemcpy(hy, yn, n * 8, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(hyo, yo, n * 8, hipMemcpyDeviceToHost));
        if (memcmp(hy, hyo, n * 8) != 0) { fprintf(stderr, "grouped block order changed y\n"); return 1; }
        if (hpcla_spmv_block_order_hint(rowptr, 3) != HPCLA_ERR_INVALID) { fprintf(stderr, "bad group accepted\n"); return 1; }
        free(hy); free(hyo);
        HIPCHECK(hipFree(yn)); HIPCHECK(hipFree(yo));
        printf("block-order hint: same bits under groups of 4\n");
    }
    /* Float32 element type (csrc/f32.hip): the same matrix with float values, against the same loop in float */
    {
        float *h_vf = (float *)malloc(nnz * 4), *h_xf = (float *)malloc(n * 4), *h_yf = (float *)malloc(n * 4);
        for (int64_t j = 0; j < nnz; ++j) h_vf[j] = (float)h_val[j];
        for (int64_t r = 0; r < n; ++r) h_xf[r] = (float)h_x[r];
        float *vf, *xf, *yf;
        double *dscal;
        HIPCHECK(hipMalloc((void **)&vf, nnz * 4));
        HIPCHECK(hipMalloc((void **)&xf, n * 4));
        HIPCHECK(hipMalloc((void **)&yf, n * 4));
        HIPCHECK(hipMalloc((void **)&dscal, 8));
        HIPCHECK(hipMemcpy(vf, h_vf, nnz * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(xf, h_xf, n * 4, hipMemcpyHostToDevice));
        CHECK(hpcla_spmv_csr_f32_i32(rowptr, colval, vf, xf, yf, n, nnz, 0, NULL));
        CHECK(hpcla_dot_f32(comm, xf, yf, n, dscal, rwork, NULL));
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemcpy(h_yf, yf, n * 4, hipMemcpyDeviceToHost));
        int64_t badf = 0;
        double dotf_ref = 0.0, dotf = 0.0;
        for (int64_t r = 0; r < n; ++r) {
            volatile float acc = 0.0f;                     /* volatile: no contraction, no wider intermediate */
            for (int64_t j = h_rp[r]; j < h_rp[r + 1]; ++j) { volatile float prod = h_vf[j] * h_xf[h_col[j]]; acc = acc + prod; }
            float a = acc;
            if (memcmp(&a, &h_yf[r], 4) != 0) ++badf;
            dotf_ref += (double)h_xf[r] * (double)a;
        }
        HIPCHECK(hipMemcpy(&dotf, dscal, 8, hipMemcpyDeviceToHost));
        if (badf) { fprintf(stderr, "Float32 SpMV mismatch in %lld rows\n", (long long)badf); return 1; }
        if (fabs(dotf - dotf_ref) > 1e-12 * fabs(dotf_ref) + 1e-9) { fprintf(stderr, "dot_f32 %g vs %g\n", dotf, dotf_ref); return 1; }
        free(h_vf); free(h_xf); free(h_yf);
        HIPCHECK(hipFree(vf)); HIPCHECK(hipFree(xf)); HIPCHECK(hipFree(yf)); HIPCHECK(hipFree(dscal));
        printf("Float32: SpMV bit-identical to the scalar float loop, dot formed in double\n");
    }
    if (hpcla_spmv_csr_f64_i32(NULL, NULL, NULL, NULL, NULL, 5, 5, 0, NULL) != HPCLA_ERR_INVALID ||
        strlen(hpcla_last_error()) == 0) { fprintf(stderr, "error convention broken\n"); return 1; }
    CHECK(hpcla_comm_destroy(comm));
    printf("C-ABI smoke PASS: n=%lld nnz=%lld, SpMV bit-identical to the scalar loop on both entry points\n",
           (long long)n, (long long)nnz);
    return 0;
}
