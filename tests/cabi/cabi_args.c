/* cabi_args.c -- the C ABI's HOST-SIDE argument validation, from plain C, with no GPU in the machine.
 *
 * Every entry point validates its arguments on the host before it touches HIP (error convention of
 * include/hpcla_rocm.h: negative status + hpcla_last_error() text; cf. the status checks of the reference's CUDA
 * extension, ext/HPCLinearAlgebraCUDAExt.jl:248-251).  This program walks those paths -- null pointers, negative
 * sizes, bad index bases, bad flags -- and is built by tests/test_sanitizers.py with AddressSanitizer +
 * UndefinedBehaviorSanitizer (CPU only: GPU sanitizers are not available on this pool), so a validation path that
 * reads through a pointer before checking it, or overflows the error buffer, is caught here.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "hpcla_rocm.h"

static int failures = 0;

#define EXPECT_ERR(call)                                                                             \
    do {                                                                                             \
        int _s = (call);                                                                             \
        const char *_m = hpcla_last_error();                                                         \
        if (_s >= 0 || !_m || !_m[0]) {                                                              \
            fprintf(stderr, "%s:%d: expected an error, got status %d\n", __FILE__, __LINE__, _s);    \
            ++failures;                                                                              \
        }                                                                                            \
    } while (0)

#define EXPECT_OK(call)                                                                              \
    do {                                                                                             \
        int _s = (call);                                                                             \
        if (_s != 0) {                                                                               \
            fprintf(stderr, "%s:%d: status %d: %s\n", __FILE__, __LINE__, _s, hpcla_last_error());   \
            ++failures;                                                                              \
        }                                                                                            \
    } while (0)

int main(void)
{
    int32_t i32buf[8] = {0};
    int64_t i64buf[8] = {0};
    double f64buf[8] = {0};
    int chosen = -1;

    if (hpcla_version() != 100) { fprintf(stderr, "version %d\n", hpcla_version()); ++failures; }
    if (hpcla_spmv_rows_per_block() != 256 || hpcla_spmm_rows_per_block() <= 0) ++failures;

    /* SpMV: sizes, index base, null arrays; an empty matrix is fine */
    EXPECT_ERR(hpcla_spmv_csr_f64_i32(NULL, NULL, NULL, NULL, NULL, -1, 0, 0, NULL));
    EXPECT_ERR(hpcla_spmv_csr_f64_i64(NULL, NULL, NULL, NULL, NULL, 4, -1, 0, NULL));
    EXPECT_ERR(hpcla_spmv_csr_f64_i32(i32buf, i32buf, f64buf, f64buf, f64buf, 4, 4, 2, NULL));
    EXPECT_ERR(hpcla_spmv_csr_f64_i32(NULL, i32buf, f64buf, f64buf, f64buf, 4, 4, 0, NULL));
    EXPECT_ERR(hpcla_spmv_csr_f64_i64(i64buf, NULL, f64buf, f64buf, f64buf, 4, 4, 1, NULL));
    EXPECT_OK(hpcla_spmv_csr_f64_i32(NULL, NULL, NULL, NULL, NULL, 0, 0, 0, NULL));
    EXPECT_ERR(hpcla_spmv_split_f64_i32(i32buf, i32buf, f64buf, f64buf, NULL, 4, f64buf, 4, 4, 0, i32buf, -1, NULL));
    EXPECT_ERR(hpcla_spmv_split_f64_i64(i64buf, i64buf, f64buf, f64buf, NULL, 4, f64buf, 4, 4, 0, i32buf, 99, NULL));

    /* the Float32 element type (csrc/f32.hip): the same checks in front of the same kind of launch */
    {
        float f32buf[8] = {0};
        EXPECT_ERR(hpcla_spmv_csr_f32_i32(NULL, NULL, NULL, NULL, NULL, -1, 0, 0, NULL));
        EXPECT_ERR(hpcla_spmv_csr_f32_i64(i64buf, i64buf, f32buf, f32buf, f32buf, 4, 4, 3, NULL));
        EXPECT_ERR(hpcla_spmv_csr_f32_i32(i32buf, NULL, f32buf, f32buf, f32buf, 4, 4, 0, NULL));
        EXPECT_OK(hpcla_spmv_csr_f32_i32(NULL, NULL, NULL, NULL, NULL, 0, 0, 0, NULL));
        EXPECT_ERR(hpcla_spmv_split_f32_i32(i32buf, i32buf, f32buf, f32buf, NULL, -4, f32buf, 4, 4, 0, NULL, 0, NULL));
        EXPECT_ERR(hpcla_spmv_split_f32_i64(i64buf, i64buf, f32buf, f32buf, f64buf, 4, f32buf, 4, 4, 0, i32buf, 99, NULL));
        EXPECT_ERR(hpcla_spmm_csr_f32_i32(i32buf, i32buf, f32buf, f32buf, 4, 9, f32buf, 4, 0, 4, 4, 4, 0, NULL));
        EXPECT_ERR(hpcla_spmm_csr_f32_i32(i32buf, i32buf, f32buf, f32buf, 2, 0, f32buf, 4, 0, 4, 4, 4, 0, NULL));
        EXPECT_ERR(hpcla_spmm_split_f32_i64(i64buf, i64buf, f32buf, f32buf, 2, NULL, 4, 4, f32buf, 4, 4, 4, 4, 0, NULL, 0, NULL));
        EXPECT_OK(hpcla_spmm_csr_f32_i64(NULL, NULL, NULL, NULL, 1, 0, NULL, 1, 0, 4, 0, 0, 0, NULL));
        EXPECT_ERR(hpcla_halo_begin_f32(NULL, f32buf, f64buf, NULL));
        EXPECT_ERR(hpcla_dot_f32(NULL, NULL, f32buf, 4, f64buf, f64buf, NULL));
        EXPECT_ERR(hpcla_nrm2sq_f32(NULL, f32buf, -1, f64buf, f64buf, NULL));
        EXPECT_ERR(hpcla_asum_f32(NULL, f32buf, 4, NULL, f64buf, NULL));
        EXPECT_ERR(hpcla_axpby_f32(1.0f, NULL, 1.0f, f32buf, f32buf, 4, NULL));
        EXPECT_ERR(hpcla_scale_f32(1.0f, f32buf, f32buf, -1, NULL));
        EXPECT_OK(hpcla_divide_f32(NULL, 2.0f, NULL, 0, NULL));
    }

    /* block order: pointer and group */
    EXPECT_ERR(hpcla_spmv_block_order_hint(NULL, 4));
    EXPECT_ERR(hpcla_spmv_block_order_hint(i32buf, 3));
    EXPECT_ERR(hpcla_spmv_block_order_hint(i32buf, 2048));
    EXPECT_ERR(hpcla_spmv_block_order_hint(i32buf, -2));
    EXPECT_OK(hpcla_spmv_block_order_hint(i32buf, 64));
    EXPECT_OK(hpcla_spmv_block_order_hint(i32buf, 0));
    EXPECT_ERR(hpcla_spmv_tune_block_order_f64_i32(NULL, NULL, NULL, NULL, NULL, 0, NULL, 10, 10, 0, NULL, &chosen));
    if (chosen != 1) { fprintf(stderr, "tuner must report the natural order on failure, got %d\n", chosen); ++failures; }
    EXPECT_ERR(hpcla_spmv_tune_block_order_f64_i64(NULL, NULL, NULL, NULL, NULL, 0, NULL, -1, 10, 0, NULL, NULL));

    /* plan-time helpers, incl. the Int64 -> Int32 narrowing entries */
    EXPECT_ERR(hpcla_remap_i32(NULL, NULL, NULL, 5, 0, NULL));
    EXPECT_ERR(hpcla_remap_i64(i64buf, i64buf, NULL, 5, 0, NULL));
    EXPECT_ERR(hpcla_remap_i32(i32buf, i32buf, i32buf, -5, 0, NULL));
    EXPECT_ERR(hpcla_remap_i64_to_i32(NULL, i32buf, i32buf, 5, 0, NULL));
    EXPECT_ERR(hpcla_remap_i64_to_i32(i64buf, i32buf, i32buf, 5, 7, NULL));
    EXPECT_OK(hpcla_remap_i64_to_i32(NULL, NULL, NULL, 0, 1, NULL));
    EXPECT_ERR(hpcla_narrow_i64_to_i32(NULL, i32buf, 5, NULL, NULL));
    EXPECT_ERR(hpcla_narrow_i64_to_i32(i64buf, i32buf, -1, NULL, NULL));
    EXPECT_OK(hpcla_narrow_i64_to_i32(NULL, NULL, 0, NULL, NULL));
    EXPECT_ERR(hpcla_classify_blocks_i32(NULL, i32buf, 10, 0, 5, 256, i32buf, NULL));
    EXPECT_ERR(hpcla_classify_blocks_i64(i64buf, i64buf, 10, 0, 5, 0, i32buf, NULL));
    EXPECT_ERR(hpcla_gather_f64_i32(NULL, i32buf, NULL, f64buf, 5, 0, NULL));
    EXPECT_ERR(hpcla_gather_f64_i64(f64buf, i64buf, NULL, f64buf, -5, 0, NULL));

    /* halo plans and communicator: null handles */
    hpcla_halo_plan_t *plan = NULL;
    EXPECT_ERR(hpcla_halo_plan_create(&plan, NULL, 0, NULL, NULL, NULL, 0, 0, NULL, NULL, 1));
    EXPECT_ERR(hpcla_halo_plan_create(NULL, NULL, 0, NULL, NULL, NULL, 0, 0, NULL, NULL, 1));
    EXPECT_ERR(hpcla_halo_begin(NULL, f64buf, NULL));
    EXPECT_ERR(hpcla_halo_end(NULL, NULL));
    EXPECT_ERR(hpcla_halo_status(NULL, &chosen));
    EXPECT_ERR(hpcla_halo_plan_chain(NULL, NULL));
    EXPECT_ERR(hpcla_comm_status(NULL, &chosen));
    EXPECT_OK(hpcla_halo_plan_destroy(NULL));

    /* the error text is bounded and terminated whatever went into it */
    const char *msg = hpcla_last_error();
    if (!msg || strlen(msg) > 4096) { fprintf(stderr, "error text unbounded\n"); ++failures; }

    if (failures) { fprintf(stderr, "C-ABI argument validation: %d FAILURE(S)\n", failures); return 1; }
    printf("C-ABI argument validation PASS\n");
    return 0;
}
