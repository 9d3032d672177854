// common.h -- shared helpers of libhpcla_rocm (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hpcla_rocm.h"

#define HPCLA_API extern "C" __attribute__((visibility("default")))

namespace hpcla {

// thread-local last-error text, returned by hpcla_last_error()
char *err_buf();
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HPCLA_CHECK_HIP(expr)                                                                     \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return hpcla::set_error(HPCLA_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                                    hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)

#define HPCLA_REQUIRE(cond, msg)                                                                  \
    do {                                                                                          \
        if (!(cond)) return hpcla::set_error(HPCLA_ERR_INVALID, "%s: %s", __func__, msg);         \
    } while (0)

// launch check: hipGetLastError after a <<<>>> launch
#define HPCLA_CHECK_LAUNCH() HPCLA_CHECK_HIP(hipGetLastError())

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// MI355X: 8 XCDs, workgroups are dealt round-robin over them (block b and b+8 share an XCD's L2).
// Map the launch index so that every XCD walks ONE contiguous slice of the work in order: the x
// entries a row block gathers were fetched into the same L2 by the previous blocks of that slice.
// Bijective for any n (MI355X_MICROARCH.md, "Workgroup dispatch"; speed only, never correctness).
constexpr uint32_t NUM_XCD = 8;
__device__ __forceinline__ uint32_t xcd_slice_index(uint32_t b, uint32_t n)
{
    uint32_t k = b % NUM_XCD, q = b / NUM_XCD;
    uint32_t per = n / NUM_XCD, rem = n % NUM_XCD;
    uint32_t start = k * per + (k < rem ? k : rem);
    return start + q;
}

}  // namespace hpcla
