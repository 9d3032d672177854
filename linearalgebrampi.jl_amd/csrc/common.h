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

// 1-D grid size guard: a launch needs < 2^31 workgroups
#define HPCLA_CHECK_GRID(nblocks, what)                                                           \
    do {                                                                                          \
        if ((int64_t)(nblocks) > 0x7fffffffLL)                                                    \
            return hpcla::set_error(HPCLA_ERR_UNSUPPORTED, "%s: %lld workgroups exceed the grid limit", \
                                    what, (long long)(nblocks));                                  \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Workgroup order: MI355X deals workgroups round-robin over its 8 XCDs (block b and b+8 share an L2).
// An XCD-sliced remap (each XCD walking one contiguous eighth of the matrix) was measured SLOWER than
// the natural blockIdx order for SpMV (0.255 vs 0.241 ms) and SpMM (+3 %): one moving window over the
// matrix keeps DRAM pages and the gathered x / B rows hot for all XCDs through the 256 MiB Infinity
// Cache.  The kernels therefore use blockIdx.x directly (profiles/r01_tune_spmv_variants_first.log).

}  // namespace hpcla
