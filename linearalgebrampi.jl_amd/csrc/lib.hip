// lib.hip -- library-level entry points: version, error text, device selection.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace hpcla {

uint64_t host_identity();   // window.hip

char *err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int hpcla_version(void) { return HPCLA_VERSION; }

HPCLA_API const char *hpcla_last_error(void) { return err_buf(); }

HPCLA_API int hpcla_device_count(int *count)
{
    if (!count) return set_error(HPCLA_ERR_INVALID, "device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        *count = 0;
        return set_error(HPCLA_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    }
    *count = n;
    return HPCLA_OK;
}

HPCLA_API int hpcla_set_device(int device)
{
    HPCLA_CHECK_HIP(hipSetDevice(device));
    return HPCLA_OK;
}

HPCLA_API int hpcla_device_identity(int device, uint64_t *id)
{
    if (!id) return set_error(HPCLA_ERR_INVALID, "device_identity: null pointer");
    char bus[64] = "";
    HPCLA_CHECK_HIP(hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device));
    uint64_t h = host_identity();
    for (const char *c = bus; *c; ++c) { h ^= (uint8_t)*c; h *= 0x100000001b3ull; }
    *id = h;
    return HPCLA_OK;
}

HPCLA_API int hpcla_device_info(int device, int *num_cus, char *arch_name_host, int arch_name_len)
{
    hipDeviceProp_t prop;
    HPCLA_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (arch_name_host && arch_name_len > 0) {
        strncpy(arch_name_host, prop.gcnArchName, (size_t)arch_name_len - 1);
        arch_name_host[arch_name_len - 1] = '\0';
    }
    return HPCLA_OK;
}
