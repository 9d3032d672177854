// stream_mix.hip -- yardstick: the SpMM workload's BYTE MIX moved as pure streams by one kernel
// (read `ra` + `rb` bytes, write `wc` bytes, 16-byte accesses, grid-stride, nothing else), to see what this
// box can do for a 40 %-write mix independently of any gather / LDS / reuse structure.  Not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_mix(const v2d *__restrict__ a, int64_t na, const v2d *__restrict__ b,
                                             v2d *__restrict__ c, int64_t nb)
{
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    v2d acc = (v2d)(0.0);
    for (int64_t j = i0; j < na; j += stride) acc += __builtin_nontemporal_load(a + j);
    for (int64_t j = i0; j < nb; j += stride) {
        v2d v = b[j];
        c[j] = v + acc;
    }
}

// per-workgroup contiguous slabs instead of grid-stride (each workgroup streams its own region, like a row block)
__global__ __launch_bounds__(256) void k_mix_slab(const v2d *__restrict__ a, int64_t na, const v2d *__restrict__ b,
                                                  v2d *__restrict__ c, int64_t nb)
{
    const int64_t nblk = gridDim.x;
    const int64_t pa = (na + nblk - 1) / nblk, pb = (nb + nblk - 1) / nblk;
    const int64_t a0 = blockIdx.x * pa, a1 = a0 + pa < na ? a0 + pa : na;
    const int64_t b0 = blockIdx.x * pb, b1 = b0 + pb < nb ? b0 + pb : nb;
    v2d acc = (v2d)(0.0);
    for (int64_t j = a0 + threadIdx.x; j < a1; j += 256) acc += __builtin_nontemporal_load(a + j);
    for (int64_t j = b0 + threadIdx.x; j < b1; j += 256) c[j] = b[j] + acc;
}

extern "C" int stream_mix(int variant, const void *a, int64_t a_bytes, const void *b, void *c, int64_t b_bytes,
                          int blocks, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0)
        k_mix<<<blocks, 256, 0, s>>>((const v2d *)a, a_bytes / 16, (const v2d *)b, (v2d *)c, b_bytes / 16);
    else
        k_mix_slab<<<blocks, 256, 0, s>>>((const v2d *)a, a_bytes / 16, (const v2d *)b, (v2d *)c, b_bytes / 16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
