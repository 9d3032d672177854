// probe_ipc_push.hip -- feasibility probe for the peer-mapped halo push.
//
// Two PROCESSES (rank 0 / rank 1, started separately; they may share one GPU) each allocate a
// "window" (flags + ghost segment), export it with hipIpcGetMemHandle through a file, map the
// other's window, and then run K rounds of: push kernel (payload stores + flag into the PEER's
// window) -> wait kernel (poll own flag, acquire, checksum the payload).  Reports the per-round
// latency and whether every word arrived.  Not part of the library; build:
//   hipcc --offload-arch=gfx950 -O3 benchmarks/probe_ipc_push.hip -o benchmarks/_build/probe_ipc_push
// usage: probe_ipc_push RANK DIR [alloc: 0 hipMalloc, 1 finegrained, 2 uncached] [count] [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>

#define CK(e)                                                                                      \
    do {                                                                                           \
        hipError_t _e = (e);                                                                       \
        if (_e != hipSuccess) {                                                                    \
            fprintf(stderr, "rank %d: %s failed: %s (line %d)\n", g_rank, #e, hipGetErrorString(_e), __LINE__); \
            exit(2);                                                                               \
        }                                                                                          \
    } while (0)

static int g_rank = 0;

struct Window {          // layout of one rank's window
    uint64_t flag;       // written by the peer: epoch of the payload below
    uint64_t pad[15];
    double ghost[1];     // count doubles
};

__global__ void push_kernel(const double *__restrict__ x, double *peer_ghost, uint64_t *peer_flag,
                            int64_t n, uint64_t epoch)
{
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
        __hip_atomic_store(peer_ghost + i, x[i] + (double)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);   // system scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(peer_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void wait_kernel(const double *ghost, const uint64_t *flag, int64_t n, uint64_t epoch,
                            double *sum_out, uint32_t *timeout_out)
{
    __shared__ double s[256];
    __shared__ int ok;
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        int good = 1;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > 20000000u) { good = 0; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok = good;
        if (!good) *timeout_out = 1;
    }
    __syncthreads();
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) a += ghost[i];
    s[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += s[i];
        *sum_out = ok ? t : -1.0;
    }
}

static void write_file(const std::string &p, const void *d, size_t n)
{
    std::string tmp = p + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    fwrite(d, 1, n, f);
    fclose(f);
    rename(tmp.c_str(), p.c_str());
}

static bool read_file(const std::string &p, void *d, size_t n, int timeout_s)
{
    for (int i = 0; i < timeout_s * 100; ++i) {
        FILE *f = fopen(p.c_str(), "rb");
        if (f) {
            size_t got = fread(d, 1, n, f);
            fclose(f);
            if (got == n) return true;
        }
        usleep(10000);
    }
    return false;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s RANK DIR [alloc] [count] [rounds]\n", argv[0]); return 1; }
    g_rank = atoi(argv[1]);
    const std::string dir = argv[2];
    const int alloc = argc > 3 ? atoi(argv[3]) : 2;
    const int64_t n = argc > 4 ? atoll(argv[4]) : 4096;
    const int rounds = argc > 5 ? atoi(argv[5]) : 200;
    const int peer = 1 - g_rank;
    CK(hipSetDevice(0));
    const size_t bytes = sizeof(Window) + n * sizeof(double);
    void *win = nullptr;
    if (alloc == 0) CK(hipMalloc(&win, bytes));
    else if (alloc == 1) CK(hipExtMallocWithFlags(&win, bytes, hipDeviceMallocFinegrained));
    else CK(hipExtMallocWithFlags(&win, bytes, hipDeviceMallocUncached));
    CK(hipMemset(win, 0, bytes));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t mine, theirs;
    CK(hipIpcGetMemHandle(&mine, win));
    write_file(dir + "/h" + std::to_string(g_rank), &mine, sizeof(mine));
    if (!read_file(dir + "/h" + std::to_string(peer), &theirs, sizeof(theirs), 60)) {
        fprintf(stderr, "rank %d: peer handle never appeared\n", g_rank);
        return 3;
    }
    void *pw = nullptr;
    CK(hipIpcOpenMemHandle(&pw, theirs, hipIpcMemLazyEnablePeerAccess));
    Window *w = (Window *)win, *p = (Window *)pw;

    double *x = nullptr, *sum = nullptr;
    uint32_t *tmo = nullptr;
    CK(hipMalloc((void **)&x, n * sizeof(double)));
    CK(hipMalloc((void **)&sum, sizeof(double)));
    CK(hipMalloc((void **)&tmo, 4));
    CK(hipMemset(tmo, 0, 4));
    double *hx = (double *)malloc(n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) hx[i] = (double)((i * 7 + g_rank) % 13);
    CK(hipMemcpy(x, hx, n * sizeof(double), hipMemcpyHostToDevice));
    // what the PEER sends me: its x[i] + epoch
    double base = 0.0;
    for (int64_t i = 0; i < n; ++i) base += (double)((i * 7 + peer) % 13);

    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int bad = 0;
    // round 1..rounds; single ghost buffer, so round r+1's push may only start once the peer has
    // consumed round r: true here because each rank's push(r+1) follows its own wait(r) in stream
    // order and wait(r) needs the peer's push(r) -- but the PEER's wait(r) may still be reading when my
    // push(r+1) lands.  The probe therefore checks sums only every round with a host sync (strict),
    // then times a free-running loop without checking.
    for (int r = 1; r <= 20; ++r) {
        push_kernel<<<1, 256, 0, s>>>(x, p->ghost, &p->flag, n, (uint64_t)r);
        wait_kernel<<<1, 256, 0, s>>>(w->ghost, &w->flag, n, (uint64_t)r, sum, tmo);
        CK(hipStreamSynchronize(s));
        double hs = 0;
        CK(hipMemcpy(&hs, sum, 8, hipMemcpyDeviceToHost));
        const double want = base + (double)r * (double)n;
        if (hs != want) { ++bad; fprintf(stderr, "rank %d round %d: sum %.1f want %.1f\n", g_rank, r, hs, want); }
        // handshake through files so that the peer has also finished reading before the next push
        char c = 1;
        write_file(dir + "/s" + std::to_string(g_rank) + "_" + std::to_string(r), &c, 1);
        if (!read_file(dir + "/s" + std::to_string(peer) + "_" + std::to_string(r), &c, 1, 60)) return 4;
    }
    CK(hipEventRecord(e0, s));
    for (int r = 21; r <= 20 + rounds; ++r) {
        push_kernel<<<1, 256, 0, s>>>(x, p->ghost, &p->flag, n, (uint64_t)r);
        wait_kernel<<<1, 256, 0, s>>>(w->ghost, &w->flag, n, (uint64_t)r, sum, tmo);
    }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t htmo = 0;
    CK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
    printf("rank %d alloc %d n %lld: strict rounds bad=%d, free-running %d rounds: %.2f us/round (push+wait), timeout=%u\n",
           g_rank, alloc, (long long)n, bad, rounds, ms * 1e3 / rounds, htmo);
    // final handshake before unmapping
    char c = 1;
    write_file(dir + "/done" + std::to_string(g_rank), &c, 1);
    read_file(dir + "/done" + std::to_string(peer), &c, 1, 60);
    CK(hipIpcCloseMemHandle(pw));
    CK(hipFree(win));
    return bad || htmo ? 5 : 0;
}
