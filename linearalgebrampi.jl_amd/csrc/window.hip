// window.hip -- peer-window push transport: halo exchange and scalar all-reduce by DIRECT stores into
// the neighbours' memory over xGMI, no RCCL kernel and no side stream.
//
// Replaces the exchange half of execute_plan! (MPI Isend/Irecv on host buffers, src/vectors.jl:431-455)
// and comm_allreduce of a scalar (src/backends.jl:264-277) on one node.  Why not RCCL here: measured on
// MI355X (profiles/r01_halo_mode_experiments.log), the ncclSend/ncclRecv group of a 2 x 32 KiB stencil
// halo costs ~13 us on the caller's stream and cannot overlap the SpMV kernel, which occupies every
// wave slot of every CU.  xGMI is point-to-point and peer memory is mappable, so the natural MI355X
// form is one-sided:
//
//   * every plan owns a WINDOW (fine-grained allocation: control lines + ghost buffers) that its
//     neighbours map once, at plan time, through hipIpcOpenMemHandle (handles travel through the host
//     runtime exactly like the RCCL unique id, ext/HPCLinearAlgebraCUDAExt.jl:411-443);
//   * per SpMV, a small PUSH kernel on the caller's stream stores x[send_indices] straight into each
//     neighbour's ghost buffer (system-scope write-through stores), drains them, and publishes the
//     step's epoch in the neighbour's flag line;
//   * the consumer is the SpMV launch itself: its boundary workgroups are dispatched LAST and poll the
//     flag lines (one lane, relaxed system-scope loads, one acquire) before their first ghost gather
//     (spmv.hip), so the exchange overlaps all interior row blocks without a second stream;
//   * ghost buffers are double-buffered and guarded by ACK lines (the consumer's next push publishes
//     "I have finished reading epoch e-1" before it waits for anything), so a fast rank can never
//     overwrite values a slow neighbour is still reading, for symmetric and asymmetric patterns alike.
//
// The step epochs live in device memory (halo_wait.h), so distributed steps are HIP-graph capturable.
// Every spin is bounded (HPCLA_PUSH_TIMEOUT_S, default 300 s -- long enough to WAIT for a slow neighbour the
// way a blocking MPI receive would): on expiry the kernel sets the plan's sticky status word, POISONS what the
// expired wait would have fed (boundary rows / dot partials / all-reduce results become NaN, a push whose ack
// wait expired stores nothing and publishes nothing) and carries on, so a grid always drains and a timeout can
// neither corrupt a neighbour nor pass as a result; hpcla_halo_status / hpcla_comm_status report it, and the
// host layers check them wherever a NaN scalar reaches the host.
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <array>
#include <map>
#include <mutex>

#include "comm_internal.h"
#include "halo_wait.h"

namespace hpcla {

// ---- host helpers ----------------------------------------------------------------------------
uint64_t host_identity()
{
    // boot id (one per running kernel) + host name: equal for all processes that can share IPC handles
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&h](const char *s, size_t n) {
        for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)s[i]; h *= 0x100000001b3ull; }
    };
    char buf[256];
    FILE *f = fopen("/proc/sys/kernel/random/boot_id", "r");
    if (f) {
        size_t n = fread(buf, 1, sizeof(buf), f);
        fclose(f);
        mix(buf, n);
    }
    if (gethostname(buf, sizeof(buf)) == 0) mix(buf, strnlen(buf, sizeof(buf)));
    return h ? h : 1;
}

int64_t spin_timeout_ticks()
{
    static const int64_t ticks = [] {
        const char *e = getenv("HPCLA_PUSH_TIMEOUT_S");
        // default 300 s: a slow neighbour (host I/O, a plan build on one rank, a debugger) must be WAITED for like a
        // blocking MPI receive would (src/vectors.jl:446); the bound only exists so that a grid drains when a peer
        // has died.  Expiry poisons the result (NaN) and sets the sticky status word; tests and bench.py set
        // shorter bounds.
        double s = e ? atof(e) : 300.0;
        if (!(s > 0.0)) s = 300.0;
        return (int64_t)(s * 1.0e8);           // wall_clock64 runs at 100 MHz
    }();
    return ticks;
}

static uint64_t device_identity_of(int device)
{
    uint64_t id = 0;
    return hpcla_device_identity(device, &id) == HPCLA_OK ? id : 0;
}

int window_alloc(void **p, size_t bytes, bool uncached)
{
    // Memory that PEERS store into while kernels of this device read it.  Control lines and vector ghost
    // segments (KiB..MiB, each value read once or twice per step): UNCACHED -- never held in this device's L2s,
    // so a value stored by a peer cannot be shadowed by a stale line whatever the acquire does or does not
    // invalidate.  Dense ghost rows (config 5: 1.9 GB gathered ~30x per row): FINE-GRAINED -- cacheable, kept
    // coherent by the system-scope acquire in front of the consumer (and a kernel boundary).
    // HPCLA_WINDOW_ALLOC=finegrained|uncached overrides both.
    static const int forced = [] {
        const char *e = getenv("HPCLA_WINDOW_ALLOC");
        return !e ? -1 : (e[0] == 'u' ? 1 : 0);
    }();
    if (forced >= 0) uncached = forced == 1;
    // either flavour is correct for either use (they differ in speed: DESIGN.md section 4); if the preferred one
    // cannot be had on this system, take the other before giving the transport up
    hipError_t ea = hipExtMallocWithFlags(p, bytes, uncached ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    if (ea != hipSuccess) {
        (void)hipGetLastError();
        ea = hipExtMallocWithFlags(p, bytes, uncached ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    }
    if (ea != hipSuccess) {
        *p = nullptr;
        return set_error(HPCLA_ERR_HIP, "window allocation of %zu bytes failed: %s", bytes, hipGetErrorString(ea));
    }
    hipError_t e = hipMemset(*p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(*p);
        *p = nullptr;
        return set_error(HPCLA_ERR_HIP, "window memset failed: %s", hipGetErrorString(e));
    }
    return HPCLA_OK;
}

// Windows exported by THIS process, by IPC handle.  A handle cannot be opened in the process that made it, and a
// process may host several ranks (one host thread per rank, each with its own communicator -- the arrangement of
// tests/cabi/cabi_ranks_threads.c, which rehearses 8 ranks inside the box's limit on GPU processes): such peers
// are reached through the exporter's own pointer.  An IPC mapping keeps the peer's memory alive until it is closed; the
// raw pointer does not, so the registry COUNTS the in-process peers that hold it: window_free by the owner while a peer
// still maps the window only marks it, and the last window_close frees it (a peer's push or ack kernels may
// still be in flight when the owner destroys its plan).
struct LocalWindow {
    void *base;
    int refs;               // in-process peers that hold `base` (window_open .. window_close)
    bool owner_freed;       // the owner called window_free while refs > 0: hipFree is owed
};
static std::mutex g_local_mu;
static std::map<std::array<uint8_t, 64>, LocalWindow> g_local_windows;

static void local_window_register(const uint8_t *ipc, void *base)
{
    std::array<uint8_t, 64> k;
    memcpy(k.data(), ipc, 64);
    std::lock_guard<std::mutex> lock(g_local_mu);
    g_local_windows[k] = LocalWindow{base, 0, false};
}

static void *local_window_acquire(const uint8_t *ipc)
{
    std::array<uint8_t, 64> k;
    memcpy(k.data(), ipc, 64);
    std::lock_guard<std::mutex> lock(g_local_mu);
    auto it = g_local_windows.find(k);
    if (it == g_local_windows.end() || it->second.owner_freed) return nullptr;
    ++it->second.refs;
    return it->second.base;
}

static void local_window_release(void *base)
{
    void *to_free = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_local_mu);
        for (auto it = g_local_windows.begin(); it != g_local_windows.end(); ++it) {
            if (it->second.base != base) continue;
            if (--it->second.refs <= 0 && it->second.owner_freed) {
                to_free = base;
                g_local_windows.erase(it);
            }
            break;
        }
    }
    if (to_free) (void)hipFree(to_free);
}

void window_free(void *win)
{
    if (!win) return;
    static const bool defer = [] {
        const char *e = getenv("HPCLA_WINDOW_DEFER_FREE");       // 0: the owner's free is immediate (diagnostics)
        return !(e && e[0] == '0');
    }();
    {
        std::lock_guard<std::mutex> lock(g_local_mu);
        for (auto it = g_local_windows.begin(); it != g_local_windows.end(); ++it) {
            if (it->second.base != win) continue;
            if (defer && it->second.refs > 0) {       // an in-process peer still maps it: its last window_close frees it
                it->second.owner_freed = true;
                return;
            }
            g_local_windows.erase(it);
            break;
        }
    }
    (void)hipFree(win);
}

// This process's device index whose identity is `device_id`, or -1 (not visible here).
static int local_device_of(uint64_t device_id)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return -1;
    for (int dev = 0; dev < ndev; ++dev)
        if (device_identity_of(dev) == device_id) return dev;
    return -1;
}

int window_open(const WindowDesc &d, int peer_rank, int my_rank, void *my_base, PeerMap *out)
{
    if (d.bytes == 0) return set_error(HPCLA_ERR_INVALID, "window_open: rank %d exported no window", peer_rank);
    if (d.pid == (uint64_t)getpid() && d.host_id == host_identity()) {
        // a window of this process: mine, or that of another rank hosted here
        if (peer_rank == my_rank) {
            out->base = my_base;
            out->opened = false;
            out->local_ref = false;
            return HPCLA_OK;
        }
        void *base = local_window_acquire(d.ipc);
        if (base) {
            // The exporter's own pointer carries none of what an IPC mapping provides.  If the hosted rank runs on ANOTHER
            // device of this process, stores through the pointer need peer access between the two devices: checked and
            // enabled here, refused -- an error code, not a GPU fault -- where the runtime reports none.
            int cur = 0;
            if (d.device_id != 0 && hipGetDevice(&cur) == hipSuccess && d.device_id != device_identity_of(cur)) {
                const int dev = local_device_of(d.device_id);
                int can = 0;
                if (dev < 0 || hipDeviceCanAccessPeer(&can, cur, dev) != hipSuccess || !can) {
                    local_window_release(base);
                    return set_error(HPCLA_ERR_UNSUPPORTED,
                                     "window_open: device %d has no peer access to the device of in-process rank %d", cur, peer_rank);
                }
                const hipError_t pe = hipDeviceEnablePeerAccess(dev, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
                    local_window_release(base);
                    return set_error(HPCLA_ERR_HIP, "hipDeviceEnablePeerAccess(rank %d) failed: %s", peer_rank, hipGetErrorString(pe));
                }
                (void)hipGetLastError();      // (already enabled: clear the sticky error)
            }
            out->base = base;
            out->opened = false;
            out->local_ref = true;
            return HPCLA_OK;
        }
        // not exported here: ANOTHER process that happens to carry this pid (ranks in separate PID namespaces) -- open it
        // over IPC like any other peer
    }
    if (d.host_id != host_identity())
        return set_error(HPCLA_ERR_UNSUPPORTED, "window_open: rank %d is on another node (push transport is per node)",
                         peer_rank);
    // The peer's GPU, if this process can see it: refuse the mapping when the runtime reports no peer access
    // between the two devices (stores through such a mapping would fault the GPU instead of returning an error).
    // An invisible peer device (HIP_VISIBLE_DEVICES per rank) cannot be asked about; the IPC open then decides.
    int cur = 0;
    if (d.device_id != 0 && hipGetDevice(&cur) == hipSuccess && d.device_id != device_identity_of(cur)) {
        const int dev = local_device_of(d.device_id);
        int can = 0;
        if (dev >= 0 && dev != cur && hipDeviceCanAccessPeer(&can, cur, dev) == hipSuccess && !can)
            return set_error(HPCLA_ERR_UNSUPPORTED, "window_open: device %d has no peer access to rank %d's device %d",
                             cur, peer_rank, dev);
    }
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t size");
    memcpy(&h, d.ipc, sizeof(h));
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess)
        return set_error(HPCLA_ERR_HIP, "hipIpcOpenMemHandle(rank %d) failed: %s", peer_rank, hipGetErrorString(e));
    out->base = p;
    out->opened = true;
    out->local_ref = false;
    return HPCLA_OK;
}

void window_close(PeerMap *m)
{
    if (m->opened && m->base) (void)hipIpcCloseMemHandle(m->base);
    if (m->local_ref && m->base) local_window_release(m->base);
    m->base = nullptr;
    m->opened = false;
    m->local_ref = false;
}

static void fill_desc(WindowDesc *d, void *win, size_t bytes)
{
    memset(d, 0, sizeof(*d));
    d->host_id = host_identity();
    d->pid = (uint64_t)getpid();
    d->bytes = bytes;
    int cur = 0;
    if (hipGetDevice(&cur) == hipSuccess) d->device_id = device_identity_of(cur);
    if (win) {
        hipIpcMemHandle_t h;
        if (hipIpcGetMemHandle(&h, win) == hipSuccess) {
            memcpy(d->ipc, &h, sizeof(h));
            local_window_register(d->ipc, win);
        } else {
            d->bytes = 0;
        }
    }
}

// ---- halo plan window ------------------------------------------------------------------------
static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t halo_ctrl_bytes(uint32_t n_flags, uint32_t n_acks)
{
    return round_up((size_t)(n_flags + n_acks + 1) * WIN_LINE, 256);
}
static size_t halo_buf_bytes(uint64_t n_ghost, uint32_t width)
{
    return round_up((size_t)n_ghost * width * sizeof(double), 256);
}
// (for comm.hip's probe: doubles between the two ghost buffers of a double-buffered window)
size_t halo_window_buf_bytes(uint64_t n_ghost, uint32_t width) { return halo_buf_bytes(n_ghost, width); }

int push_plan_alloc(hpcla_halo_plan *p)
{
    const uint32_t nf = (uint32_t)p->recv_ranks.size(), na = (uint32_t)p->send_ranks.size();
    const size_t ctrl = halo_ctrl_bytes(nf, na);
    const size_t buf = halo_buf_bytes((uint64_t)p->n_ghost, (uint32_t)p->width);
    // vectors (the fused SpMV computes the buffer of its epoch in the kernel): two buffers, one step of slack
    // between neighbours; dense ghost rows -- of ANY width, a one-column B included (HPCLA_HALO_SINGLE_BUFFER) --
    // whose consumers take the ghost pointer from the host between halo_begin and halo_end: one buffer, the ack
    // wait orders producer and consumer strictly
    p->nbuf = (p->width == 1 && !p->single_buffer && (int64_t)buf <= WIN_DOUBLE_BUFFER_MAX) ? 2 : 1;
    p->win_bytes = ctrl + buf * p->nbuf;
    int rc = window_alloc(&p->win, p->win_bytes, p->width == 1);
    if (rc) return rc;
    uint8_t *base = reinterpret_cast<uint8_t *>(p->win);
    p->flags = reinterpret_cast<uint64_t *>(base);
    p->acks = reinterpret_cast<uint64_t *>(base + (size_t)nf * WIN_LINE);
    p->status = reinterpret_cast<uint32_t *>(base + (size_t)(nf + na) * WIN_LINE);
    p->ghost = reinterpret_cast<double *>(base + ctrl);
    // step counter {done, ticket}: ordinary device memory, only this rank's kernels touch it
    hipError_t e = hipMalloc((void **)&p->epoch_dev, EPOCH_BYTES);
    if (e == hipSuccess) e = hipMemset(p->epoch_dev, 0, EPOCH_BYTES);
    // plan-time fills run on the null stream; the plan's kernels run on the CALLER's stream, which may be a non-blocking
    // one (AMDGPU.jl's task streams, the threads-as-ranks test): settle the fill before anything can be launched over it
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return set_error(HPCLA_ERR_HIP, "window step counter: %s", hipGetErrorString(e));
    return HPCLA_OK;
}

static EpochRef epoch_ref(const hpcla_halo_plan *p, int64_t n_wait_readers)
{
    EpochRef r;
    r.done = p->epoch_dev;
    r.n_readers = (uint32_t)(p->push_blocks + n_wait_readers);
    return r;
}

constexpr int PUSH_THREADS = 256;
constexpr int64_t PUSH_CHUNK_DOUBLES = 4096;    // 32 KiB per workgroup
constexpr int PUSH_MAX_CHUNKS = 64;

// standalone producer (hpcla_halo_begin, packed path): the same per-workgroup code the fused SpMV launch
// runs in its leading workgroups (halo_wait.h)
template <typename I>
__global__ __launch_bounds__(PUSH_THREADS) void halo_push_kernel(PushArgs a)
{
    halo_push_block<I, PUSH_THREADS>(a, (int)blockIdx.x);
}

// standalone consumer (hpcla_halo_end, SpMM boundary blocks): kernels launched after it on the same
// stream start behind its acquire
// On an expired wait the ghost buffer is filled with NaN (failure path only; the single workgroup takes its
// time): the kernels behind it then compute NaN from it instead of a plausible result from stale rows.
__global__ __launch_bounds__(64) void halo_wait_kernel(HaloWait w, int64_t ghost_doubles)
{
    const uint32_t waited = halo_wait_block(w, w.first_wait_reader);
    if (waited & HALO_WAIT_TIMED_OUT) {
        double *g = const_cast<double *>(w.ghost0) + (int64_t)(waited & ~HALO_WAIT_TIMED_OUT) * w.buf_stride;
        for (int64_t i = threadIdx.x; i < ghost_doubles; i += 64) g[i] = halo_poison();
    }
}

// n_wait_readers: waiting workgroups of the exchange (boundary blocks of the fused launch, or 1 for the
// standalone wait kernel); together with the plan's push workgroups they are the exchange's epoch readers
HaloWait push_wait_args(const hpcla_halo_plan *p, int64_t n_wait_readers)
{
    HaloWait w;
    w.flags = p->flags;
    w.n_flags = (int)p->recv_ranks.size();
    w.er = epoch_ref(p, n_wait_readers);
    w.status = p->status;
    w.timeout_ticks = spin_timeout_ticks();
    w.ghost0 = p->ghost;
    w.buf_stride = (int64_t)(halo_buf_bytes((uint64_t)p->n_ghost, (uint32_t)p->width) / sizeof(double));
    w.nbuf = p->nbuf;
    w.first_wait_reader = (uint32_t)p->push_blocks;
    return w;
}

// Host view of "the ghost buffer of the exchange completed last".  Single-buffered plans (dense ghost rows):
// a constant.  Double-buffered plans: reads the device step counter, i.e. SYNCHRONISES the device -- only the
// API-parity path (execute_plan) asks; the fused SpMV computes its buffer in the kernel.
double *push_ghost_ptr(const hpcla_halo_plan *p)
{
    if (!p->win || p->nbuf < 2) return p->ghost;
    uint64_t done = 0;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(&done, p->epoch_dev, sizeof(done), hipMemcpyDeviceToHost) != hipSuccess)
        return p->ghost;
    const size_t buf = halo_buf_bytes((uint64_t)p->n_ghost, (uint32_t)p->width);
    return p->ghost + (done % 2) * (buf / sizeof(double));
}

// launch arguments of the plan's push workgroups (the caller launches them: halo_push_kernel, or the leading
// workgroups of the fused SpMV).  Nothing in them changes from step to step: the epoch is read from the plan's
// device-resident step counter, so the step is capturable into a HIP graph.
int push_begin(hpcla_halo_plan *p, const double *x, int64_t n_wait_readers, PushArgs *out)
{
    if (!p->attached) return set_error(HPCLA_ERR_INVALID, "halo push: plan has no attached peer windows");
    if (p->n_send_total > 0 && !x) return set_error(HPCLA_ERR_INVALID, "halo push: null x");
    out->x = x;
    out->idx = p->send_idx;
    out->targets = (const PushTarget *)p->push_desc_dev;
    out->map = (const int32_t *)p->push_block_map_dev;
    out->ack_out = (uint64_t *const *)p->ack_desc_dev;
    out->n_ack_out = (int)p->recv_ranks.size();
    out->arrive = p->arrive;
    out->status = p->status;
    out->er = epoch_ref(p, n_wait_readers);
    out->w = p->width;
    out->timeout_ticks = spin_timeout_ticks();
    out->n_blocks = (int)p->push_blocks;
    return HPCLA_OK;
}

int push_post(hpcla_halo_plan *p, const double *x, int64_t n_wait_readers, void *stream)
{
    PushArgs a;
    int rc = push_begin(p, x, n_wait_readers, &a);
    if (rc) return rc;
    if (p->idx_is_i64)
        halo_push_kernel<int64_t><<<(uint32_t)a.n_blocks, PUSH_THREADS, 0, as_stream(stream)>>>(a);
    else
        halo_push_kernel<int32_t><<<(uint32_t)a.n_blocks, PUSH_THREADS, 0, as_stream(stream)>>>(a);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

// the standalone consumer: ONE waiting workgroup (launched even without recv neighbours: it is one of the
// exchange's epoch readers)
int push_wait_kernel_launch(hpcla_halo_plan *p, void *stream)
{
    halo_wait_kernel<<<1, 64, 0, as_stream(stream)>>>(push_wait_args(p, 1), p->n_ghost * (int64_t)p->width);
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

// Stand-in for waiting workgroups that will never be launched (an error between push_post and the consuming launch):
// the exchange's epoch readers were counted when it was posted, so the missing ones are released here -- each waits
// for the flags like the real consumer would and releases the step counter -- and the plan stays in lockstep with its
// neighbours instead of hanging every later exchange.
__global__ __launch_bounds__(64) void halo_release_kernel(HaloWait w)
{
    (void)halo_wait_block(w, w.first_wait_reader + blockIdx.x);
}

int push_abandon_waiters(hpcla_halo_plan *p, int64_t n_wait_readers, void *stream)
{
    if (n_wait_readers <= 0) return HPCLA_OK;
    halo_release_kernel<<<(uint32_t)n_wait_readers, 64, 0, as_stream(stream)>>>(push_wait_args(p, n_wait_readers));
    HPCLA_CHECK_LAUNCH();
    return HPCLA_OK;
}

void push_free(hpcla_halo_plan *p)
{
    for (auto &m : p->peer_maps) window_close(&m);
    p->peer_maps.clear();
    if (p->push_desc_dev) (void)hipFree(p->push_desc_dev);
    if (p->ack_desc_dev) (void)hipFree(p->ack_desc_dev);
    if (p->push_block_map_dev) (void)hipFree(p->push_block_map_dev);
    if (p->arrive) (void)hipFree(p->arrive);
    if (p->epoch_dev) (void)hipFree(p->epoch_dev);
    p->push_desc_dev = p->ack_desc_dev = p->push_block_map_dev = nullptr;
    p->arrive = nullptr;
    p->epoch_dev = nullptr;
    if (p->win) {
        window_free(p->win);
        p->win = nullptr;
        p->ghost = nullptr;                    // lived inside the window
    }
    p->attached = false;
}

// ---- scalar all-reduce through the communicator window ---------------------------------------------
// slot (parity, rank): one 128-byte line {u64 epoch; double value[AR_MAX]} inside every rank's window.
// Each rank stores its partial into its slot of EVERY window, then sums the nranks slots of its own
// window in rank order: one kernel, one xGMI hop, and every rank adds the same numbers in the same
// order, so the result is bit-identical on all ranks (the reference asserts uniformity,
// test/test_utils.jl).
__global__ __launch_bounds__(64) void window_allreduce_kernel(uint64_t *const *__restrict__ peer_slots,
                                                              uint64_t *my_slots, uint32_t *status,
                                                              double *buf, int count, int op, int nranks,
                                                              int my_rank, uint64_t *done, int64_t timeout)
{
    __shared__ double s_val[64][AR_MAX];
    const int j = threadIdx.x;
    // the all-reduce counter lives in device memory (graph-capturable): this kernel is its only reader/writer
    const uint64_t epoch = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    const uint64_t parity = epoch & 1;
    if (j < nranks) {
        uint64_t *dst = peer_slots[j] + (parity * (uint64_t)nranks + (uint64_t)my_rank) * WIN_LINE_U64;
        for (int c = 0; c < count; ++c)
            __hip_atomic_store(reinterpret_cast<double *>(dst + 1 + c), buf[c], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __atomic_thread_fence(__ATOMIC_RELEASE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(dst, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t *src = my_slots + (parity * (uint64_t)nranks + (uint64_t)j) * WIN_LINE_U64;
        const bool arrived = spin_until_ge(src, epoch, (int64_t)wall_clock64(), timeout, status);
        // acquire between the epoch poll and the payload loads (system scope: the payload came over xGMI); the
        // loads below are relaxed atomics, which the fence orders behind the poll in the memory model too
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        for (int c = 0; c < count; ++c)
            s_val[j][c] = arrived ? __hip_atomic_load(reinterpret_cast<const double *>(src + 1 + c), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_SYSTEM)
                                  : __builtin_nan("");     // a partial that never arrived poisons the result
    }
    __syncthreads();
    if (j < count) {
        double acc = s_val[0][j];
        for (int r = 1; r < nranks; ++r)
            acc = op == 0 ? acc + s_val[r][j] : (op == 2 ? acc * s_val[r][j] : ((s_val[r][j] > acc || s_val[r][j] != s_val[r][j]) ? s_val[r][j] : acc));
        buf[j] = acc;
    }
    if (j == 0) __hip_atomic_store(done, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every lane has read it (barrier above)
}

int window_allreduce(hpcla_comm *comm, double *buf, int64_t count, int op, void *stream)
{
    hipStream_t s = as_stream(stream);
    uint64_t *my_slots = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(comm->win) + WIN_LINE);
    uint32_t *status = reinterpret_cast<uint32_t *>(comm->win);
    for (int64_t off = 0; off < count; off += AR_MAX) {
        const int n = (int)((count - off) < AR_MAX ? (count - off) : AR_MAX);
        window_allreduce_kernel<<<1, 64, 0, s>>>((uint64_t *const *)comm->peer_slots_dev, my_slots, status,
                                                 buf + off, n, op, comm->nranks, comm->rank, comm->ar_done_dev,
                                                 spin_timeout_ticks());
        HPCLA_CHECK_LAUNCH();
    }
    return HPCLA_OK;
}

void comm_window_free(hpcla_comm *comm)
{
    for (auto &m : comm->peers) window_close(&m);
    comm->peers.clear();
    if (comm->peer_slots_dev) (void)hipFree(comm->peer_slots_dev);
    comm->peer_slots_dev = nullptr;
    if (comm->ar_done_dev) (void)hipFree(comm->ar_done_dev);
    comm->ar_done_dev = nullptr;
    window_free(comm->win);
    comm->win = nullptr;
    comm->win_attached = false;
}

}  // namespace hpcla

using namespace hpcla;

// ---- C ABI: communicator window ---------------------------------------------------------------------
HPCLA_API int hpcla_comm_window_export(hpcla_comm_t *comm, uint8_t *desc_host)
{
    if (!comm || !desc_host) return set_error(HPCLA_ERR_INVALID, "comm_window_export: null pointer");
    if (comm->nranks > 64)
        return set_error(HPCLA_ERR_UNSUPPORTED, "comm_window_export: the window all-reduce supports <= 64 ranks");
    if (!comm->win) {
        comm->win_bytes = (size_t)WIN_LINE * (1 + 2 * (size_t)comm->nranks);
        int rc = window_alloc(&comm->win, comm->win_bytes, true);
        if (rc) return rc;
    }
    WindowDesc d;
    fill_desc(&d, comm->win, comm->win_bytes);
    if (d.bytes == 0) return set_error(HPCLA_ERR_HIP, "comm_window_export: hipIpcGetMemHandle failed");
    memcpy(desc_host, &d, sizeof(d));
    return HPCLA_OK;
}

HPCLA_API int hpcla_comm_window_attach(hpcla_comm_t *comm, const uint8_t *all_descs_host)
{
    if (!comm || !all_descs_host) return set_error(HPCLA_ERR_INVALID, "comm_window_attach: null pointer");
    if (!comm->win) return set_error(HPCLA_ERR_INVALID, "comm_window_attach: export first");
    if (comm->win_attached) return HPCLA_OK;
    const int n = comm->nranks;
    comm->peers.assign(n, PeerMap());
    std::vector<void *> slots(n);
    for (int r = 0; r < n; ++r) {
        WindowDesc d;
        memcpy(&d, all_descs_host + (size_t)r * sizeof(WindowDesc), sizeof(d));
        if (d.bytes != comm->win_bytes) {
            comm_window_free(comm);
            return set_error(HPCLA_ERR_INVALID, "comm_window_attach: rank %d exported %llu bytes, expected %zu", r,
                             (unsigned long long)d.bytes, comm->win_bytes);
        }
        int rc = window_open(d, r, comm->rank, comm->win, &comm->peers[r]);
        if (rc) { comm_window_free(comm); return rc; }
        slots[r] = reinterpret_cast<uint8_t *>(comm->peers[r].base) + WIN_LINE;
    }
    hipError_t e = hipMalloc((void **)&comm->peer_slots_dev, sizeof(void *) * n);
    if (e == hipSuccess) e = hipMemcpy(comm->peer_slots_dev, slots.data(), sizeof(void *) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&comm->ar_done_dev, sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemset(comm->ar_done_dev, 0, sizeof(uint64_t));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);      // null-stream fill, non-blocking caller streams (see push_plan_alloc)
    if (e != hipSuccess) {
        comm_window_free(comm);
        return set_error(HPCLA_ERR_HIP, "comm_window_attach: %s", hipGetErrorString(e));
    }
    comm->win_attached = true;
    return HPCLA_OK;
}

__global__ void selftest_set_kernel(double *p, double v) { *p = v; }

// Connection test, run once after attach: an all-reduce of (rank + 1) through the windows with its own
// (short) timeout.  *ok = 1 iff every peer's store arrived here and the sum is n(n+1)/2.  The host layer
// all-gathers `ok` and, unless every rank passed, detaches the windows everywhere and stays on RCCL.
HPCLA_API int hpcla_comm_window_selftest(hpcla_comm_t *comm, double timeout_s, int *ok)
{
    if (!comm || !ok) return set_error(HPCLA_ERR_INVALID, "comm_window_selftest: null pointer");
    *ok = 0;
    if (!comm->win_attached) return set_error(HPCLA_ERR_INVALID, "comm_window_selftest: no attached window");
    double *buf = nullptr;
    HPCLA_CHECK_HIP(hipMalloc((void **)&buf, sizeof(double)));
    const double mine = (double)(comm->rank + 1);
    // a stream of its own, not the null stream: ranks hosted by ONE process (a host thread each) would otherwise queue
    // their test kernels behind each other on the process's null stream, the first spinning for partials the later ones
    // cannot deliver
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    // The operand goes in as a kernel argument and the results come back behind the stream's synchronisation: nothing is
    // parked behind the waiting kernel on behalf of the host (collective hygiene for thread-ranks of one process: comm.hip,
    // hpcla_halo_plan_probe)
    double got = 0.0;
    uint32_t st = 1;
    if (e == hipSuccess) {
        selftest_set_kernel<<<1, 1, 0, s>>>(buf, mine);
        uint64_t *my_slots = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(comm->win) + WIN_LINE);
        window_allreduce_kernel<<<1, 64, 0, s>>>((uint64_t *const *)comm->peer_slots_dev, my_slots,
                                                 reinterpret_cast<uint32_t *>(comm->win), buf, 1, 0,
                                                 comm->nranks, comm->rank, comm->ar_done_dev,
                                                 (int64_t)((timeout_s > 0 ? timeout_s : 5.0) * 1.0e8));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipMemcpy(&got, buf, sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&st, comm->win, sizeof(st), hipMemcpyDeviceToHost);
    }
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(buf);
    if (e != hipSuccess) return set_error(HPCLA_ERR_HIP, "comm_window_selftest: %s", hipGetErrorString(e));
    const double want = 0.5 * comm->nranks * (comm->nranks + 1.0);
    *ok = (st == 0 && got == want) ? 1 : 0;
    return HPCLA_OK;
}

// give up the communicator window (after a failed connection test): back to RCCL for everything
HPCLA_API int hpcla_comm_window_detach(hpcla_comm_t *comm)
{
    if (!comm) return set_error(HPCLA_ERR_INVALID, "comm_window_detach: null communicator");
    comm_window_free(comm);
    return HPCLA_OK;
}

// give up the push transport of one plan (a neighbour could not be mapped): the ghost segment stays where
// it is and RCCL receives into it
HPCLA_API int hpcla_halo_plan_detach(hpcla_halo_plan_t *plan)
{
    if (!plan) return set_error(HPCLA_ERR_INVALID, "halo_plan_detach: null plan");
    for (auto &m : plan->peer_maps) window_close(&m);
    plan->peer_maps.clear();
    plan->attached = false;
    return HPCLA_OK;
}

HPCLA_API int hpcla_comm_status(hpcla_comm_t *comm, int *timed_out)
{
    if (!comm || !timed_out) return set_error(HPCLA_ERR_INVALID, "comm_status: null pointer");
    *timed_out = 0;
    if (!comm->win) return HPCLA_OK;
    uint32_t v = 0;
    HPCLA_CHECK_HIP(hipMemcpy(&v, comm->win, sizeof(v), hipMemcpyDeviceToHost));
    *timed_out = v ? 1 : 0;
    return HPCLA_OK;
}

// ---- C ABI: halo plan window ------------------------------------------------------------------------
HPCLA_API int hpcla_halo_plan_export(hpcla_halo_plan_t *plan, uint8_t *desc_host, int64_t *table_host)
{
    if (!plan || !desc_host || !table_host) return set_error(HPCLA_ERR_INVALID, "halo_plan_export: null pointer");
    const int n = plan->comm->nranks;
    for (int i = 0; i < HPCLA_WINDOW_TABLE_ROWS * n; ++i) table_host[i] = -1;
    WindowDesc d;
    fill_desc(&d, plan->win, plan->win ? plan->win_bytes : 0);
    d.n_ghost = (uint64_t)plan->n_ghost;
    d.nbuf = (uint32_t)plan->nbuf;
    d.width = (uint32_t)plan->width;
    d.n_flags = (uint32_t)plan->recv_ranks.size();
    d.n_acks = (uint32_t)plan->send_ranks.size();
    memcpy(desc_host, &d, sizeof(d));
    if (!plan->win) return HPCLA_OK;           // RCCL-only plan: exports an empty descriptor
    for (size_t j = 0; j < plan->recv_ranks.size(); ++j) {
        const int r = plan->recv_ranks[j];
        table_host[0 * n + r] = (int64_t)j;                    // flag line of source rank r
        table_host[1 * n + r] = plan->recv_off[j];             // where r's segment starts in my ghost
        table_host[3 * n + r] = plan->recv_counts[j];
    }
    for (size_t i = 0; i < plan->send_ranks.size(); ++i)
        table_host[2 * n + plan->send_ranks[i]] = (int64_t)i;  // ack line of consumer rank
    return HPCLA_OK;
}

HPCLA_API int hpcla_halo_plan_attach(hpcla_halo_plan_t *plan, const uint8_t *all_descs_host,
                                     const int64_t *all_tables_host)
{
    if (!plan || !all_descs_host || !all_tables_host)
        return set_error(HPCLA_ERR_INVALID, "halo_plan_attach: null pointer");
    if (!plan->win) return set_error(HPCLA_ERR_INVALID, "halo_plan_attach: the plan was created without a window");
    if (plan->attached) return HPCLA_OK;
    hpcla_halo_plan *p = plan;
    const int n = p->comm->nranks, me = p->comm->rank;
    const size_t ns = p->send_ranks.size(), nr = p->recv_ranks.size();
    auto desc_of = [&](int r) {
        WindowDesc d;
        memcpy(&d, all_descs_host + (size_t)r * sizeof(WindowDesc), sizeof(d));
        return d;
    };
    auto table_of = [&](int r) { return all_tables_host + (size_t)r * HPCLA_WINDOW_TABLE_ROWS * n; };
    // one mapping per distinct peer rank (a rank may be both a send and a recv neighbour)
    std::vector<PeerMap> by_rank(n);
    std::vector<char> have(n, 0);
    auto map_rank = [&](int r) -> int {
        if (have[r]) return HPCLA_OK;
        int rc = window_open(desc_of(r), r, me, p->win, &by_rank[r]);
        if (rc == HPCLA_OK) have[r] = 1;
        return rc;
    };
    auto fail = [&](int rc) {
        for (int r = 0; r < n; ++r)
            if (have[r]) window_close(&by_rank[r]);
        return rc;
    };
    std::vector<PushTarget> targets(ns);
    std::vector<int32_t> bmap;
    for (size_t i = 0; i < ns; ++i) {
        const int q = p->send_ranks[i];
        int rc = map_rank(q);
        if (rc) return fail(rc);
        const WindowDesc d = desc_of(q);
        const int64_t *T = table_of(q);
        const int64_t fs = T[0 * n + me], go = T[1 * n + me], cnt = T[3 * n + me];
        if (fs < 0 || go < 0 || cnt != p->send_counts[i] || (int)d.width != p->width)
            return fail(set_error(HPCLA_ERR_INVALID,
                                  "halo_plan_attach: rank %d does not expect %lld entries from rank %d (has %lld)", q,
                                  (long long)p->send_counts[i], me, (long long)cnt));
        uint8_t *base = reinterpret_cast<uint8_t *>(by_rank[q].base);
        const size_t ctrl = halo_ctrl_bytes(d.n_flags, d.n_acks);
        const size_t buf = halo_buf_bytes(d.n_ghost, d.width);
        PushTarget &t = targets[i];
        t.ghost = reinterpret_cast<double *>(base + ctrl) + go * p->width;
        t.buf_stride = (int64_t)(buf / sizeof(double));
        t.flag = reinterpret_cast<uint64_t *>(base + (size_t)fs * WIN_LINE);
        t.ack = p->acks + i * WIN_LINE_U64;
        t.count = p->send_counts[i];
        t.src_off = p->send_off[i];
        t.first = p->send_contig[i] ? p->send_first[i] : -1;
        t.nbuf = (int32_t)d.nbuf;
        int64_t ch = (t.count * p->width + PUSH_CHUNK_DOUBLES - 1) / PUSH_CHUNK_DOUBLES;
        if (ch < 1) ch = 1;
        if (ch > PUSH_MAX_CHUNKS) ch = PUSH_MAX_CHUNKS;
        t.nchunks = (int32_t)ch;
        for (int32_t c = 0; c < t.nchunks; ++c) { bmap.push_back((int32_t)i); bmap.push_back(c); }
    }
    if (bmap.empty()) { bmap.push_back(-1); bmap.push_back(0); }   // acks only
    std::vector<uint64_t *> ack_out(nr);
    for (size_t j = 0; j < nr; ++j) {
        const int r = p->recv_ranks[j];
        int rc = map_rank(r);
        if (rc) return fail(rc);
        const WindowDesc d = desc_of(r);
        const int64_t as = table_of(r)[2 * n + me];
        if (as < 0)
            return fail(set_error(HPCLA_ERR_INVALID, "halo_plan_attach: rank %d does not send to rank %d", r, me));
        ack_out[j] = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(by_rank[r].base) +
                                                  ((size_t)d.n_flags + (size_t)as) * WIN_LINE);
    }
#define ATT_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            push_free_device_tables(p);                                                            \
            return fail(set_error(HPCLA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)));  \
        }                                                                                          \
    } while (0)
    auto push_free_device_tables = [](hpcla_halo_plan *q) {
        if (q->push_desc_dev) (void)hipFree(q->push_desc_dev);
        if (q->ack_desc_dev) (void)hipFree(q->ack_desc_dev);
        if (q->push_block_map_dev) (void)hipFree(q->push_block_map_dev);
        if (q->arrive) (void)hipFree(q->arrive);
        q->push_desc_dev = q->ack_desc_dev = q->push_block_map_dev = nullptr;
        q->arrive = nullptr;
    };
    ATT_HIP(hipMalloc(&p->push_desc_dev, sizeof(PushTarget) * (ns ? ns : 1)));
    if (ns) ATT_HIP(hipMemcpy(p->push_desc_dev, targets.data(), sizeof(PushTarget) * ns, hipMemcpyHostToDevice));
    ATT_HIP(hipMalloc(&p->ack_desc_dev, sizeof(uint64_t *) * (nr ? nr : 1)));
    if (nr) ATT_HIP(hipMemcpy(p->ack_desc_dev, ack_out.data(), sizeof(uint64_t *) * nr, hipMemcpyHostToDevice));
    ATT_HIP(hipMalloc(&p->push_block_map_dev, sizeof(int32_t) * bmap.size()));
    ATT_HIP(hipMemcpy(p->push_block_map_dev, bmap.data(), sizeof(int32_t) * bmap.size(), hipMemcpyHostToDevice));
    ATT_HIP(hipMalloc((void **)&p->arrive, sizeof(uint64_t) * (ns ? ns : 1)));
    ATT_HIP(hipMemset(p->arrive, 0, sizeof(uint64_t) * (ns ? ns : 1)));
    ATT_HIP(hipStreamSynchronize(nullptr));                      // null-stream fill, non-blocking caller streams (see push_plan_alloc)
#undef ATT_HIP
    p->push_blocks = (int64_t)(bmap.size() / 2);
    // keep the mappings (closed at destroy)
    for (int r = 0; r < n; ++r)
        if (have[r]) p->peer_maps.push_back(by_rank[r]);
    p->attached = true;
    return HPCLA_OK;
}

HPCLA_API int hpcla_halo_status(hpcla_halo_plan_t *plan, int *timed_out)
{
    if (!plan || !timed_out) return set_error(HPCLA_ERR_INVALID, "halo_status: null pointer");
    *timed_out = 0;
    if (!plan->status) return HPCLA_OK;
    uint32_t v = 0;
    HPCLA_CHECK_HIP(hipMemcpy(&v, plan->status, sizeof(v), hipMemcpyDeviceToHost));
    *timed_out = v ? 1 : 0;
    return HPCLA_OK;
}
