// comm_internal.h -- communicator / halo-plan objects shared by comm.hip (RCCL transport) and
// window.hip (peer-window push transport over xGMI).  Library-internal.
#pragma once
#include <rccl/rccl.h>

#include <memory>
#include <vector>

#include "common.h"
#include "halo_wait.h"

// ---- peer window --------------------------------------------------------------------------------
// A window is ONE fine-grained device allocation of the owning rank that every peer of the node maps
// with hipIpcOpenMemHandle and WRITES into directly over xGMI (stores with system scope).  It starts
// with a control block of 128-byte lines (one polled word per line), followed by the payload.
//
//   halo plan window:   [flag line per recv neighbour][ack line per send neighbour][status line]
//                       [ghost buffer 0][ghost buffer 1 (only when double-buffered)]
//   communicator window:[status line][2 parities x nranks slots x 128 B: {epoch, value[0..AR_MAX)}]
//
// flag[i]  (written by recv neighbour i's push): epoch of the payload it has completed in my ghost;
// ack[i]   (written by send neighbour i, the consumer of what I push): last epoch it has finished reading.
namespace hpcla {

constexpr int WIN_LINE = 128;                  // bytes per control line
constexpr int WIN_LINE_U64 = WIN_LINE / 8;
static_assert(WIN_LINE_U64 == WIN_FLAG_STRIDE_U64, "flag stride");
constexpr int AR_MAX = 8;                      // doubles per window all-reduce (larger counts: RCCL)
constexpr int64_t WIN_DOUBLE_BUFFER_MAX = 64ll << 20;   // ghost bytes up to which two buffers are kept

struct WindowDesc {                            // HPCLA_WINDOW_DESC_BYTES, exchanged by the host runtime
    uint8_t ipc[64];                           // hipIpcMemHandle_t
    uint64_t host_id;                          // hash of the node identity: IPC needs one node
    uint64_t pid;
    uint64_t bytes;                            // size of the allocation (0 = this rank exports nothing)
    uint64_t n_ghost;                          // halo: ghost entries per buffer (indices, not doubles)
    uint32_t nbuf;                             // halo: 1 or 2 ghost buffers
    uint32_t width;
    uint32_t n_flags, n_acks;
    uint64_t device_id;                        // identity of the GPU holding the window (hpcla_device_identity): peer-access check
    uint8_t pad[8];
};
static_assert(sizeof(WindowDesc) == HPCLA_WINDOW_DESC_BYTES, "WindowDesc size");

struct PeerMap {                               // a peer's window mapped into this process
    void *base = nullptr;                      // peer's allocation base in MY address space
    bool opened = false;                       // hipIpcOpenMemHandle'd (false: same process, local pointer)
    bool local_ref = false;                    // holds a reference on another in-process rank's window (window_close releases it)
};

uint64_t host_identity();
size_t halo_window_buf_bytes(uint64_t n_ghost, uint32_t width);
int window_alloc(void **p, size_t bytes, bool uncached);
int window_open(const WindowDesc &d, int peer_rank, int my_rank, void *my_base, PeerMap *out);
void window_close(PeerMap *m);
void window_free(void *win);                   // hipFree + forget the process-local registration
int64_t spin_timeout_ticks();                  // HPCLA_PUSH_TIMEOUT_S (default 300 s) in wall_clock64 ticks

}  // namespace hpcla

struct hpcla_comm {
    int nranks = 1;
    int rank = 0;
    ncclComm_t nccl = nullptr;                 // null for the serial communicator
    // communicator window (scalar all-reduce by direct peer writes); absent until attach
    void *win = nullptr;
    size_t win_bytes = 0;
    std::vector<hpcla::PeerMap> peers;         // [nranks]
    void **peer_slots_dev = nullptr;           // device array [nranks]: base of every rank's slot area
    bool win_attached = false;
    uint64_t *ar_done_dev = nullptr;           // device: window all-reduces completed (read + bumped by the kernel)
};

namespace hpcla {
struct SideStream {
    hipStream_t s = nullptr;
    ~SideStream() { if (s) (void)hipStreamDestroy(s); }
};
}  // namespace hpcla

struct hpcla_halo_plan {
    hpcla_comm *comm = nullptr;
    int width = 1;
    std::vector<int> send_ranks, recv_ranks;
    std::vector<int64_t> send_counts, recv_counts, send_off, recv_off;
    std::vector<int64_t> send_first;           // first index of neighbour i when its run is contiguous
    std::vector<char> send_contig;
    bool need_pack = false;
    int64_t n_send_total = 0, n_ghost = 0;
    void *send_idx = nullptr;                  // device copy of the concatenated send indices
    int idx_is_i64 = 0;
    double *send_buf = nullptr;                // device, n_send_total * width (RCCL transport only)
    double *ghost = nullptr;                   // device, n_ghost * width (buffer 0)
    hipStream_t side = nullptr;                // == side_owner->s
    // The exchange stream is SHARED by the plans of a chain (hpcla_halo_plan_chain) and goes with the LAST of them,
    // whatever order they are destroyed in (round 3 let a follower borrow the leader's stream without tracking it:
    // destroying the leader first left the followers with a dangling hipStream_t).
    std::shared_ptr<hpcla::SideStream> side_owner;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    // contiguity of the caller's (ascending) interior block list, probed once per list
    const int32_t *probed_list = nullptr;
    int64_t probed_n = -1, probed_first = -1;
    bool probed_contig = false;

    // ---- peer-window transport (window.hip) ----
    void *win = nullptr;                       // this rank's window (ghost lives inside it) or null
    size_t win_bytes = 0;
    int nbuf = 1;                              // ghost buffers in the window
    bool single_buffer = false;                // HPCLA_HALO_SINGLE_BUFFER: never double-buffer (begin/end consumers)
    uint64_t *flags = nullptr, *acks = nullptr;      // local control lines (stride WIN_LINE_U64)
    uint32_t *status = nullptr;                // local: nonzero after a spin timed out
    std::vector<hpcla::PeerMap> peer_maps;     // mapped windows of my neighbours (one per distinct rank)
    void *push_desc_dev = nullptr;             // device array of PushTarget, one per send neighbour
    void *ack_desc_dev = nullptr;              // device array of uint64_t* : where my acks go, one per recv neighbour
    int64_t push_blocks = 0;                   // grid of the push kernel
    void *push_block_map_dev = nullptr;        // device int32[push_blocks][2]: (neighbour, chunk)
    uint64_t *arrive = nullptr;                // device, local: arrival counters, one per send neighbour
    bool attached = false;
    uint64_t *epoch_dev = nullptr;             // device {done, top, shard counters}: the plan's step counter (halo_wait.h)
    // scratch of hpcla_halo_plan_probe, released with the plan: freeing inside the probe would be a device-wide
    // synchronisation in the middle of a collective (comm.hip, hpcla_halo_plan_probe)
    std::vector<void *> probe_scratch;
};

namespace hpcla {

// RCCL entry points, resolved at first use (comm.hip)
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

enum HaloMode { HALO_SERIAL = 0, HALO_OVERLAP = 1, HALO_PUSH = 2 };
// mode of one plan: HPCLA_HALO_MODE = serial | overlap | push; unset = push when the plan's window is
// attached, else serial
HaloMode halo_mode_of(const hpcla_halo_plan *plan);
bool halo_want_window();                       // false when HPCLA_HALO_MODE names an RCCL mode

// window.hip
// n_wait_readers = waiting workgroups of the exchange (boundary blocks of a fused launch; 1 for the wait kernel)
int push_begin(hpcla_halo_plan *plan, const double *x, int64_t n_wait_readers, PushArgs *out);   // launch args
int push_post(hpcla_halo_plan *plan, const double *x, int64_t n_wait_readers, void *stream);     // own kernel
int push_wait_kernel_launch(hpcla_halo_plan *plan, void *stream);          // standalone wait (halo_end)
// after push_post(plan, x, n, ...) whose n consuming workgroups cannot be launched: release them (keeps the epochs in step)
int push_abandon_waiters(hpcla_halo_plan *plan, int64_t n_wait_readers, void *stream);
double *push_ghost_ptr(const hpcla_halo_plan *plan);                       // host view (may synchronise)
HaloWait push_wait_args(const hpcla_halo_plan *plan, int64_t n_wait_readers);
void push_free(hpcla_halo_plan *plan);
int push_plan_alloc(hpcla_halo_plan *plan);                                // window instead of a plain ghost
int window_allreduce(hpcla_comm *comm, double *buf, int64_t count, int op, void *stream);
void comm_window_free(hpcla_comm *comm);

}  // namespace hpcla
