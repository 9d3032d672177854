// comm.hip -- RCCL communicator, GPU-resident halo exchange and the fused distributed SpMV.
//
// Replaces the execute half of the reference's VectorPlan (execute_plan!, src/vectors.jl:394-463):
// there every SpMV copies x to the host, packs send buffers on the CPU, runs MPI Isend/Irecv on host
// memory, unpacks and copies the gathered vector back to the device, strictly before the kernel.
// Here the exchange never leaves the GPUs: a pack kernel (skipped when a neighbour's indices are one
// contiguous run, as for stencil slabs) and one ncclGroup of ncclSend/ncclRecv over xGMI; receives
// land directly in the ghost segment the row blocks read (no unpack, no local copy).  The group runs
// either on the caller's stream ahead of one fused launch, or on the plan's side stream while the
// interior row blocks are computed on the caller's stream (spmv_dist_impl has the measurements that
// decide the default); the SpMM path and hpcla_halo_begin/end always use the side stream.
//
// librccl is loaded lazily with dlopen so the single-GPU path has no RCCL dependency; inside a
// PyTorch process the already-loaded librccl.so.1 is reused (same SONAME).
//
// Second transport (window.hip): on one node the halo and the scalar all-reduce can bypass RCCL
// altogether -- peers map each other's ghost windows and push into them directly over xGMI, and the
// SpMV's own boundary workgroups wait for the data (HPCLA_HALO_MODE=push, the default once a plan's
// windows are attached).  This file dispatches between the two.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "comm_internal.h"

namespace hpcla {

int spmv_split_i32(const int32_t *, const int32_t *, const double *, const double *, const double *,
                   int64_t, double *, int64_t, int64_t, int, const int32_t *, int64_t, void *,
                   double *, int64_t);
int spmv_split_i64(const int64_t *, const int64_t *, const double *, const double *, const double *,
                   int64_t, double *, int64_t, int64_t, int, const int32_t *, int64_t, void *,
                   double *, int64_t);
int spmv_fused_i32(const int32_t *, const int32_t *, const double *, const double *, const double *, int64_t,
                   double *, int64_t, int64_t, int, const int32_t *, int64_t, int64_t, const int32_t *, int64_t,
                   const HaloWait &, const PushArgs &, void *, double *);
int spmv_fused_i64(const int64_t *, const int64_t *, const double *, const double *, const double *, int64_t,
                   double *, int64_t, int64_t, int, const int32_t *, int64_t, int64_t, const int32_t *, int64_t,
                   const HaloWait &, const PushArgs &, void *, double *);
int reduce_partials_sum(const double *partial, int64_t np, double *scratch, double *out,
                        void *stream);   // vecops.hip

// ---- RCCL entry points, resolved at first use (RcclApi: comm_internal.h) ----------------------
static RcclApi g_rccl;

static int rccl_load()
{
    if (g_rccl.handle) return HPCLA_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return set_error(HPCLA_ERR_RCCL, "cannot dlopen librccl: %s", dlerror());
#define HPCLA_SYM(field, name)                                                                    \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                      \
    if (!g_rccl.field) return set_error(HPCLA_ERR_RCCL, "librccl lacks symbol %s", name);
    HPCLA_SYM(GetUniqueId, "ncclGetUniqueId")
    HPCLA_SYM(CommInitRank, "ncclCommInitRank")
    HPCLA_SYM(CommDestroy, "ncclCommDestroy")
    HPCLA_SYM(Send, "ncclSend")
    HPCLA_SYM(Recv, "ncclRecv")
    HPCLA_SYM(GroupStart, "ncclGroupStart")
    HPCLA_SYM(GroupEnd, "ncclGroupEnd")
    HPCLA_SYM(AllReduce, "ncclAllReduce")
    HPCLA_SYM(GetErrorString, "ncclGetErrorString")
#undef HPCLA_SYM
    g_rccl.handle = h;
    return HPCLA_OK;
}

#define HPCLA_CHECK_RCCL(expr)                                                                    \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess)                                                                    \
            return hpcla::set_error(HPCLA_ERR_RCCL, "%s failed: %s (%s:%d)", #expr,               \
                                    g_rccl.GetErrorString(_r), __FILE__, __LINE__);               \
    } while (0)

}  // namespace hpcla

namespace hpcla {

// pack: buf[(off+i)*w + c] = x[idx[off+i]*w + c]
template <typename I>
__global__ __launch_bounds__(256) void pack_kernel(const double *__restrict__ x,
                                                   const I *__restrict__ idx,
                                                   double *__restrict__ buf, int64_t n, int w)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t total = n * w;
    for (; t < total; t += stride) {
        const int64_t i = t / w;
        const int c = (int)(t - i * w);
        buf[t] = x[(int64_t)idx[i] * w + c];
    }
}

int allreduce_on(hpcla_comm_t *comm, double *buf, int64_t count, int op, void *stream)
{
    if (!comm) return set_error(HPCLA_ERR_INVALID, "allreduce: null communicator");
    if (count < 0) return set_error(HPCLA_ERR_INVALID, "allreduce: negative count");
    if (op < 0 || op > 2) return set_error(HPCLA_ERR_INVALID, "allreduce: op must be 0 (sum), 1 (max) or 2 (product)");
    if (count == 0) return HPCLA_OK;
    if (comm->nranks == 1 && !comm->nccl) return HPCLA_OK;
    if (!buf) return set_error(HPCLA_ERR_INVALID, "allreduce: null buffer");
    // scalars (dot / norm / CG's rr): one-hop peer-window kernel, identical bits on every rank;
    // long vectors (the dense transposed mat-vec): RCCL's ring, unless this communicator has none
    if (comm->win_attached && (count <= AR_MAX || !comm->nccl)) return window_allreduce(comm, buf, count, op, stream);
    if (!comm->nccl)
        return set_error(HPCLA_ERR_INVALID, "allreduce: communicator has neither RCCL nor an attached window");
    HPCLA_CHECK_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble,
                                      op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclProd), comm->nccl, as_stream(stream)));
    return HPCLA_OK;
}

bool halo_active(const hpcla_halo_plan_t *plan)
{
    return plan && !(plan->send_ranks.empty() && plan->recv_ranks.empty());
}

// HPCLA_HALO_MODE: "push" | "serial" | "overlap"; unset = push where the plan's peer windows are
// attached (one node), else serial -- see spmv_dist_impl
static int g_halo_mode_override = -2;          // -2: not set -> environment
static int halo_mode_env()
{
    static const int m = [] {
        const char *e = getenv("HPCLA_HALO_MODE");
        if (!e || !e[0]) return -1;
        if (e[0] == 'o') return (int)HALO_OVERLAP;
        if (e[0] == 'p') return (int)HALO_PUSH;
        return (int)HALO_SERIAL;
    }();
    return g_halo_mode_override != -2 ? g_halo_mode_override : m;
}

HaloMode halo_mode_of(const hpcla_halo_plan *plan)
{
    const int m = halo_mode_env();
    if (plan && plan->attached && (m < 0 || m == (int)HALO_PUSH)) return HALO_PUSH;
    if (plan && plan->comm && !plan->comm->nccl && plan->attached) return HALO_PUSH;   // no RCCL to fall back on
    return m == (int)HALO_OVERLAP ? HALO_OVERLAP : HALO_SERIAL;
}

bool halo_want_window()
{
    const int m = halo_mode_env();
    return m < 0 || m == (int)HALO_PUSH;
}

}  // namespace hpcla

using namespace hpcla;

HPCLA_API int hpcla_set_halo_mode(int mode)
{
    if (mode < -1 || mode > 2) return set_error(HPCLA_ERR_INVALID, "set_halo_mode: mode must be -1 (auto), 0, 1 or 2");
    g_halo_mode_override = mode;
    return HPCLA_OK;
}

HPCLA_API int hpcla_comm_get_unique_id(uint8_t *id_host)
{
    if (!id_host) return set_error(HPCLA_ERR_INVALID, "get_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == HPCLA_UNIQUE_ID_BYTES, "ncclUniqueId size");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    HPCLA_CHECK_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_host, &id, sizeof(id));
    return HPCLA_OK;
}

HPCLA_API int hpcla_comm_init_rank(hpcla_comm_t **comm, const uint8_t *id_host, int nranks,
                                   int rank)
{
    return hpcla_comm_init_rank_ex(comm, id_host, nranks, rank, 0);
}

HPCLA_API int hpcla_comm_init_rank_ex(hpcla_comm_t **comm, const uint8_t *id_host, int nranks,
                                      int rank, int flags)
{
    if (!comm) return set_error(HPCLA_ERR_INVALID, "comm_init_rank: null output");
    if (nranks < 1 || rank < 0 || rank >= nranks)
        return set_error(HPCLA_ERR_INVALID, "comm_init_rank: bad rank %d of %d", rank, nranks);
    hpcla_comm *c = new (std::nothrow) hpcla_comm();
    if (!c) return set_error(HPCLA_ERR_ALLOC, "comm_init_rank: out of memory");
    c->nranks = nranks;
    c->rank = rank;
    // nranks == 1: serial communicator (CommSerial, src/backends.jl:63).  HPCLA_FORCE_RCCL=1 makes
    // a real one-rank RCCL communicator, used by the self-send test of the exchange code.
    const char *force = getenv("HPCLA_FORCE_RCCL");
    if ((flags & HPCLA_COMM_NO_RCCL) == 0 && (nranks > 1 || (force && force[0] == '1'))) {
        if (!id_host) { delete c; return set_error(HPCLA_ERR_INVALID, "comm_init_rank: null id"); }
        int rc = rccl_load();
        if (rc) { delete c; return rc; }
        ncclUniqueId id;
        memcpy(&id, id_host, sizeof(id));
        ncclResult_t r = g_rccl.CommInitRank(&c->nccl, nranks, id, rank);
        if (r != ncclSuccess) {
            delete c;
            return set_error(HPCLA_ERR_RCCL, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        }
    }
    *comm = c;
    return HPCLA_OK;
}

HPCLA_API int hpcla_comm_rank(const hpcla_comm_t *comm, int *rank)
{
    if (!comm || !rank) return set_error(HPCLA_ERR_INVALID, "comm_rank: null pointer");
    *rank = comm->rank;
    return HPCLA_OK;
}

HPCLA_API int hpcla_comm_size(const hpcla_comm_t *comm, int *nranks)
{
    if (!comm || !nranks) return set_error(HPCLA_ERR_INVALID, "comm_size: null pointer");
    *nranks = comm->nranks;
    return HPCLA_OK;
}

HPCLA_API int hpcla_comm_destroy(hpcla_comm_t *comm)
{
    if (!comm) return HPCLA_OK;
    comm_window_free(comm);
    if (comm->nccl) {
        ncclResult_t r = g_rccl.CommDestroy(comm->nccl);
        comm->nccl = nullptr;
        if (r != ncclSuccess) {
            delete comm;
            return set_error(HPCLA_ERR_RCCL, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
        }
    }
    delete comm;
    return HPCLA_OK;
}

HPCLA_API int hpcla_allreduce_f64(hpcla_comm_t *comm, double *buf, int64_t count, int op,
                                  void *stream)
{
    return allreduce_on(comm, buf, count, op, stream);
}

// ---- contiguous-range exchange (repartition plans) -----------------------------------------------
// Device form of execute_plan!(::VectorRepartitionPlan) (src/vectors.jl:624-671) and of the dense /
// sparse-values repartitions (src/dense.jl:1711-1760, src/sparse.jl:4443-4535): every message is a
// contiguous range of the source, every receive lands at a fixed offset of the result, so there is
// nothing to pack or unpack -- one ncclGroup between the callers' buffers plus one device copy for
// the part that stays.  Offsets and counts are in units of `width` doubles (1 for vectors / nzval,
// ncols for row-major dense rows).
HPCLA_API int hpcla_exchange_ranges_f64(hpcla_comm_t *comm, const double *src, double *dst,
                                        int n_send, const int *send_ranks,
                                        const int64_t *send_offsets, const int64_t *send_counts,
                                        int n_recv, const int *recv_ranks,
                                        const int64_t *recv_offsets, const int64_t *recv_counts,
                                        int64_t local_src_offset, int64_t local_dst_offset,
                                        int64_t local_count, int width, void *stream)
{
    if (!comm) return set_error(HPCLA_ERR_INVALID, "exchange_ranges: null communicator");
    if (n_send < 0 || n_recv < 0 || local_count < 0 || width < 1)
        return set_error(HPCLA_ERR_INVALID, "exchange_ranges: bad counts/width");
    if ((n_send > 0 && (!send_ranks || !send_offsets || !send_counts)) ||
        (n_recv > 0 && (!recv_ranks || !recv_offsets || !recv_counts)))
        return set_error(HPCLA_ERR_INVALID, "exchange_ranges: null list");
    for (int i = 0; i < n_send; ++i)
        if (send_ranks[i] < 0 || send_ranks[i] >= comm->nranks || send_offsets[i] < 0 || send_counts[i] < 0)
            return set_error(HPCLA_ERR_INVALID, "exchange_ranges: bad send entry %d", i);
    for (int i = 0; i < n_recv; ++i)
        if (recv_ranks[i] < 0 || recv_ranks[i] >= comm->nranks || recv_offsets[i] < 0 || recv_counts[i] < 0)
            return set_error(HPCLA_ERR_INVALID, "exchange_ranges: bad recv entry %d", i);
    if ((n_send > 0 || local_count > 0) && !src)
        return set_error(HPCLA_ERR_INVALID, "exchange_ranges: null source");
    if ((n_recv > 0 || local_count > 0) && !dst)
        return set_error(HPCLA_ERR_INVALID, "exchange_ranges: null destination");
    hipStream_t s = as_stream(stream);
    if (local_count > 0)
        HPCLA_CHECK_HIP(hipMemcpyAsync(dst + local_dst_offset * width, src + local_src_offset * width,
                                       (size_t)local_count * width * sizeof(double),
                                       hipMemcpyDeviceToDevice, s));
    if (n_send == 0 && n_recv == 0) return HPCLA_OK;
    if (!comm->nccl)
        return set_error(HPCLA_ERR_INVALID, "exchange_ranges: messages on a serial communicator");
    HPCLA_CHECK_RCCL(g_rccl.GroupStart());
    for (int i = 0; i < n_recv; ++i)
        if (recv_counts[i] > 0) {
            ncclResult_t r = g_rccl.Recv(dst + recv_offsets[i] * width, (size_t)recv_counts[i] * width,
                                         ncclDouble, recv_ranks[i], comm->nccl, s);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return set_error(HPCLA_ERR_RCCL, "ncclRecv failed: %s", g_rccl.GetErrorString(r)); }
        }
    for (int i = 0; i < n_send; ++i)
        if (send_counts[i] > 0) {
            ncclResult_t r = g_rccl.Send(src + send_offsets[i] * width, (size_t)send_counts[i] * width,
                                         ncclDouble, send_ranks[i], comm->nccl, s);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return set_error(HPCLA_ERR_RCCL, "ncclSend failed: %s", g_rccl.GetErrorString(r)); }
        }
    HPCLA_CHECK_RCCL(g_rccl.GroupEnd());
    return HPCLA_OK;
}

// ---- halo plan ----------------------------------------------------------------------------------
static void halo_free(hpcla_halo_plan *p)
{
    if (!p) return;
    if (p->send_idx) (void)hipFree(p->send_idx);
    if (p->send_buf) (void)hipFree(p->send_buf);
    push_free(p);                                  // peer mappings + the window (which holds the ghost)
    for (void *q : p->probe_scratch) (void)hipFree(q);
    p->probe_scratch.clear();
    if (p->ghost) (void)hipFree(p->ghost);
    if (p->ev_ready) (void)hipEventDestroy(p->ev_ready);
    if (p->ev_done) (void)hipEventDestroy(p->ev_done);
    delete p;                                      // the exchange stream goes with the last plan that shares it (side_owner)
}

template <typename I>
static void scan_contiguous(hpcla_halo_plan *p, const std::vector<I> &idx)
{
    p->need_pack = false;
    for (size_t i = 0; i < p->send_ranks.size(); ++i) {
        const int64_t off = p->send_off[i], cnt = p->send_counts[i];
        bool contig = true;
        for (int64_t k = 1; k < cnt; ++k)
            if ((int64_t)idx[off + k] != (int64_t)idx[off] + k) { contig = false; break; }
        p->send_contig[i] = contig ? 1 : 0;
        p->send_first[i] = cnt > 0 ? (int64_t)idx[off] : 0;
        if (!contig) p->need_pack = true;
    }
}

static int halo_plan_create_impl(hpcla_halo_plan_t **plan, hpcla_comm_t *comm, int n_send,
                                 const int32_t *send_ranks_host,
                                 const int64_t *send_counts_host, const void *send_idx,
                                 int idx_is_i64, int n_recv, const int32_t *recv_ranks_host,
                                 const int64_t *recv_counts_host, int width, int flags)
{
    if (!plan || !comm) return set_error(HPCLA_ERR_INVALID, "halo_plan_create: null plan/comm");
    if (n_send < 0 || n_recv < 0 || width < 1)
        return set_error(HPCLA_ERR_INVALID, "halo_plan_create: bad counts/width");
    if ((n_send > 0 && (!send_ranks_host || !send_counts_host)) ||
        (n_recv > 0 && (!recv_ranks_host || !recv_counts_host)))
        return set_error(HPCLA_ERR_INVALID, "halo_plan_create: null rank/count lists");
    if ((n_send > 0 || n_recv > 0) && !comm->nccl && !comm->win_attached)
        return set_error(HPCLA_ERR_INVALID,
                         "halo_plan_create: neighbours given but the communicator is serial");
    hpcla_halo_plan *p = new (std::nothrow) hpcla_halo_plan();
    if (!p) return set_error(HPCLA_ERR_ALLOC, "halo_plan_create: out of memory");
    p->comm = comm;
    p->width = width;
    p->single_buffer = (flags & HPCLA_HALO_SINGLE_BUFFER) != 0;
    p->idx_is_i64 = idx_is_i64 ? 1 : 0;
    for (int i = 0; i < n_send; ++i) {
        if (send_ranks_host[i] < 0 || send_ranks_host[i] >= comm->nranks || send_counts_host[i] < 0) {
            halo_free(p);
            return set_error(HPCLA_ERR_INVALID, "halo_plan_create: bad send entry %d", i);
        }
        p->send_ranks.push_back(send_ranks_host[i]);
        p->send_counts.push_back(send_counts_host[i]);
        p->send_off.push_back(p->n_send_total);
        p->n_send_total += send_counts_host[i];
    }
    for (int i = 0; i < n_recv; ++i) {
        if (recv_ranks_host[i] < 0 || recv_ranks_host[i] >= comm->nranks || recv_counts_host[i] < 0) {
            halo_free(p);
            return set_error(HPCLA_ERR_INVALID, "halo_plan_create: bad recv entry %d", i);
        }
        p->recv_ranks.push_back(recv_ranks_host[i]);
        p->recv_counts.push_back(recv_counts_host[i]);
        p->recv_off.push_back(p->n_ghost);
        p->n_ghost += recv_counts_host[i];
    }
    p->send_contig.assign(n_send, 1);
    p->send_first.assign(n_send, 0);
    if (p->n_send_total > 0 && !send_idx) {
        halo_free(p);
        return set_error(HPCLA_ERR_INVALID, "halo_plan_create: null send_idx");
    }
#define HALO_HIP(expr)                                                                            \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            halo_free(p);                                                                         \
            return set_error(HPCLA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));       \
        }                                                                                         \
    } while (0)
    const size_t isz = idx_is_i64 ? 8 : 4;
    if (p->n_send_total > 0) {
        HALO_HIP(hipMalloc(&p->send_idx, p->n_send_total * isz));
        HALO_HIP(hipMemcpy(p->send_idx, send_idx, p->n_send_total * isz, hipMemcpyDeviceToDevice));
        // host scan: a neighbour whose indices are one ascending run is sent straight from x
        if (idx_is_i64) {
            std::vector<int64_t> h(p->n_send_total);
            HALO_HIP(hipMemcpy(h.data(), p->send_idx, p->n_send_total * isz, hipMemcpyDeviceToHost));
            scan_contiguous(p, h);
        } else {
            std::vector<int32_t> h(p->n_send_total);
            HALO_HIP(hipMemcpy(h.data(), p->send_idx, p->n_send_total * isz, hipMemcpyDeviceToHost));
            scan_contiguous(p, h);
        }
        if (p->need_pack && comm->nccl)
            HALO_HIP(hipMalloc((void **)&p->send_buf, p->n_send_total * width * sizeof(double)));
    }
    // ghost segment: inside a peer-mappable window when the push transport may serve this plan (the
    // communicator's own window is attached, i.e. all ranks share a node, and HPCLA_HALO_MODE does not
    // name an RCCL mode), else a plain allocation that RCCL receives into
    if (comm->win_attached && (n_send > 0 || n_recv > 0) && (halo_want_window() || !comm->nccl)) {
        int rcw = push_plan_alloc(p);
        if (rcw && comm->nccl) {
            // no peer-mappable window to be had: this plan stays an RCCL plan (it exports an empty descriptor, so
            // the collective attach step keeps every rank's copy of it on RCCL)
            push_free(p);
            rcw = HPCLA_OK;
            if (p->n_ghost > 0) {
                HALO_HIP(hipMalloc((void **)&p->ghost, p->n_ghost * width * sizeof(double)));
                HALO_HIP(hipMemset(p->ghost, 0, p->n_ghost * width * sizeof(double)));
            }
        }
        if (rcw) { halo_free(p); return rcw; }
    } else if (p->n_ghost > 0) {
        HALO_HIP(hipMalloc((void **)&p->ghost, p->n_ghost * width * sizeof(double)));
        HALO_HIP(hipMemset(p->ghost, 0, p->n_ghost * width * sizeof(double)));
    }
    HALO_HIP(hipStreamSynchronize(nullptr));     // plan-time fills are on the null stream; the caller's stream may be non-blocking
    {
        // highest stream priority, so that side-stream work is dispatched ahead of the caller's queued
        // workgroups wherever a CU has room for it (next to the SpMV kernel none has: spmv_dist_impl)
        int prio_least = 0, prio_greatest = 0;
        HALO_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        p->side_owner = std::make_shared<hpcla::SideStream>();
        HALO_HIP(hipStreamCreateWithPriority(&p->side_owner->s, hipStreamNonBlocking, prio_greatest));
        p->side = p->side_owner->s;
    }
    HALO_HIP(hipEventCreateWithFlags(&p->ev_ready, hipEventDisableTiming));
    HALO_HIP(hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming));
#undef HALO_HIP
    *plan = p;
    return HPCLA_OK;
}

HPCLA_API int hpcla_halo_plan_create(hpcla_halo_plan_t **plan, hpcla_comm_t *comm, int n_send,
                                     const int32_t *send_ranks_host,
                                     const int64_t *send_counts_host, const void *send_idx,
                                     int idx_is_i64, int n_recv, const int32_t *recv_ranks_host,
                                     const int64_t *recv_counts_host, int width)
{
    return halo_plan_create_impl(plan, comm, n_send, send_ranks_host, send_counts_host, send_idx, idx_is_i64, n_recv,
                                 recv_ranks_host, recv_counts_host, width, 0);
}

HPCLA_API int hpcla_halo_plan_create_ex(hpcla_halo_plan_t **plan, hpcla_comm_t *comm, int n_send,
                                        const int32_t *send_ranks_host,
                                        const int64_t *send_counts_host, const void *send_idx,
                                        int idx_is_i64, int n_recv, const int32_t *recv_ranks_host,
                                        const int64_t *recv_counts_host, int width, int flags)
{
    if (flags & ~HPCLA_HALO_SINGLE_BUFFER) return set_error(HPCLA_ERR_INVALID, "halo_plan_create_ex: unknown flags");
    return halo_plan_create_impl(plan, comm, n_send, send_ranks_host, send_counts_host, send_idx, idx_is_i64, n_recv,
                                 recv_ranks_host, recv_counts_host, width, flags);
}

HPCLA_API int hpcla_halo_plan_chain(hpcla_halo_plan_t *plan, hpcla_halo_plan_t *leader)
{
    if (!plan || !leader || plan == leader) return set_error(HPCLA_ERR_INVALID, "halo_plan_chain: bad plans");
    if (!leader->side) return set_error(HPCLA_ERR_INVALID, "halo_plan_chain: the leader has no exchange stream");
    if (plan->side == leader->side) return HPCLA_OK;
    if (plan->side) HPCLA_CHECK_HIP(hipStreamSynchronize(plan->side));
    plan->side_owner = leader->side_owner;         // drops this plan's own stream (destroyed if nobody else shares it)
    plan->side = leader->side;
    return HPCLA_OK;
}

HPCLA_API int hpcla_halo_plan_destroy(hpcla_halo_plan_t *plan)
{
    if (!plan) return HPCLA_OK;
    if (plan->side) (void)hipStreamSynchronize(plan->side);
    halo_free(plan);
    return HPCLA_OK;
}

HPCLA_API int hpcla_halo_ghost_ptr(hpcla_halo_plan_t *plan, double **ghost, int64_t *n_ghost)
{
    if (!plan) return set_error(HPCLA_ERR_INVALID, "halo_ghost_ptr: null plan");
    // push transport: the buffer of the exchange posted last (double-buffered window)
    if (ghost) *ghost = plan->attached ? push_ghost_ptr(plan) : plan->ghost;
    if (n_ghost) *n_ghost = plan->n_ghost;
    return HPCLA_OK;
}

static int halo_post(hpcla_halo_plan_t *plan, const double *x, void *stream, bool record_done,
                     bool inline_on_main = false);

HPCLA_API int hpcla_halo_begin(hpcla_halo_plan_t *plan, const double *x, void *stream)
{
    if (plan && halo_active(plan) && halo_mode_of(plan) == HALO_PUSH) {
        // the push kernel IS the copy engine: run it on the side stream (after everything enqueued on the
        // caller's stream) so that a large payload (SpMM ghost rows) overlaps the caller's interior work
        if (plan->n_send_total > 0 && !x) return set_error(HPCLA_ERR_INVALID, "halo_begin: null x");
        HPCLA_CHECK_HIP(hipEventRecord(plan->ev_ready, as_stream(stream)));
        HPCLA_CHECK_HIP(hipStreamWaitEvent(plan->side, plan->ev_ready, 0));
        int rc = push_post(plan, x, 1, plan->side);          // readers: the push workgroups + halo_end's wait kernel
        if (rc) return rc;
        HPCLA_CHECK_HIP(hipEventRecord(plan->ev_done, plan->side));
        return HPCLA_OK;
    }
    return halo_post(plan, x, stream, true);
}

// posts the exchange on the plan's side stream; with record_done == false the caller enqueues more
// work on the side stream (the boundary row blocks) and records ev_done itself
static int halo_post(hpcla_halo_plan_t *plan, const double *x, void *stream, bool record_done,
                     bool inline_on_main)
{
    if (!plan) return set_error(HPCLA_ERR_INVALID, "halo_begin: null plan");
    if (plan->send_ranks.empty() && plan->recv_ranks.empty()) return HPCLA_OK;
    if (plan->n_send_total > 0 && !x) return set_error(HPCLA_ERR_INVALID, "halo_begin: null x");
    hipStream_t main = as_stream(stream);
    const int w = plan->width;
    // the exchange may start once everything enqueued on the caller's stream (the producer of x, and
    // the previous consumer of the ghost segment) has finished
    hipStream_t cs = inline_on_main ? main : plan->side;      // stream that carries the exchange
    if (!inline_on_main) {
        HPCLA_CHECK_HIP(hipEventRecord(plan->ev_ready, main));
        HPCLA_CHECK_HIP(hipStreamWaitEvent(plan->side, plan->ev_ready, 0));
    }
    if (plan->need_pack) {
        const int64_t total = plan->n_send_total * w;
        int64_t g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        if (plan->idx_is_i64)
            pack_kernel<int64_t><<<(uint32_t)g, 256, 0, cs>>>(
                x, (const int64_t *)plan->send_idx, plan->send_buf, plan->n_send_total, w);
        else
            pack_kernel<int32_t><<<(uint32_t)g, 256, 0, cs>>>(
                x, (const int32_t *)plan->send_idx, plan->send_buf, plan->n_send_total, w);
        HPCLA_CHECK_LAUNCH();
    }
    ncclComm_t nc = plan->comm->nccl;
    if (!nc) return set_error(HPCLA_ERR_INVALID, "halo exchange: this communicator has no RCCL transport");
    // a failed Send/Recv must not leave the thread inside an open group (later RCCL calls would be
    // queued into it), and once the side stream carries work the caller's stream must still be joined
    // to it: close the group, record ev_done, make the caller wait, then report
    ncclResult_t bad = ncclSuccess;
    const char *what = "";
    HPCLA_CHECK_RCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < plan->recv_ranks.size() && bad == ncclSuccess; ++i) {
        if (plan->recv_counts[i] == 0) continue;
        bad = g_rccl.Recv(plan->ghost + plan->recv_off[i] * w, (size_t)(plan->recv_counts[i] * w), ncclDouble,
                          plan->recv_ranks[i], nc, cs);
        what = "ncclRecv";
    }
    for (size_t i = 0; i < plan->send_ranks.size() && bad == ncclSuccess; ++i) {
        if (plan->send_counts[i] == 0) continue;
        const double *src = plan->send_contig[i] ? x + plan->send_first[i] * w
                                                 : plan->send_buf + plan->send_off[i] * w;
        bad = g_rccl.Send(src, (size_t)(plan->send_counts[i] * w), ncclDouble, plan->send_ranks[i], nc, cs);
        what = "ncclSend";
    }
    const ncclResult_t endr = g_rccl.GroupEnd();
    if (bad != ncclSuccess || endr != ncclSuccess) {
        if (!inline_on_main) {
            (void)hipEventRecord(plan->ev_done, plan->side);
            (void)hipStreamWaitEvent(main, plan->ev_done, 0);
        }
        if (bad != ncclSuccess)
            return set_error(HPCLA_ERR_RCCL, "%s failed: %s", what, g_rccl.GetErrorString(bad));
        return set_error(HPCLA_ERR_RCCL, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(endr));
    }
    if (record_done && !inline_on_main) HPCLA_CHECK_HIP(hipEventRecord(plan->ev_done, plan->side));
    return HPCLA_OK;
}

namespace hpcla {
// the exchange on the caller's stream itself (serial mode), for packed.hip
int halo_exchange_inline(hpcla_halo_plan_t *plan, const double *x, void *stream)
{
    return halo_post(plan, x, stream, false, true);
}

bool halo_serial_mode(const hpcla_halo_plan_t *plan) { return halo_mode_of(plan) != HALO_OVERLAP; }
}  // namespace hpcla

HPCLA_API int hpcla_halo_end(hpcla_halo_plan_t *plan, void *stream)
{
    if (!plan) return set_error(HPCLA_ERR_INVALID, "halo_end: null plan");
    if (plan->send_ranks.empty() && plan->recv_ranks.empty()) return HPCLA_OK;
    HPCLA_CHECK_HIP(hipStreamWaitEvent(as_stream(stream), plan->ev_done, 0));   // my own exchange work is done
    // push transport: additionally wait (on the device) until every neighbour has published this epoch
    if (halo_mode_of(plan) == HALO_PUSH) return push_wait_kernel_launch(plan, stream);
    return HPCLA_OK;
}

// ---- connection test of ONE plan on the real topology ---------------------------------------------------
// x[e] = rank * 2^40 + e (+ shift): a ghost slot filled by rank r from its row i holds r * 2^40 + i*w + j.
__global__ void probe_reset_kernel(unsigned long long *bad)
{
    bad[0] = 0ULL;           // mismatches seen
    bad[1] = ~0ULL;          // first mismatching ghost element (minimum)
}

__global__ void probe_fill_kernel(double *x, int64_t n, double base, double shift)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = base + (double)i + shift;
}

// EVERY workgroup checks EVERY sampled element (the grid is a few workgroups per XCD): after round one each XCD's L2
// (and the L1 of every CU that ran a workgroup) holds the sampled lines of the ghost buffer, so round two -- which
// rewrites a single-buffered (cacheable, fine-grained) dense ghost window with DIFFERENT values -- reads them WARM on
// every XCD.  A line a peer's xGMI store failed to invalidate here shows up as a wrong value, the plan fails its
// connection test on the real topology and stays on RCCL.  (Vector ghost windows are uncached and double-buffered.)
// The ghost buffer of the exchange completed last is found HERE, from the plan's device-resident step counter (`done`, written
// by the wait kernel that precedes this launch on the stream): buffer done % nbuf.  The host's way to the same pointer --
// hpcla_halo_ghost_ptr on a double-buffered plan -- synchronises the whole DEVICE, which a collective must not do (below).
__global__ void probe_check_kernel(const double *ghost0, int64_t buf_stride, int nbuf, const uint64_t *done,
                                   const int64_t *slots, const int64_t *rows,
                                   const int32_t *owners, int64_t n_check, int w, double shift,
                                   unsigned long long *bad)      // bad[0] = count, bad[1] = first wrong element
{
    const uint64_t completed = (done && nbuf > 1) ? __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const double *ghost = ghost0 + (int64_t)(completed % (uint64_t)(nbuf > 1 ? nbuf : 1)) * buf_stride;
    const int64_t total = n_check * w;
    for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
        const int64_t t = e / w;
        const int j = (int)(e % w);
        const double want = (double)owners[t] * 1099511627776.0 + (double)(rows[t] * w + j) + shift;
        const int64_t at = slots[t] * w + j;
        if (ghost[at] != want) {
            atomicAdd(&bad[0], 1ULL);
            atomicMin(&bad[1], (unsigned long long)at);
        }
    }
}

constexpr int PROBE_CHECK_BLOCKS = 64;          // 8 per XCD (workgroups are dealt round-robin over the XCDs)

HPCLA_API int hpcla_halo_plan_probe(hpcla_halo_plan_t *plan, int64_t n_local_rows, const int64_t *check_slots_host,
                                    const int64_t *check_rows_host, int64_t n_check, void *stream, int *ok)
{
    if (!plan || !ok) return set_error(HPCLA_ERR_INVALID, "halo_plan_probe: null pointer");
    *ok = 0;
    if (n_local_rows < 0 || n_check < 0 || (n_check > 0 && (!check_slots_host || !check_rows_host)))
        return set_error(HPCLA_ERR_INVALID, "halo_plan_probe: bad arguments");
    if (plan->send_ranks.empty() && plan->recv_ranks.empty()) { *ok = 1; return HPCLA_OK; }
    const int w = plan->width;
    // owner of every checked slot, from the plan's own segment table
    std::vector<int32_t> owners((size_t)n_check);
    for (int64_t t = 0; t < n_check; ++t) {
        const int64_t sl = check_slots_host[t];
        if (sl < 0 || sl >= plan->n_ghost || check_rows_host[t] < 0)
            return set_error(HPCLA_ERR_INVALID, "halo_plan_probe: slot %lld outside the ghost (%lld indices)",
                             (long long)sl, (long long)plan->n_ghost);
        int32_t who = -1;
        for (size_t j = 0; j < plan->recv_ranks.size(); ++j)
            if (sl >= plan->recv_off[j] && sl < plan->recv_off[j] + plan->recv_counts[j]) { who = plan->recv_ranks[j]; break; }
        if (who < 0) return set_error(HPCLA_ERR_INVALID, "halo_plan_probe: slot %lld lies in no segment", (long long)sl);
        owners[(size_t)t] = who;
    }
    hipStream_t s = as_stream(stream);
    const int64_t nx = n_local_rows * w > 0 ? n_local_rows * w : 1;
    double *x = nullptr;
    int64_t *d_slots = nullptr, *d_rows = nullptr;
    int32_t *d_own = nullptr;
    unsigned long long *d_bad = nullptr;
    // COLLECTIVE HYGIENE (round 6).  Ranks may be host threads of ONE process sharing the device (tests/cabi/
    // cabi_ranks_threads.c: how 8 ranks are rehearsed on a box that allows 6 GPU processes).  Between the first launch of
    // this function and its last read-back some rank's kernel may be spinning for a launch another rank has not made yet,
    // so nothing in here may wait for the DEVICE: a device-wide wait on a slow rank includes a fast rank's wait kernel,
    // which spins for the slow rank's NEXT push -- a cycle only the spin bound breaks.  That was the stall of round 6:
    // hpcla_halo_ghost_ptr on a double-buffered plan is hipDeviceSynchronize + a read of the step counter, and the probe
    // called it once per round (8 of 24 probes timed out in round 1 at 8 thread-ranks; 0 of 24 since the check kernel finds
    // the buffer itself: profiles/MEASUREMENTS_r06.md section E).  For the same reason: no hipFree in here (with in-process
    // peers the scratch goes at the plan's destroy), no blocking copy on the process-wide null stream, no async copy to or
    // from pageable host memory behind a waiting kernel.  Copies run on the CALLER's stream while it is idle.
    std::vector<void *> mine;
    auto scratch = [&](void **q, size_t bytes) {
        hipError_t a = hipMalloc(q, bytes);
        if (a == hipSuccess) mine.push_back(*q);
        return a;
    };
    hipError_t e = scratch((void **)&x, (size_t)nx * sizeof(double));
    if (e == hipSuccess) e = scratch((void **)&d_bad, 2 * sizeof(unsigned long long));
    if (e == hipSuccess && n_check) {
        e = scratch((void **)&d_slots, (size_t)n_check * sizeof(int64_t));
        if (e == hipSuccess) e = scratch((void **)&d_rows, (size_t)n_check * sizeof(int64_t));
        if (e == hipSuccess) e = scratch((void **)&d_own, (size_t)n_check * sizeof(int32_t));
        // (the caller's stream is idle here -- nothing of this plan has been launched yet -- and a copy from pageable memory
        //  returns once the host buffer has been read: `owners` may go out of scope afterwards)
        if (e == hipSuccess) e = hipMemcpyAsync(d_slots, check_slots_host, (size_t)n_check * sizeof(int64_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_rows, check_rows_host, (size_t)n_check * sizeof(int64_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_own, owners.data(), (size_t)n_check * sizeof(int32_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    int rc = HPCLA_OK;
    bool good = (e == hipSuccess);
    char why[256] = "";
    // two exchanges: one per ghost buffer of a double-buffered plan, and round two's values differ from round
    // one's, so a stale buffer cannot pass.  Every rank with a plan makes the same two exchanges.
    for (int round = 0; round < 2 && e == hipSuccess && rc == HPCLA_OK; ++round) {
        const double shift = round ? 0.25 : 0.0;
        // (a kernel, not a copy from pageable host memory: see the read-back below)
        probe_reset_kernel<<<1, 1, 0, s>>>(d_bad);
        int64_t g = (nx + 255) / 256;
        probe_fill_kernel<<<(uint32_t)(g > 4096 ? 4096 : g), 256, 0, s>>>(x, nx, (double)plan->comm->rank * 1099511627776.0, shift);
        rc = hpcla_halo_begin(plan, x, stream);
        if (rc == HPCLA_OK) rc = hpcla_halo_end(plan, stream);
        if (rc != HPCLA_OK) break;
        // (NOT hpcla_halo_ghost_ptr: COLLECTIVE HYGIENE above; the check kernel derives the buffer from the step counter)
        const bool attached = plan->attached && plan->win;
        const int nbuf = attached ? plan->nbuf : 1;
        const int64_t stride = nbuf > 1 ? (int64_t)(hpcla::halo_window_buf_bytes((uint64_t)plan->n_ghost, (uint32_t)plan->width) / sizeof(double)) : 0;
        if (n_check)
            probe_check_kernel<<<PROBE_CHECK_BLOCKS, 256, 0, s>>>(plan->ghost, stride, nbuf, attached ? plan->epoch_dev : nullptr,
                                                                  d_slots, d_rows, d_own, n_check, w, shift, d_bad);
        // read-backs on the caller's stream AFTER it has drained (an idle stream: the copies start at once)
        unsigned long long bad[2] = {0, 0};
        uint32_t timed_out = 0;
        e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipMemcpyAsync(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && plan->status) e = hipMemcpyAsync(&timed_out, plan->status, sizeof(timed_out), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) break;
        if (timed_out) {
            good = false;
            snprintf(why, sizeof(why), "halo_plan_probe: a push or a wait timed out in round %d", round);
            break;
        }
        if (bad[0]) {
            good = false;
            snprintf(why, sizeof(why), "halo_plan_probe: round %d: %llu of %lld checked ghost values wrong, first at ghost element %llu",
                     round, (bad[0] + PROBE_CHECK_BLOCKS - 1) / PROBE_CHECK_BLOCKS, (long long)(n_check * w), bad[1]);
            break;
        }
    }
    // the scratch (n_local_rows * width doubles: 2 GB for config 5's width-16 plan): freed here when every peer is another
    // PROCESS -- its kernels are not this process's to wait for --, kept until the plan's destroy when a peer is a thread of
    // this process (COLLECTIVE HYGIENE above: hipFree waits for the whole device)
    bool in_process_peer = false;
    for (const auto &m : plan->peer_maps) in_process_peer = in_process_peer || m.local_ref;
    if (in_process_peer) plan->probe_scratch.insert(plan->probe_scratch.end(), mine.begin(), mine.end());
    else for (void *q : mine) (void)hipFree(q);
    if (e != hipSuccess) return set_error(HPCLA_ERR_HIP, "halo_plan_probe: %s", hipGetErrorString(e));
    if (rc != HPCLA_OK) return rc;
    if (!good) (void)set_error(HPCLA_ERR_INVALID, "%s", why);        // the reason, for hpcla_last_error()
    *ok = good ? 1 : 0;
    return HPCLA_OK;
}

// contiguity of the (ascending) interior block list, probed once per list: a contiguous run is addressed
// by its base, because a per-workgroup list load would sit on the critical path of every workgroup
// (measured: +30 us per 4096^2 SpMV)
static int probe_interior(hpcla_halo_plan_t *plan, const int32_t *interior, int64_t n_interior)
{
    if (plan->probed_list == interior && plan->probed_n == n_interior) return HPCLA_OK;
    int32_t ends[2] = {0, 0};
    HPCLA_CHECK_HIP(hipMemcpy(&ends[0], interior, sizeof(int32_t), hipMemcpyDeviceToHost));
    HPCLA_CHECK_HIP(hipMemcpy(&ends[1], interior + (n_interior - 1), sizeof(int32_t), hipMemcpyDeviceToHost));
    plan->probed_list = interior;
    plan->probed_n = n_interior;
    plan->probed_first = ends[0];
    plan->probed_contig = ((int64_t)ends[1] - (int64_t)ends[0] + 1 == n_interior);
    return HPCLA_OK;
}

template <typename I, typename F, typename G>
static int spmv_dist_impl(F split_fn, G fused_fn, hpcla_halo_plan_t *plan, const I *rowptr, const I *colval,
                          const double *nzval, const double *x, int64_t n_own, double *y,
                          int64_t nrows, int64_t nnz, int index_base, const int32_t *interior,
                          int64_t n_interior, const int32_t *boundary, int64_t n_boundary,
                          void *stream, double *dot_partial = nullptr)
{
    const bool has_halo = plan && !(plan->send_ranks.empty() && plan->recv_ranks.empty());
    if (!has_halo) {
        // no neighbours: every column is owned; one launch over all row blocks
        return split_fn(rowptr, colval, nzval, x, plan ? plan->ghost : nullptr, n_own, y, nrows,
                        nnz, index_base, nullptr, 0, stream, dot_partial, -1);
    }
    if ((n_interior > 0 && !interior) || (n_boundary > 0 && !boundary))
        return set_error(HPCLA_ERR_INVALID, "spmv_dist: null block list");
    // side stream: exchange, then the boundary row blocks (they need the ghosts); caller's stream:
    // the interior row blocks, concurrently.  Both write disjoint rows of y; the caller's stream
    // joins the side stream at the end.
    // Two ways to order the step (HPCLA_HALO_MODE, default "serial"):
    //  serial   exchange on the CALLER's stream, then one launch over all row blocks.  No side stream, no
    //           events, no interior/boundary split.
    //  overlap  exchange + boundary blocks on the plan's side stream, interior blocks on the caller's.
    // Measured on MI355X (profiles/r01_halo_mode_experiments.log): the SpMV kernel keeps 8 workgroups =
    // all 32 wave slots of every CU busy, so a side-stream RCCL kernel only becomes resident when the
    // interior grid drains -- "overlap" then costs interior + exchange + boundary + two cross-stream hops
    // (+32 us per 4096^2 step) against exchange + kernel (+13 us) for "serial".  Overlap pays only next
    // to kernels that leave CU room (the SpMM path, 6 workgroups per CU, keeps it).
    const HaloMode mode = halo_mode_of(plan);
    if (mode == HALO_PUSH) {
        //  push     (default on one node) ONE launch: its leading workgroups store this rank's boundary values
        //           straight into the neighbours' ghost windows and publish the step's epoch; the interior blocks
        //           follow; the boundary blocks come last and poll the local flag lines before their first ghost
        //           gather (window.hip, halo_wait.h, spmv.hip).  No RCCL kernel, no side stream, no events, no
        //           extra launch; the exchange overlaps all interior blocks.
        if (n_interior > 0) {
            int rcq = probe_interior(plan, interior, n_interior);
            if (rcq) return rcq;
        }
        PushArgs pa;
        int rcp;
        const bool own_push_kernel = (plan->idx_is_i64 != 0) != (sizeof(I) == 8);
        if (!own_push_kernel) {
            rcp = push_begin(plan, x, n_boundary, &pa);    // the push rides in the leading workgroups of the launch
        } else {                                           // send lists typed unlike the matrix: push kernel of its own
            rcp = push_post(plan, x, n_boundary, stream);  // (commits the step's epoch readers: see below)
            memset(&pa, 0, sizeof(pa));
        }
        if (rcp) return rcp;
        const bool contig = n_interior > 0 && plan->probed_contig;
        // (the ghost argument is buffer 0; a boundary workgroup computes the buffer of its epoch after its wait)
        const int rcf = fused_fn(rowptr, colval, nzval, x, plan->ghost, n_own, y, nrows, nnz, index_base,
                                 contig ? nullptr : interior, contig ? plan->probed_first : 0, n_interior, boundary,
                                 n_boundary, push_wait_args(plan, n_boundary), pa, stream, dot_partial);
        // a refused launch behind an ALREADY POSTED push would strand the exchange's waiting readers (the step counter
        // never advances, every later exchange of the plan reuses the epoch): release them without computing
        if (rcf && own_push_kernel) (void)push_abandon_waiters(plan, n_boundary, stream);
        return rcf;
    }
    if (mode == HALO_SERIAL) {
        int rc0 = halo_post(plan, x, stream, false, true);
        if (rc0) return rc0;
        return split_fn(rowptr, colval, nzval, x, plan->ghost, n_own, y, nrows, nnz, index_base, nullptr, 0,
                        stream, dot_partial, -1);
    }
    int rc = halo_post(plan, x, stream, false);
    if (rc) return rc;
    // from here on the side stream carries work that writes the ghost segment / y: whatever fails below, the
    // caller's stream is joined to it before the error is reported
    auto join_and_fail = [&](int code) {
        (void)hipEventRecord(plan->ev_done, plan->side);
        (void)hipStreamWaitEvent(as_stream(stream), plan->ev_done, 0);
        return code;
    };
    if (n_boundary > 0) {
        rc = split_fn(rowptr, colval, nzval, x, plan->ghost, n_own, y, nrows, nnz, index_base,
                      boundary, n_boundary, plan->side, dot_partial, -1);
        if (rc) return join_and_fail(rc);
    }
    HPCLA_CHECK_HIP(hipEventRecord(plan->ev_done, plan->side));
    if (n_interior > 0) {
        rc = probe_interior(plan, interior, n_interior);
        if (rc) return join_and_fail(rc);
        if (plan->probed_contig)
            rc = split_fn(rowptr, colval, nzval, x, plan->ghost, n_own, y, nrows, nnz, index_base,
                          nullptr, n_interior, stream, dot_partial, plan->probed_first);
        else
            rc = split_fn(rowptr, colval, nzval, x, plan->ghost, n_own, y, nrows, nnz, index_base,
                          interior, n_interior, stream, dot_partial, -1);
        if (rc) return join_and_fail(rc);
    }
    return hpcla_halo_end(plan, stream);
}

// y = A*x and out = x.y in one pass over A (CG's p.Ap): the SpMV workgroups leave per-row-block
// partials in `work`, summed in index order afterwards, then all-reduced.  Needs x partitioned like
// A's rows (n_own == nrows) and interior+boundary lists that cover every row block exactly once.
template <typename I, typename F, typename G>
static int spmv_dist_dot_impl(F split_fn, G fused_fn, hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const I *rowptr,
                              const I *colval, const double *nzval, const double *x, int64_t n_own,
                              double *y, int64_t nrows, int64_t nnz, int index_base,
                              const int32_t *interior, int64_t n_interior, const int32_t *boundary,
                              int64_t n_boundary, double *dot_out_dev, void *work, void *stream)
{
    if (!dot_out_dev || !work) return set_error(HPCLA_ERR_INVALID, "spmv_dist_dot: null out/work");
    if (n_own != nrows)
        return set_error(HPCLA_ERR_INVALID, "spmv_dist_dot: x must be partitioned like the rows of A");
    const int rpb = hpcla_spmv_rows_per_block();
    const int64_t all_blocks = (nrows + rpb - 1) / rpb;
    const bool has_halo = plan && !(plan->send_ranks.empty() && plan->recv_ranks.empty());
    if (has_halo && n_interior + n_boundary != all_blocks)
        return set_error(HPCLA_ERR_INVALID, "spmv_dist_dot: block lists must cover every row block");
    double *scratch = reinterpret_cast<double *>(work);          // 2048 doubles of stage-1 scratch
    double *partial = scratch + 2048;                            // then one double per row block
    int rc = spmv_dist_impl<I>(split_fn, fused_fn, plan, rowptr, colval, nzval, x, n_own, y, nrows, nnz, index_base,
                               interior, n_interior, boundary, n_boundary, stream, partial);
    if (rc) return rc;
    rc = reduce_partials_sum(partial, all_blocks, scratch, dot_out_dev, stream);
    if (rc) return rc;
    if (comm) return allreduce_on(comm, dot_out_dev, 1, 0, stream);
    return HPCLA_OK;
}

HPCLA_API int64_t hpcla_spmv_dot_work_bytes(int64_t nrows)
{
    const int rpb = hpcla_spmv_rows_per_block();
    return (int64_t)sizeof(double) * (2048 + (nrows + rpb - 1) / rpb + 1);
}

HPCLA_API int hpcla_spmv_dist_dot_f64_i32(hpcla_halo_plan_t *plan, hpcla_comm_t *comm,
                                          const int32_t *rowptr, const int32_t *colval_split,
                                          const double *nzval, const double *x, int64_t n_own,
                                          double *y, int64_t nrows, int64_t nnz, int index_base,
                                          const int32_t *interior_blocks, int64_t n_interior,
                                          const int32_t *boundary_blocks, int64_t n_boundary,
                                          double *dot_out_dev, void *work, void *stream)
{
    return spmv_dist_dot_impl<int32_t>(spmv_split_i32, spmv_fused_i32, plan, comm, rowptr, colval_split, nzval, x, n_own,
                                       y, nrows, nnz, index_base, interior_blocks, n_interior,
                                       boundary_blocks, n_boundary, dot_out_dev, work, stream);
}

HPCLA_API int hpcla_spmv_dist_dot_f64_i64(hpcla_halo_plan_t *plan, hpcla_comm_t *comm,
                                          const int64_t *rowptr, const int64_t *colval_split,
                                          const double *nzval, const double *x, int64_t n_own,
                                          double *y, int64_t nrows, int64_t nnz, int index_base,
                                          const int32_t *interior_blocks, int64_t n_interior,
                                          const int32_t *boundary_blocks, int64_t n_boundary,
                                          double *dot_out_dev, void *work, void *stream)
{
    return spmv_dist_dot_impl<int64_t>(spmv_split_i64, spmv_fused_i64, plan, comm, rowptr, colval_split, nzval, x, n_own,
                                       y, nrows, nnz, index_base, interior_blocks, n_interior,
                                       boundary_blocks, n_boundary, dot_out_dev, work, stream);
}

HPCLA_API int hpcla_spmv_dist_f64_i32(hpcla_halo_plan_t *plan, const int32_t *rowptr,
                                      const int32_t *colval_split, const double *nzval,
                                      const double *x, int64_t n_own, double *y, int64_t nrows,
                                      int64_t nnz, int index_base, const int32_t *interior_blocks,
                                      int64_t n_interior, const int32_t *boundary_blocks,
                                      int64_t n_boundary, void *stream)
{
    return spmv_dist_impl<int32_t>(spmv_split_i32, spmv_fused_i32, plan, rowptr, colval_split, nzval, x, n_own, y,
                                   nrows, nnz, index_base, interior_blocks, n_interior,
                                   boundary_blocks, n_boundary, stream);
}

HPCLA_API int hpcla_spmv_dist_f64_i64(hpcla_halo_plan_t *plan, const int64_t *rowptr,
                                      const int64_t *colval_split, const double *nzval,
                                      const double *x, int64_t n_own, double *y, int64_t nrows,
                                      int64_t nnz, int index_base, const int32_t *interior_blocks,
                                      int64_t n_interior, const int32_t *boundary_blocks,
                                      int64_t n_boundary, void *stream)
{
    return spmv_dist_impl<int64_t>(spmv_split_i64, spmv_fused_i64, plan, rowptr, colval_split, nzval, x, n_own, y,
                                   nrows, nnz, index_base, interior_blocks, n_interior,
                                   boundary_blocks, n_boundary, stream);
}

// ---- k fused CG iterations in ONE host call ----------------------------------------------------------
// The iteration a caller composes from A*p, dot, the broadcasts and norm (the reference has no Krylov
// solver: src/sparse.jl:2096-2128, src/vectors.jl:798-812, 1203-1226, 758-765), enqueued `iters` times on
// `stream` without returning to the host language in between: per iteration the launches of
// hpcla_spmv_dist_dot_* (Ap = A p, pAp), hpcla_cg_residual_f64 (r -= a Ap, rr') and hpcla_cg_direction_f64
// (x += a p, p = r + (rr'/rr) p) -- the same kernels with the same arguments as the three separate calls,
// hence the same bits.  rr_hist_dev[j] holds sum r_j^2: [0] on entry, [1..iters] written here.
template <typename I, typename F, typename G>
static int cg_iterations_impl(F split_fn, G fused_fn, hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const I *rowptr,
                              const I *colval, const double *nzval, int64_t nrows, int64_t nnz, int index_base,
                              const int32_t *interior, int64_t n_interior, const int32_t *boundary,
                              int64_t n_boundary, double *x, double *r, double *p, double *Ap, double *rr_hist_dev,
                              double *pAp_dev, void *dot_work, void *reduce_work, int iters, void *stream)
{
    if (iters < 0) return set_error(HPCLA_ERR_INVALID, "cg_iterations: negative iteration count");
    if (!rr_hist_dev || !pAp_dev || !dot_work || !reduce_work)
        return set_error(HPCLA_ERR_INVALID, "cg_iterations: null scalar / work buffer");
    if (nrows > 0 && (!x || !r || !p || !Ap)) return set_error(HPCLA_ERR_INVALID, "cg_iterations: null vector");
    for (int j = 0; j < iters; ++j) {
        double *rr = rr_hist_dev + j;
        int rc = spmv_dist_dot_impl<I>(split_fn, fused_fn, plan, comm, rowptr, colval, nzval, p, nrows, Ap, nrows, nnz,
                                       index_base, interior, n_interior, boundary, n_boundary, pAp_dev, dot_work,
                                       stream);
        if (rc) return rc;
        rc = hpcla_cg_residual_f64(comm, 1.0, rr, pAp_dev, Ap, r, nrows, rr + 1, reduce_work, stream);
        if (rc) return rc;
        rc = hpcla_cg_direction_f64(1.0, rr, pAp_dev, 1.0, rr + 1, rr, r, x, p, nrows, stream);
        if (rc) return rc;
    }
    return HPCLA_OK;
}

HPCLA_API int hpcla_cg_iterations_f64_i32(hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const int32_t *rowptr,
                                          const int32_t *colval_split, const double *nzval, int64_t nrows,
                                          int64_t nnz, int index_base, const int32_t *interior_blocks,
                                          int64_t n_interior, const int32_t *boundary_blocks, int64_t n_boundary,
                                          double *x, double *r, double *p, double *Ap, double *rr_hist_dev,
                                          double *pAp_dev, void *dot_work, void *reduce_work, int iters,
                                          void *stream)
{
    return cg_iterations_impl<int32_t>(spmv_split_i32, spmv_fused_i32, plan, comm, rowptr, colval_split, nzval, nrows,
                                       nnz, index_base, interior_blocks, n_interior, boundary_blocks, n_boundary, x, r,
                                       p, Ap, rr_hist_dev, pAp_dev, dot_work, reduce_work, iters, stream);
}

HPCLA_API int hpcla_cg_iterations_f64_i64(hpcla_halo_plan_t *plan, hpcla_comm_t *comm, const int64_t *rowptr,
                                          const int64_t *colval_split, const double *nzval, int64_t nrows,
                                          int64_t nnz, int index_base, const int32_t *interior_blocks,
                                          int64_t n_interior, const int32_t *boundary_blocks, int64_t n_boundary,
                                          double *x, double *r, double *p, double *Ap, double *rr_hist_dev,
                                          double *pAp_dev, void *dot_work, void *reduce_work, int iters,
                                          void *stream)
{
    return cg_iterations_impl<int64_t>(spmv_split_i64, spmv_fused_i64, plan, comm, rowptr, colval_split, nzval, nrows,
                                       nnz, index_base, interior_blocks, n_interior, boundary_blocks, n_boundary, x, r,
                                       p, Ap, rr_hist_dev, pAp_dev, dot_work, reduce_work, iters, stream);
}
