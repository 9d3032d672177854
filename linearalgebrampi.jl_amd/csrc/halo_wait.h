// halo_wait.h -- consumer side of the peer-window push transport (window.hip): wait inside a kernel
// until every recv neighbour has published the step's epoch, then make its payload visible.
#pragma once
#include "common.h"

namespace hpcla {

constexpr int WIN_FLAG_STRIDE_U64 = 16;        // one polled word per 128-byte line

// The step counter of a plan lives in DEVICE memory: `done` = exchanges completed, `ticket` = workgroups of
// the exchange in flight that no longer need its number.  Every workgroup that takes part in an exchange
// (push workgroups, waiting workgroups: its `n_readers`, possibly spread over two kernels) reads
// e = done + 1 when it starts and releases it once; the LAST release stores done = e.  Nobody writes `done`
// while a reader can still read it, and no launch argument changes from step to step -- so a step can be
// captured into a HIP graph and replayed.
//
// The releases are counted in TWO levels (EPOCH_SHARDS counters on lines of their own, then one top counter):
// a 3-D slab has thousands of waiting workgroups, and read-modify-write atomics on ONE address serialise at
// ~40 ns each (2048 boundary blocks on a single ticket cost +84 us per step; sharded: back to +10).
constexpr uint32_t EPOCH_SHARDS = 64;
constexpr uint32_t EPOCH_SHARD_STRIDE = 16;          // u32 words between shard counters (64 bytes)
constexpr size_t EPOCH_BYTES = 64 + 64 + (size_t)EPOCH_SHARDS * EPOCH_SHARD_STRIDE * 4;   // done | top | shards

struct EpochRef {
    uint64_t *done;          // + 16 u32 words: the top counter; + 32 words: the shard counters
    uint32_t n_readers;
};

__device__ __forceinline__ uint64_t epoch_current(const EpochRef &r)
{
    return __hip_atomic_load(r.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
}

// ONE lane per reader workgroup; `reader` = this workgroup's index among the exchange's readers (0..n_readers-1)
__device__ __forceinline__ void epoch_release(const EpochRef &r, uint64_t e, uint32_t reader)
{
    uint32_t *top = reinterpret_cast<uint32_t *>(r.done) + 16;
    uint32_t *shards = reinterpret_cast<uint32_t *>(r.done) + 32;
    const uint32_t sh = reader % EPOCH_SHARDS;
    const uint32_t in_shard = (r.n_readers - sh + EPOCH_SHARDS - 1) / EPOCH_SHARDS;     // readers with this residue
    uint32_t *cnt = shards + sh * EPOCH_SHARD_STRIDE;
    if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != in_shard) return;
    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t active = r.n_readers < EPOCH_SHARDS ? r.n_readers : EPOCH_SHARDS;
    if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != active) return;
    __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r.done, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what a consuming launch needs: the plan's LOCAL flag lines, its step counter, and where the ghost buffer of
// a given epoch is
struct HaloWait {
    const uint64_t *flags;                     // n_flags lines, stride WIN_FLAG_STRIDE_U64
    int n_flags;
    EpochRef er;
    uint32_t *status;                          // set to 1 when a spin timed out (the grid still drains)
    int64_t timeout_ticks;                     // wall_clock64 ticks (100 MHz)
    const double *ghost0;                      // ghost buffer 0; buffer of epoch e = ghost0 + (e % nbuf) * stride
    int64_t buf_stride;
    int nbuf;
    uint32_t first_wait_reader;                // reader index of the first waiting workgroup (= the plan's push blocks)
};

// Called by EVERY thread of a workgroup (it contains a barrier).  One lane polls with relaxed
// system-scope loads (no fence per poll), then ONE system-scope acquire drops this CU's stale lines;
// the other waves read the ghosts only after the barrier (MI355X visibility rules: the acquire is per
// CU, the barrier holds the other waves until it has completed).
// Returns the INDEX of the ghost buffer of the exchange waited for (epoch % nbuf) in the low bits, as a
// wave-uniform value (readfirstlane: the caller's pointer arithmetic then stays in scalar registers), and
// HALO_WAIT_TIMED_OUT in bit 31 when the spin gave up: the caller must then POISON what it would have computed
// from the ghosts (NaN), so that an expired wait can never pass as a result -- the reference's MPI exchange
// would block instead (src/vectors.jl:446).  `reader` = this workgroup's index among the exchange's epoch readers.
constexpr uint32_t HALO_WAIT_TIMED_OUT = 0x80000000u;

__device__ __forceinline__ uint32_t halo_wait_block(const HaloWait &w, uint32_t reader)
{
    __shared__ uint32_t s_buf;
    if (threadIdx.x == 0) {
        const uint64_t epoch = epoch_current(w.er);
        const int64_t t0 = (int64_t)wall_clock64();
        bool ok = true;
        for (int i = 0; i < w.n_flags && ok; ++i) {
            const uint64_t *f = w.flags + (int64_t)i * WIN_FLAG_STRIDE_U64;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
                // A DEAD plan drains at once: the status word is sticky, so once any wait of this plan has expired
                // every later wait (the other boundary workgroups of this launch, the k iterations a CG call has
                // already enqueued behind it) gives up on its first failed poll instead of spinning out its own full
                // bound -- k enqueued steps cost one timeout, not k.  Only read on the slow path (a flag not there yet).
                if (__hip_atomic_load(w.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                if ((int64_t)wall_clock64() - t0 > w.timeout_ticks) {
                    __hip_atomic_store(w.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);           // system scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_buf = (uint32_t)(epoch % (uint64_t)w.nbuf) | (ok ? 0u : HALO_WAIT_TIMED_OUT);
        epoch_release(w.er, epoch, reader);                // the number is not needed any more (the DATA is:
    }                                                      // the producers learn that from the next step's acks)
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(s_buf);
}

__device__ __forceinline__ double halo_poison() { return __builtin_nan(""); }

// ---- producer side -----------------------------------------------------------------------------
// One send neighbour of a plan, as the push code sees it (built at attach time, window.hip).
struct PushTarget {
    double *ghost;          // peer's ghost buffer 0 at my segment (peer memory, mapped here)
    int64_t buf_stride;     // doubles between the peer's two buffers
    uint64_t *flag;         // peer's flag line for me
    const uint64_t *ack;    // LOCAL ack line this peer writes: last epoch it has finished reading
    int64_t count;          // entries (x width doubles)
    int64_t src_off;        // offset of this neighbour's run in send_idx
    int64_t first;          // first index of x when the run is contiguous, else -1
    int32_t nbuf;           // the peer's ghost buffer count
    int32_t nchunks;        // workgroups that push to this peer
};

struct PushArgs {
    const double *x;
    const void *idx;                    // send indices (int32 or int64, the plan's type)
    const PushTarget *targets;
    const int32_t *map;                 // [n_blocks][2]: (target or -1, chunk)
    uint64_t *const *ack_out;           // where this rank's acks go (peer memory), one per recv neighbour
    int n_ack_out;
    uint64_t *arrive;                   // local arrival counters, one per send neighbour
    uint32_t *status;
    EpochRef er;                        // the plan's step counter (device memory)
    int w;                              // doubles per index
    int64_t timeout_ticks;
    int n_blocks;                       // push workgroups (>= 1 whenever the plan has neighbours)
};

__device__ __forceinline__ bool spin_until_ge(const uint64_t *word, uint64_t want, int64_t t0, int64_t timeout,
                                              uint32_t *status)
{
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        // sticky status already set (an earlier wait of this plan / communicator expired): give up at once, so that
        // work enqueued behind a dead peer drains in ONE timeout (read on the slow path only)
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        __builtin_amdgcn_s_sleep(2);
        if ((int64_t)wall_clock64() - t0 > timeout) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// Work of push workgroup `b` (all threads of the workgroup call it; NT = threads per workgroup).
// Block 0 first publishes this rank's acks ("I have finished reading epoch-1": true in stream order, the
// consumer of epoch-1 precedes this launch).  Then: wait until the peer has released the buffer of this
// epoch, store the chunk with system-scope write-through stores, drain, and let the LAST chunk of the
// neighbour publish the epoch in the peer's flag line.
template <typename I, int NT>
__device__ __forceinline__ void halo_push_block(const PushArgs &a, int b)
{
    const uint64_t epoch = epoch_current(a.er);        // every lane reads it (one request per wave); released below
    if (b == 0)
        for (int j = threadIdx.x; j < a.n_ack_out; j += NT)
            __hip_atomic_store(a.ack_out[j], epoch - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int t = a.map[2 * b], c = a.map[2 * b + 1];
    if (t < 0) {                                       // acks only (no send neighbour)
        __syncthreads();
        if (threadIdx.x == 0) epoch_release(a.er, epoch, (uint32_t)b);
        return;
    }
    const PushTarget T = a.targets[t];
    // the consumer must have released the buffer this epoch overwrites.  If that wait EXPIRES the chunk is NOT
    // stored and the epoch is NOT published: the consumer may still be reading the buffer, and its own wait for
    // this epoch then expires too and poisons its result (halo_wait_block) -- a timeout can corrupt nothing.
    __shared__ uint32_t s_ack_ok;
    if (threadIdx.x == 0)
        s_ack_ok = (epoch > (uint64_t)T.nbuf)
                       ? (spin_until_ge(T.ack, epoch - (uint64_t)T.nbuf, (int64_t)wall_clock64(), a.timeout_ticks, a.status) ? 1u : 0u)
                       : 1u;
    __syncthreads();
    if (!s_ack_ok) {                                   // workgroup-uniform
        // (the arrival counter is NOT bumped: no later chunk of this or any later epoch can then be "the last
        //  one", so this neighbour never sees a flag from this plan again -- the plan is dead, status says so)
        if (threadIdx.x == 0) epoch_release(a.er, epoch, (uint32_t)b);
        return;
    }
    double *dst = T.ghost + (int64_t)(epoch % (uint64_t)T.nbuf) * T.buf_stride;
    const I *idx = reinterpret_cast<const I *>(a.idx);
    const int w = a.w;
    const int64_t total = T.count * w;
    int64_t per = (total + T.nchunks - 1) / T.nchunks;
    per += per & 1;                                    // even chunk starts: a 16-byte pair never straddles two chunks
    const int64_t lo = (int64_t)c * per < total ? (int64_t)c * per : total;
    const int64_t hi = lo + per < total ? lo + per : total;
    if (T.first >= 0) {
        const double *src = a.x + T.first * w;
        // contiguous run (every slab partition): 16-byte write-through stores where source and destination
        // are 16-byte aligned TOGETHER (an 8-byte store is one fabric write per lane and costs ~2.7x per byte,
        // MI355X_MICROARCH.md visibility table); the ragged ends and the misaligned case go 8 bytes at a time
        const bool pair = (((reinterpret_cast<uintptr_t>(src + lo) ^ reinterpret_cast<uintptr_t>(dst + lo)) & 15) == 0);
        int64_t i0 = lo, i1 = hi;
        if (pair) {
            if ((reinterpret_cast<uintptr_t>(dst + lo) & 15) != 0 && lo < hi) i0 = lo + 1;     // one leading double
            i1 = i0 + ((hi - i0) & ~(int64_t)1);
            if (threadIdx.x == 0 && i0 > lo)
                __hip_atomic_store(dst + lo, src[lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int64_t i = i0 + 2 * (int64_t)threadIdx.x; i < i1; i += 2 * NT) {
                typedef double v2d __attribute__((ext_vector_type(2)));
                const v2d v = *reinterpret_cast<const v2d *>(src + i);
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(v) : "memory");
            }
            if (threadIdx.x == 0 && i1 < hi)
                __hip_atomic_store(dst + i1, src[i1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            for (int64_t i = lo + threadIdx.x; i < hi; i += NT)
                __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else if (w == 1) {
        // gathered values: the DESTINATION is contiguous, so two gathered values leave as one 16-byte store
        // (chunk starts are even; only a misaligned ghost segment falls back to 8-byte stores)
        if ((reinterpret_cast<uintptr_t>(dst + lo) & 15) == 0) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const int64_t hi2 = lo + ((hi - lo) & ~(int64_t)1);
            for (int64_t i = lo + 2 * (int64_t)threadIdx.x; i < hi2; i += 2 * NT) {
                v2d v;
                v.x = a.x[(int64_t)idx[T.src_off + i]];
                v.y = a.x[(int64_t)idx[T.src_off + i + 1]];
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(v) : "memory");
            }
            if (threadIdx.x == 0 && hi2 < hi)
                __hip_atomic_store(dst + hi2, a.x[(int64_t)idx[T.src_off + hi2]], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            for (int64_t i = lo + threadIdx.x; i < hi; i += NT)
                __hip_atomic_store(dst + i, a.x[(int64_t)idx[T.src_off + i]], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else if ((w & 1) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(a.x)) & 15) == 0) {
        // gathered ROWS of an even number of doubles (SpMM ghost rows): a row is contiguous on both sides and
        // 16-byte aligned, so it travels as 16-byte write-through stores like a contiguous run
        for (int64_t i = lo + 2 * (int64_t)threadIdx.x; i < hi; i += 2 * NT) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const int64_t e = i / w;
            const int cc = (int)(i - e * w);
            const v2d v = *reinterpret_cast<const v2d *>(a.x + (int64_t)idx[T.src_off + e] * w + cc);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(v) : "memory");
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += NT) {
            const int64_t e = i / w;
            const int cc = (int)(i - e * w);
            __hip_atomic_store(dst + i, a.x[(int64_t)idx[T.src_off + e] * w + cc], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // every storing wave drains its write-through stores, then ONE lane publishes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = true;
        if (T.nchunks > 1) {
            const uint64_t old = __hip_atomic_fetch_add(a.arrive + t, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (old + 1 == epoch * (uint64_t)T.nchunks);
        }
        if (last) {
            __atomic_thread_fence(__ATOMIC_RELEASE);             // system scope
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(T.flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        epoch_release(a.er, epoch, (uint32_t)b);           // push workgroups are readers 0 .. n_blocks-1
    }
}

}  // namespace hpcla
