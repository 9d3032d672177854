"""Partitions and structural hashes (reference: src/HPCLinearAlgebra.jl:255-289, src/sparse.jl:97-127).

Partitions are boundary arrays of length nranks+1.  The reference is 1-based
(``p[1]=1, p[end]=n+1``); here they are 0-based (``p[0]=0, p[-1]=n``): rank r owns
``[p[r], p[r+1])``.  That is the only translation.

Hashes: the reference uses 256-bit Blake3 digests purely as memoization keys -- they are compared
for equality, never against constants (test/test_indexing.jl:149) -- so any stable 256-bit hash
serves; blake2b-256 from hashlib is used.
"""
from __future__ import annotations

import hashlib

import numpy as np


def uniform_partition(n: int, nranks: int) -> np.ndarray:
    """src/HPCLinearAlgebra.jl:279-289: the first ``n mod nranks`` ranks get ``n div nranks + 1``."""
    per_rank, remainder = divmod(int(n), int(nranks))
    part = np.empty(nranks + 1, dtype=np.int64)
    part[0] = 0
    for r in range(1, nranks + 1):
        part[r] = part[r - 1] + per_rank + (1 if r <= remainder else 0)
    return part


def compute_partition_hash(partition: np.ndarray) -> bytes:
    """src/HPCLinearAlgebra.jl:255-259."""
    return hashlib.blake2b(np.ascontiguousarray(partition, dtype=np.int64).tobytes(),
                           digest_size=32).digest()


def _update_with_len(h, arr: np.ndarray) -> None:
    arr = np.ascontiguousarray(arr)
    h.update(np.int64(arr.size).tobytes())     # length prefix (src/sparse.jl:100-110)
    h.update(memoryview(arr).cast("B"))


def compute_structural_hash(row_partition, col_indices, rowptr, colval, comm) -> bytes:
    """src/sparse.jl:97-121: hash(row_partition | col_indices | rowptr | colval) locally, Allgather
    the 32-byte digests, hash the concatenation."""
    from .backends import comm_allgather_bytes
    h = hashlib.blake2b(digest_size=32)
    _update_with_len(h, np.asarray(row_partition, dtype=np.int64))
    _update_with_len(h, np.asarray(col_indices, dtype=np.int64))
    _update_with_len(h, np.asarray(rowptr))      # native Ti: Ti is part of the cache key anyway
    _update_with_len(h, np.asarray(colval))
    local = h.digest()
    all_hashes = comm_allgather_bytes(comm, local)
    g = hashlib.blake2b(digest_size=32)
    for d in all_hashes:
        g.update(d)
    return g.digest()


_DIG_P1 = np.uint64(0x9E3779B97F4A7C15)
_DIG_P2 = np.uint64(0xD1B54A32D192ED03)
_DIG_S = (np.uint64(0x243F6A8885A308D3), np.uint64(0x13198A2E03707344), np.uint64(0xA4093822299F31D0),
          np.uint64(0x082EFA98EC4E6C89))


def array_digest(a: np.ndarray) -> bytes:
    """Host twin of ``hpcla_digest_*`` (csrc/construct.hip): four 64-bit words,
    ``sum_i mix(a[i]*P1 + (i+1)*P2 + S_k) mod 2^64``; order-sensitive, exact, so host-built and device-built
    structures get the same key."""
    a = np.ascontiguousarray(a)
    out = np.zeros(4, dtype=np.uint64)
    step = 1 << 22                                   # bounded temporaries
    with np.errstate(over="ignore"):
        for lo in range(0, a.size, step):
            blk = a[lo:lo + step].astype(np.int64).view(np.uint64)
            t = blk * _DIG_P1 + (np.arange(lo + 1, lo + 1 + blk.size, dtype=np.uint64)) * _DIG_P2
            for k in range(4):
                z = t + _DIG_S[k]
                z ^= z >> np.uint64(29)
                z *= np.uint64(0xBF58476D1CE4E5B9)
                z ^= z >> np.uint64(32)
                out[k] += z.sum(dtype=np.uint64)
    return out.tobytes()


def structural_hash_from_digests(row_partition, digests, comm) -> bytes:
    """compute_structural_hash (src/sparse.jl:97-121) with the three array passes replaced by their digests
    (``array_digest`` on the host or ``hpcla_digest_*`` on the device): local 32-byte hash, Allgather, hash
    of the concatenation."""
    from .backends import comm_allgather_bytes
    h = hashlib.blake2b(digest_size=32)
    _update_with_len(h, np.asarray(row_partition, dtype=np.int64))
    for n, d in digests:                             # (length, 32-byte digest) of col_indices, rowptr, colval
        h.update(np.int64(n).tobytes())
        h.update(d)
    g = hashlib.blake2b(digest_size=32)
    for d in comm_allgather_bytes(comm, h.digest()):
        g.update(d)
    return g.digest()


def owner_of(partition: np.ndarray, gidx: np.ndarray) -> np.ndarray:
    """``searchsortedlast(partition, idx) - 1`` clamped to nranks-1 (src/sparse.jl:1890-1894)."""
    nranks = len(partition) - 1
    own = np.searchsorted(partition, gidx, side="right") - 1
    return np.minimum(own, nranks - 1)
