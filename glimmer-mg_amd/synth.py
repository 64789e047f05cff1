"""Synthetic reads: the job is one stream of 2-bit bases; 64-bit word k of the
stream is SplitMix64 output k+1 of `seed`, base j of a word is (z >> 2j) & 3; read r is bases
[r*L, (r+1)*L).  oracle/ref_drivers/ref_bench.cc generates the same stream in C++."""
import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def packed_reads(n_reads, L, seed):
    """-> (packed uint32[ceil(total/16)+1], offsets uint64[n_reads+1]) in the include/gmg.h layout."""
    total = int(n_reads) * int(L)
    n64 = (total + 31) // 32
    with np.errstate(over="ignore"):
        k = np.arange(1, n64 + 1, dtype=np.uint64)
        z = _mix64(np.uint64(seed) + k * GAMMA)
    rem = total % 32
    if rem and n64:
        z[-1] &= np.uint64((1 << (2 * rem)) - 1)
    packed = np.zeros(2 * n64 + 2, np.uint32)
    packed[:2 * n64] = z.view(np.uint32)
    off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
    return packed, off


def packed_reads_range(first_base, n_bases, L, seed):
    """bases [first_base, first_base + n_bases) of the stream of `seed` as a batch of its own (both multiples of L):
    what one batch of a job too big for one call looks like"""
    assert first_base % L == 0 and n_bases % L == 0
    w0, sh = first_base // 32, 2 * (first_base % 32)
    n64 = (n_bases + 31) // 32
    with np.errstate(over="ignore"):
        k = np.arange(w0 + 1, w0 + n64 + 2, dtype=np.uint64)         # one word more: the batch may start inside a word
        z = _mix64(np.uint64(seed) + k * GAMMA)
    if sh:
        z = (z[:-1] >> np.uint64(sh)) | (z[1:] << np.uint64(64 - sh))
    else:
        z = z[:-1].copy()
    rem = n_bases % 32
    if rem and n64:
        z[-1] &= np.uint64((1 << (2 * rem)) - 1)
    packed = np.zeros(2 * n64 + 2, np.uint32)
    packed[:2 * n64] = z.view(np.uint32)
    off = np.arange(n_bases // L + 1, dtype=np.uint64) * np.uint64(L)
    return packed, off


def unpack_ascii(packed, first_base, n):
    """bases [first_base, first_base+n) of a packed stream as a lower-case acgt bytes object"""
    g = np.arange(first_base, first_base + n, dtype=np.int64)
    codes = (packed[g >> 4] >> (2 * (g & 15)).astype(np.uint32)) & np.uint32(3)
    return np.frombuffer(b"acgt", np.uint8)[codes].tobytes()
