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


def read_genome_codes(fasta_path):
    """the bases of a one-record FASTA file as 2-bit codes after the reference's load-time filter (tolower (Filter (ch)),
    src/Common/gene.cc:1139-1175: a/c/g/t kept, the IUPAC letters mapped as there, anything else -> c)"""
    lut = np.full(256, 1, np.uint8)                                  # everything else -> 'c'
    for ch, code in (("a", 0), ("c", 1), ("g", 2), ("t", 3), ("r", 2), ("y", 1), ("s", 1), ("w", 3), ("m", 1), ("k", 3),
                     ("b", 1), ("d", 2), ("h", 1), ("v", 1)):
        lut[ord(ch)] = lut[ord(ch.upper())] = code
    raw = b"".join(line.strip() for line in open(fasta_path, "rb") if not line.startswith(b">"))
    return lut[np.frombuffer(raw, np.uint8)]


def genome_reads(fasta_path, n_reads, L, seed, first_read=0):
    """SURVEY.md 8(d)'s second input distribution: reads [first_read, first_read + n_reads) of a job whose read r is L bases cut
    from the genome at a start drawn uniformly (SplitMix64 output r + 1 of `seed`), bit 63 of the same draw choosing the strand
    (1: the reverse complement).  -> (packed, offsets) in the include/gmg.h layout."""
    g = read_genome_codes(fasta_path)
    G = len(g)
    assert G > L
    both = np.concatenate([g, (3 - g)[::-1]])                        # strand 1 at [G, 2G)
    with np.errstate(over="ignore"):
        z = _mix64(np.uint64(seed) + np.arange(first_read + 1, first_read + n_reads + 1, dtype=np.uint64) * GAMMA)
    start = ((z & np.uint64((1 << 62) - 1)) % np.uint64(G - L + 1)).astype(np.int64) + (z >> np.uint64(63)).astype(np.int64) * G
    total = int(n_reads) * int(L)
    packed = np.zeros((total + 15) // 16 + 2, np.uint32)
    shifts = (2 * np.arange(16, dtype=np.uint32))[None, :]
    step = max(1, (1 << 24) // L // 16 * 16)                          # reads per piece (a multiple of 16: pieces end on word boundaries when L is odd too)
    carry = np.zeros(0, np.uint8)
    w = 0
    for r0 in range(0, n_reads, step):
        r1 = min(n_reads, r0 + step)
        codes = both[(start[r0:r1, None] + np.arange(L, dtype=np.int64)[None, :]).ravel()]
        codes = np.concatenate([carry, codes])
        nfull = len(codes) // 16
        packed[w:w + nfull] = (codes[:nfull * 16].reshape(-1, 16).astype(np.uint32) << shifts).sum(axis=1, dtype=np.uint32)
        w += nfull
        carry = codes[nfull * 16:]
    if len(carry):
        packed[w] = int((carry.astype(np.uint64) << (2 * np.arange(len(carry), dtype=np.uint64))).sum())
    off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
    return packed, off
