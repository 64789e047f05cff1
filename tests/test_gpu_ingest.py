"""FASTA ingest on the device (include/gmg.h: gmg_fasta_ingest) against the reference's Fasta_Read
(tests/golden/nasty.fasta_read.txt, seqs.fasta_read.sha256: dumps of the real reference) and against the oracle's
restatement on seeded random byte soup.  Records, headers, every base after tolower (Filter (ch)), offsets and the
g/c count must be identical."""
import hashlib
import os

import numpy as np
import pytest

from conftest import DATA, GOLD
from test_oracle_fasta import dump

pytestmark = pytest.mark.gpu


def device_records(gpu, data):
    reads, headers, gc = gpu.Reads.from_fasta_bytes(data)
    packed, off = reads.download()
    assert len(off) == reads.n_reads + 1 == len(headers) + 1 and off[0] == 0 and off[-1] == reads.total_bases
    seqs = [gpu.synth.unpack_ascii(packed, int(off[r]), int(off[r + 1] - off[r])) for r in range(reads.n_reads)]
    return reads, list(zip(headers, seqs)), gc


def test_ingest_nasty_file_matches_reference(gpu):
    data = open(os.path.join(DATA, "nasty.fa"), "rb").read()
    _, records, gc = device_records(gpu, data)
    assert dump(records, gc) == open(os.path.join(GOLD, "nasty.fasta_read.txt"), "rb").read()


def test_ingest_seqs_fa_matches_reference_and_scores_like_the_host_path(gpu, seqs_fa):
    data = open(os.path.join(DATA, "seqs.fa"), "rb").read()
    reads, records, gc = device_records(gpu, data)
    want = open(os.path.join(GOLD, "seqs.fasta_read.sha256")).read().strip()
    assert hashlib.sha256(dump(records, gc)).hexdigest() == want
    assert abs(gc / reads.total_bases - float(np.load(os.path.join(GOLD, "frames_nc.npz"))["gc"])) < 1e-15
    nc, indep = gpu.Icm.open(os.path.join(DATA, "NC_000915.icm")), gpu.Icm.indep(0.5)
    a = gpu.frame_score6(nc, indep, reads)
    b = gpu.frame_score6(nc, indep, gpu.Reads.from_strings(seqs_fa[1]))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ingest_random_byte_soup_vs_oracle(gpu, oracle, seed):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"acgtACGTnNrRyYkKmM>>>\n\n\n\n\r\t  x*-0", np.uint8)
    pieces = []
    for _ in range(400):
        pieces.append(alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 200)))].tobytes())
        if rng.random() < 0.3:
            pieces.append(b">hdr %d  \n" % rng.integers(0, 1000) + b"acgt" * int(rng.integers(0, 300)))
    tail = [b"", b">", b">   ", b"> x", b"\n", b"acgt", b">h\n"][seed % 7]
    data = b"".join(pieces) + tail
    _, records, gc = device_records(gpu, data)
    want, want_gc = oracle.fasta_records(data)
    assert records == want and gc == want_gc and len(want) > 50


@pytest.mark.parametrize("seed", [11, 12])
def test_ingest_block_summaries_equal_the_scans_on_a_large_soup(gpu, seed):
    """the two device forms (block summaries -- the default -- and the first version's two scans over every byte, `ingest_scans` = 1) on
    ~6 MB of byte soup with records, headers and blank runs that straddle the 4 KiB blocks and the 16-byte lanes: same reads, same
    header extents, same g/c count; lengths that are no multiple of 16 or 4,096"""
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"acgtACGTacgtacgtnNrRyY>\n\n\r\t  x-", np.uint8)
    pieces = [b"junk in front of the first record\nacgt\n" if seed % 2 else b""]
    for _ in range(3000):
        pieces.append(alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 3000)))].tobytes())
        if rng.random() < 0.4:
            pieces.append(b">" + b" " * int(rng.integers(0, 40)) + b"hdr %d" % rng.integers(0, 10 ** 9) + b">" * int(rng.integers(0, 3)) + b"\n")
    data = b"".join(pieces) + [b">  ", b"acg"][seed % 2]
    out = []
    for scans in (0, 1):
        with gpu.option("ingest_scans", scans):
            out.append(device_records(gpu, data))
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2] and len(out[0][1]) > 500
    assert len(data) % 4096 not in (0, 1) and len(data) > 4_000_000
    # the chunked upload (inputs of 64 MiB and more): 16 pieces, each parsed and packed as it arrives, into arrays sized by a bound
    with gpu.option("ingest_piece_min", 1 << 20):
        piecewise = device_records(gpu, data)
    assert piecewise[1] == out[1][1] and piecewise[2] == out[1][2]


def test_ingest_default_path_at_chunked_size(gpu):
    """a 70 MB file with the default switches (>= 64 MiB: the 16-piece upload, every piece parsed and packed as it arrives) against the
    first version (two scans over every byte, one plain copy): the same packed words, offsets, header extents and g/c count"""
    rng = np.random.default_rng(23)
    n_reads, L = 134_000, 500
    bases = np.frombuffer(b"acgtacgtacgtacgn", np.uint8)[rng.integers(0, 16, size=(n_reads, L))]
    body = np.full((n_reads, L + 8), ord("\n"), np.uint8)
    body[:, np.arange(L) + np.arange(L) // 70] = bases
    hdr = np.frombuffer(b"".join(b">r%06d  x y\n" % i for i in range(n_reads)), np.uint8).reshape(n_reads, -1)
    data = np.ascontiguousarray(np.concatenate([hdr, body], axis=1).reshape(-1)).tobytes()
    assert len(data) > 64 << 20
    out = []
    for scans in (0, 1):
        with gpu.option("ingest_scans", scans):
            reads, headers, gc = gpu.Reads.from_fasta_bytes(data)
            packed, off = reads.download()
            out.append((packed.tobytes(), off.tobytes(), headers, gc, reads.n_reads))
    assert out[0] == out[1] and out[0][4] == n_reads and out[0][2][-1] == b"r%06d  x y" % (n_reads - 1)


def test_ingest_piecewise_edges(gpu, oracle):
    """the chunked upload on inputs built around its seams: a '>' whose blanks run across the end of a piece (the header's begin is
    completed when all bytes are there), across several pieces, up to the end of the file; more records than the bound the arrays
    were sized by (one per 16 bytes: the plain order takes over); a piece that ends inside a header line"""
    def run(data):
        with gpu.option("ingest_piece_min", 1):
            _, records, gc = device_records(gpu, data)
        want, want_gc = oracle.fasta_records(data)
        assert records == want and gc == want_gc
        return len(want)
    n = 16 * 8192
    for at in (8192 - 3, 8192 - 1, 8192, 3 * 8192 - 20):                      # (pieces of this input: 8,192 + 4,096 bytes)
        body = bytearray(b"acgtacgtacgtacg\n" * (n // 16))
        blanks = 40 if at != 3 * 8192 - 20 else 12_300 * 2                    # ... across two seams
        body[at:at + 1 + blanks + 4] = b">" + b" " * blanks + b"hdr\n"
        assert run(b">first\n" + bytes(body)) == 2
    assert run(b">a\nacgt\n" * 30_000) == 30_000                             # 8 bytes per record: beyond the bound
    assert run(b">x\n" + b"acgt" * 40_000 + b"\n>" + b" " * 5000) == 1         # "> <blanks> EOF" is no record
    assert run(b">" + b"h" * 100_000 + b"\nacgt\n>y\nggg") == 2                 # pieces inside a header line


def test_ingest_edge_inputs(gpu, oracle):
    for data in [b"", b"no record at all\nacgt\n", b">", b">only a header", b">h\n", b">a\nA", b"\n\n>  \n\n", b">x\nacgt>y\n>z"]:
        reads, records, gc = device_records(gpu, data)
        want, want_gc = oracle.fasta_records(data)
        assert records == want and gc == want_gc, data


def test_ingest_refuses_two_gibibytes_instead_of_wrapping(gpu):
    """the parser's byte counters are 32-bit on the device: a call with >= 2^31 - 1 bytes is REFUSED (GMG_EINVAL) before a single byte
    is looked at -- the length is a lie here, the buffer is 16 bytes -- and gmg_fasta_split gives the pieces that do fit"""
    import ctypes as C
    lib = gpu.capi.lib()
    buf = C.create_string_buffer(b">r\nacgtacgtacg\n", 16)
    reads, index = C.c_void_p(), C.c_void_p()
    for n in (2 ** 31 - 1, 2 ** 31, 2 ** 32 + 5, 2 ** 40):
        assert lib.gmg_fasta_ingest(buf, n, C.byref(reads), C.byref(index)) == -1          # GMG_EINVAL
        assert b"2^31" in lib.gmg_last_error() and not reads.value and not index.value
    # a 2.1 GiB "file" of one short record repeated: split into pieces of < 2^31 bytes each, cut at record starts
    rec = b">read\n" + b"acgtacgtgg" * 50 + b"\n"
    n_rec = (2 ** 31 + 2 ** 27) // len(rec)
    data = rec * n_rec
    cuts = np.zeros(16, np.uint64)
    pieces = lib.gmg_fasta_split(data, len(data), 2 ** 30, cuts.ctypes.data_as(C.c_void_p), 15)
    assert 2 <= pieces <= 15 and cuts[0] == 0 and cuts[pieces] == len(data)
    sizes = np.diff(cuts[:pieces + 1].astype(np.int64))
    assert sizes.max() < 2 ** 31 - 1 and all(data[int(c):int(c) + 1] == b">" for c in cuts[:pieces])
    total = 0
    for k in range(pieces):                             # every piece is a valid input of its own
        r, _, _ = gpu.Reads.from_fasta_bytes(data[int(cuts[k]):int(cuts[k + 1])])
        total += r.n_reads
        r.close()
    assert total == n_rec
