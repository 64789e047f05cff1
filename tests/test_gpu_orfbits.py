"""k_mg_find_orfs_bits (glimmer-mg_amd/csrc/gmg_mg_orfbits.h; option mg_orfs_bits = 1): Find_Orfs on bit masks, a wave per window of
reads, six lanes per read -- against the oracle (glimmer_base.cc:638-817 restated) and against the one-lane-per-read kernels it replaces (option
mg_orfs_bits = 0), record by record: the Orf_t fields, Score_Orf_Starts' bounds, the order inside a read, and (through
gmg_mg_score_reads) the number of starts the write pass counts for every ORF."""
import os

import numpy as np
import pytest

from conftest import DATA

pytestmark = pytest.mark.gpu


def random_reads(rng, lengths, at=0.25):
    p = [at, 0.5 - at, 0.5 - at, at]                     # a, c, g, t
    return ["".join("acgt"[c] for c in rng.choice(4, size=int(n), p=p)) for n in lengths]


def special_reads():
    return ["acg" * 200,                                 # no stop codon in any forward frame
            "taa" * 100 + "a",                           # stops back to back
            "tta" * 120,                                 # reverse stops back to back
            "atg" + "gct" * 150 + "taa" + "cc",          # one clean forward gene
            "gg" + "tta" + "agc" * 150 + "cat" + "g",    # its reverse-strand twin
            "atg" * 300,                                 # start codons only
            "cat" * 341 + "c",                           # reverse start codons only, 1,024 bases
            "a" * 1024, "t" * 33, "g" * 32,
            "atgtaa" * 100, "ttacat" * 100, "atgtga" * 5 + "gct" * 60 + "tag" + "ctattaatgcat" * 20]


BATCHES = {
    # lengths: what the windows are cut by
    "uniform_500": lambda rng: random_reads(rng, [500] * 203),
    "uniform_96": lambda rng: random_reads(rng, [96] * 500),
    "uniform_1024": lambda rng: random_reads(rng, [1024] * 41),
    "uniform_33": lambda rng: random_reads(rng, [33] * 300),
    "ragged": lambda rng: random_reads(rng, list(rng.integers(32, 700, size=400))) + special_reads(),
    "ragged_with_tiny_reads": lambda rng: random_reads(rng, [int(x) for x in rng.choice([1, 2, 3, 5, 8, 17, 31, 32, 33, 64, 75, 76, 200, 400, 1024],
                                                                                          size=600)]),
    "word_edges": lambda rng: random_reads(rng, [32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 1023, 1024, 1022, 31, 30, 1, 992, 993, 994] * 6),
    "at_rich": lambda rng: random_reads(rng, list(rng.integers(100, 900, size=200)), at=0.4),       # stop codons every few codons
    "gc_rich": lambda rng: random_reads(rng, list(rng.integers(100, 1024, size=200)), at=0.08),     # ORFs that span whole reads
}
OPTIONS = {
    "default": dict(),
    "truncated": dict(allow_truncated=True),
    "g90": dict(min_gene_len=90, allow_truncated=True),
    "g32": dict(min_gene_len=32, allow_truncated=True),
    "sets": dict(min_gene_len=45, allow_truncated=True, start_codons=("atg", "rtg", "ttg", "ctg"), stop_codons=("taa", "tag")),
    "start_is_stop": dict(min_gene_len=36, allow_truncated=True, start_codons=("atg", "tga", "tta"), stop_codons=("taa", "tga")),
}


def fields(orfs):
    return np.stack([orfs[k].astype(np.int64) for k in ("read", "frame", "stop_position", "orf_len", "gene_len", "lo", "hi")], 1)


@pytest.mark.parametrize("opt", sorted(OPTIONS))
@pytest.mark.parametrize("batch", sorted(BATCHES))
def test_find_orfs_bits_equals_the_oracle_and_the_per_read_kernel(gpu, oracle, batch, opt):
    rng = np.random.default_rng(sum(map(ord, batch + opt)))
    seqs = BATCHES[batch](rng)
    kw = OPTIONS[opt]
    reads = gpu.Reads.from_strings(seqs)
    with gpu.option("mg_orfs_bits", 1):
        orfs, off = gpu.find_orfs(reads, **kw)
    with gpu.option("mg_orfs_bits", 0):
        orfs0, off0 = gpu.find_orfs(reads, **kw)
    assert np.array_equal(off, off0)
    assert np.array_equal(fields(orfs), fields(orfs0))
    prm = oracle.mg_params(**dict(dict(allow_truncated=False), **kw))
    n = 0
    for r, seq in enumerate(seqs):
        want = oracle.find_orfs(seq, prm)
        mine = orfs[int(off[r]):int(off[r + 1])]
        got = np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1).reshape(-1, 4)
        assert np.array_equal(got, want), "Find_Orfs differs on read %d (len %d)" % (r, len(seq))
        n += len(want)
    assert n > 0 or max(map(len, seqs)) < kw.get("min_gene_len", 75)


@pytest.mark.parametrize("mode", ["default", "indel", "sub", "g90"])
def test_front_half_on_the_bit_mask_finder_equals_the_per_read_kernels(gpu, mode):
    """gmg_mg_score_reads end to end: the ORF records with their start counts (count_starts), start lists and verdicts"""
    rng = np.random.default_rng(77)
    seqs = random_reads(rng, list(rng.integers(40, 900, size=3000))) + special_reads()
    reads = gpu.Reads.from_strings(seqs)
    nc = gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))
    kw = dict(indel=dict(allow_indels=True), sub=dict(allow_subs=True), g90=dict(min_gene_len=90, ignore_score_len=150)).get(mode, {})
    with gpu.option("mg_orfs_bits", 1):
        a = gpu.mg_score_reads(nc, gpu.Icm.indep(0.45), reads, **kw)
    with gpu.option("mg_orfs_bits", 0):
        b = gpu.mg_score_reads(nc, gpu.Icm.indep(0.45), reads, **kw)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes()
    assert len(a[0]) > 1000


def test_a_million_bases_of_short_and_long_reads(gpu):
    """many windows, reads of every length up to the kernel's limit, both window shapes; against the per-read kernel"""
    rng = np.random.default_rng(5)
    for lengths in ([500] * 20000, list(rng.integers(1, 1025, size=20000)), [1024] * 3000, list(rng.integers(32, 100, size=50000))):
        codes = rng.integers(0, 4, size=int(np.sum(lengths)), dtype=np.uint8)
        offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
        reads = gpu.Reads.from_codes(codes, offs) if hasattr(gpu.Reads, "from_codes") else gpu.Reads.from_strings(
            ["".join("acgt"[c] for c in codes[int(a):int(b)]) for a, b in zip(offs[:-1], offs[1:])])
        for kw in (dict(allow_truncated=True), dict(min_gene_len=60)):
            with gpu.option("mg_orfs_bits", 1):
                orfs, off = gpu.find_orfs(reads, **kw)
            with gpu.option("mg_orfs_bits", 0):
                orfs0, off0 = gpu.find_orfs(reads, **kw)
            assert np.array_equal(off, off0) and np.array_equal(fields(orfs), fields(orfs0)) and len(orfs) > 1000


@pytest.mark.parametrize("opt", sorted(OPTIONS))
@pytest.mark.parametrize("batch", ["ragged", "ragged_with_tiny_reads", "word_edges", "uniform_500", "at_rich"])
def test_write_pass_forms_of_the_orf_scan_agree(gpu, batch, opt):
    """k_mg_find_orfs_ev with its events from register masks (mg_orfs_events = 2, the default), from the LDS queue (1), and
    k_mg_find_orfs<write> at every position (0): the same records, and through gmg_mg_score_reads the same start counts and lists"""
    rng = np.random.default_rng(sum(map(ord, batch + opt)) + 1)
    seqs = BATCHES[batch](rng) + ["a" * 40 + "atg" + "gct" * 40 + "taa", "tta" + "cat" * 30 + "c" * 33]
    kw = OPTIONS[opt]
    reads = gpu.Reads.from_strings(seqs)
    nc = gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))
    stops = kw.get("stop_codons", ("taa", "tag", "tga"))
    got = {}
    for form in (2, 1, 0):
        with gpu.option("mg_orfs_events", form):
            orfs, off = gpu.find_orfs(reads, **kw)
            full = gpu.mg_score_reads(nc, gpu.Icm.indep(0.45, stops), reads, **kw)
        got[form] = (fields(orfs).tobytes(), off.tobytes(), tuple(x.tobytes() for x in full))
    assert got[2] == got[1] == got[0]
