"""gmg_score_reads_strings (include/gmg.h): whole-read Score_String of every read and of its reverse complement under
many ICMs (Phymm's scoreReadsGlim.pl / simple-score; BASELINE configs[3]) against
  * the reference's own committed outputs sample-run/glimmer-mg/results/icm-{0..5}.scores.tmp (tests/golden/data/),
  * the oracle on ragged random reads (reads shorter than the window, batch tails, periodicity-1 and periodicity-3
    models in one call).  Sums must be bit-identical doubles."""
import os

import numpy as np
import pytest

from conftest import DATA

pytestmark = pytest.mark.gpu


def revcomp(s):
    return s[::-1].translate(str.maketrans("acgt", "tgca"))


def test_scores_tmp_of_the_reference_all_six_icms_in_one_call(gpu, seqs_fa):
    """icm-N.scores.tmp: "<read> <Score_String of the read>" printed with %.4f by simple-score (forward strand only in
    the committed files; the reverse-complement column is checked against the oracle below)"""
    reads = gpu.Reads.from_strings(seqs_fa[1])
    models = [gpu.Icm.open(os.path.join(DATA, "cluster-%d.icm" % i)) for i in range(6)]
    got = gpu.score_reads_strings(models, reads)
    assert got.shape == (6, 999, 2)
    n = 0
    for i in range(6):
        for r, line in enumerate(open(os.path.join(DATA, "icm-%d.scores.tmp" % i))):
            f = line.split()
            assert "%.4f" % got[i, r, 0] == f[1].strip()
            n += 1
    assert n == 6 * 999


def test_strings_vs_oracle_on_ragged_reads_mixed_models(gpu, oracle):
    rng = np.random.default_rng(31)
    lengths = [int(x) for x in rng.integers(1, 900, size=200)] + [1, 2, 10, 11, 12, 13, 500, 3000]
    seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]
    reads = gpu.Reads.from_strings(seqs)
    names = ["cluster-0.icm", "NC_000915.icm", "cluster-4.icm", "seqs.cluster-4.run1.filt.gicm"]
    got = gpu.score_reads_strings([gpu.Icm.open(os.path.join(DATA, n)) for n in names], reads)
    for k, name in enumerate(names):
        m = oracle.read(os.path.join(DATA, name))
        for r, s in enumerate(seqs):
            assert got[k, r, 0] == oracle.score_string(m, s, 0), (name, r)
            assert got[k, r, 1] == oracle.score_string(m, revcomp(s), 0), (name, r)


def test_strings_full_size_slice_properties(gpu, oracle):
    """200k x 500 bp (the configs[3] shape per model pass): determinism, locality (a read alone scores the same),
    sampled reads equal the oracle, strand symmetry (scoring the reverse complement swaps the two columns)"""
    n, L = 200_000, 500
    packed, off = gpu.synth.packed_reads(n, L, 5)
    reads = gpu.Reads(packed, off)
    models = [gpu.Icm.open(os.path.join(DATA, "cluster-%d.icm" % i)) for i in (1, 3)]
    a = gpu.score_reads_strings(models, reads)
    assert np.array_equal(a, gpu.score_reads_strings(models, reads))
    rng = np.random.default_rng(2)
    sample = [0, 1, n - 1, n - 2, n - 3, n - 4] + [int(x) for x in rng.integers(0, n, 30)]
    strs = [gpu.synth.unpack_ascii(packed, r * L, L).decode() for r in sample]
    alone = gpu.score_reads_strings(models, gpu.Reads.from_strings(strs + [revcomp(s) for s in strs]))
    for k, i in enumerate((1, 3)):
        m = oracle.read(os.path.join(DATA, "cluster-%d.icm" % i))
        for j, (r, s) in enumerate(zip(sample, strs)):
            assert a[k, r, 0] == alone[k, j, 0] == oracle.score_string(m, s, 0)
            assert a[k, r, 1] == alone[k, j, 1] == oracle.score_string(m, revcomp(s), 0)
            assert (alone[k, len(strs) + j, 0], alone[k, len(strs) + j, 1]) == (a[k, r, 1], a[k, r, 0])


def test_strings_edge_batches(gpu, oracle):
    m = gpu.Icm.open(os.path.join(DATA, "cluster-2.icm"))
    om = oracle.read(os.path.join(DATA, "cluster-2.icm"))
    assert gpu.score_reads_strings([], gpu.Reads.from_strings(["acgt"])).shape == (0, 1, 2)
    seqs = ["a", "ac", "acgtacgtacg", "acgtacgtacgt", "t" * 2047, "g" * 2048, "c" * 2049]      # around the window and the chunk size
    got = gpu.score_reads_strings([m, m], gpu.Reads.from_strings(seqs))
    for r, s in enumerate(seqs):
        assert got[0, r, 0] == got[1, r, 0] == oracle.score_string(om, s, 0)
        assert got[0, r, 1] == got[1, r, 1] == oracle.score_string(om, revcomp(s), 0)


def test_fused_sums_equal_the_two_pass_sums_and_the_oracle_on_ragged_long_reads(gpu, oracle):
    """reads of 86 .. 1,400 bases (the fused form: sums inside the main pass, order of the additions proven irrelevant per
    read) against the two-pass form (one running sum per string in string order) and the oracle; read boundaries fall at
    every phase of the kernel's 128-base spans and 32,768-base rounds"""
    rng = np.random.default_rng(77)
    lengths = [86, 87, 127, 128, 129, 255, 256, 257, 1400, 2048, 2049, 4097] + [int(x) for x in rng.integers(86, 1400, size=700)]
    seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]
    reads = gpu.Reads.from_strings(seqs)
    names = ["cluster-0.icm", "cluster-3.icm", "cluster-5.icm"]
    models = [gpu.Icm.open(os.path.join(DATA, n)) for n in names]
    with gpu.option("strings_fused", 1):
        fused = gpu.score_reads_strings(models, reads)
    with gpu.option("strings_fused", 0):
        two_pass = gpu.score_reads_strings(models, reads)
    assert fused.tobytes() == two_pass.tobytes()
    for k, name in enumerate(names):
        m = oracle.read(os.path.join(DATA, name))
        for r in list(range(12)) + [int(x) for x in rng.integers(0, len(seqs), 40)]:
            assert fused[k, r, 0] == oracle.score_string(m, seqs[r], 0), (name, r)
            assert fused[k, r, 1] == oracle.score_string(m, revcomp(seqs[r]), 0), (name, r)


def test_fused_sums_of_uniform_batches_equal_the_two_pass_sums_for_every_read(gpu, oracle):
    """batches of ONE read length (the kernel then finds the reads by arithmetic; from 192 bases on every 128-base span takes the
    single path with row scans and the share of the read that ends inside a row moved over): lengths around that switch, the span,
    the chunk and the round sizes, every read against the two-pass form, some against the oracle"""
    rng = np.random.default_rng(78)
    names = ["cluster-1.icm", "cluster-4.icm"]
    models = [gpu.Icm.open(os.path.join(DATA, n)) for n in names]
    om = oracle.read(os.path.join(DATA, names[0]))
    for L in (86, 127, 128, 129, 191, 192, 193, 199, 200, 255, 256, 257, 333, 500, 501, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 4099):
        n = max(70, 140_000 // L)
        codes = rng.integers(0, 4, size=(n, L))
        seqs = ["".join("acgt"[c] for c in row) for row in codes]
        reads = gpu.Reads.from_strings(seqs)
        with gpu.option("strings_fused", 1):
            fused = gpu.score_reads_strings(models, reads)
        with gpu.option("strings_fused", 0):
            two_pass = gpu.score_reads_strings(models, reads)
        assert fused.tobytes() == two_pass.tobytes(), L
        for r in (0, 1, n // 2, n - 1):
            assert fused[0, r, 0] == oracle.score_string(om, seqs[r], 0), (L, r)
            assert fused[0, r, 1] == oracle.score_string(om, revcomp(seqs[r]), 0), (L, r)


def test_fused_form_refuses_models_whose_values_could_make_the_order_matter(gpu, oracle, tmp_path):
    """a model with a probability so close to 1 that its logarithm is tiny (and one with a zero probability: -FLT_MAX):
    the first takes the two-pass form from the start, reads that meet the second are recomputed in string order --
    either way the sums are the oracle's"""
    src = os.path.join(DATA, "cluster-2.icm")
    raw = bytearray(open(src, "rb").read())
    rec0 = 150 + 24                                     # first record: int32 id, 4 floats, int16 mip
    tiny = np.frombuffer(raw, "<f4", 4, rec0 + 4).copy()
    tiny[1] = np.float32(-3.0e-7)                       # exponent field 105: below what the fused form takes
    raw[rec0 + 4:rec0 + 20] = tiny.tobytes()
    p1 = tmp_path / "tiny.icm"
    p1.write_bytes(bytes(raw))
    raw2 = bytearray(open(src, "rb").read())
    zero = np.frombuffer(raw2, "<f4", 4, rec0 + 4).copy()
    zero[2] = np.float32(-3.4028234663852886e38)        # log of a zero probability (icm.cc:1345-1349)
    raw2[rec0 + 4:rec0 + 20] = zero.tobytes()
    p2 = tmp_path / "zero.icm"
    p2.write_bytes(bytes(raw2))
    rng = np.random.default_rng(5)
    seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=int(n))) for n in rng.integers(100, 600, size=300)]
    reads = gpu.Reads.from_strings(seqs)
    for path in (p1, p2):
        got = gpu.score_reads_strings([gpu.Icm.open(str(path))], reads)[0]
        m = oracle.read(str(path))
        for r, s_ in enumerate(seqs):
            assert got[r, 0] == oracle.score_string(m, s_, 0) and got[r, 1] == oracle.score_string(m, revcomp(s_), 0), (path.name, r)


def test_zero_probability_leaves_on_uniform_batches(gpu, oracle, tmp_path):
    """-FLT_MAX (the logarithm of a zero probability, icm.cc:1345-1349) in LEAF rows, batches of ONE read length >= 192: the
    uniform path of the fused form moves the share of a read that ends inside a row of 16 lanes through the next read's
    accumulator, which a value of that size would swamp -- such a model (exponent range > 23) must take the two-pass form.
    Every read against the oracle, and the fused switch changes nothing."""
    src = os.path.join(DATA, "cluster-2.icm")
    raw = bytearray(open(src, "rb").read())
    n_rec = (len(raw) - 174 - 4) // 22
    rng = np.random.default_rng(44)
    hit = 0
    for r in rng.permutation(n_rec)[:3000]:
        off = 174 + 22 * int(r)
        (nid,) = np.frombuffer(raw, "<i4", 1, off)
        if nid >= 5461:                                 # a leaf of the depth-7 tree
            v = np.frombuffer(raw, "<f4", 4, off + 4).copy()
            v[int(rng.integers(0, 4))] = np.float32(-3.4028234663852886e38)
            raw[off + 4:off + 20] = v.tobytes()
            hit += 1
    assert hit > 1000
    path = tmp_path / "zero_leaves.icm"
    path.write_bytes(bytes(raw))
    m, om = gpu.Icm.open(str(path)), oracle.read(str(path))
    for L in (192, 250, 500):
        codes = rng.integers(0, 4, size=(400, L))
        seqs = ["".join("acgt"[c] for c in row) for row in codes]
        reads = gpu.Reads.from_strings(seqs)
        with gpu.option("strings_fused", 1):
            a = gpu.score_reads_strings([m], reads)[0]
        with gpu.option("strings_fused", 0):
            b = gpu.score_reads_strings([m], reads)[0]
        assert a.tobytes() == b.tobytes()
        n_huge = 0
        for r, s_ in enumerate(seqs):
            want = (oracle.score_string(om, s_, 0), oracle.score_string(om, revcomp(s_), 0))
            assert (a[r, 0], a[r, 1]) == want, (L, r)
            n_huge += int(want[0] < -1e38) + int(want[1] < -1e38)
        assert n_huge > 20                              # reads that met a zero probability, and their neighbours, are all right


def test_configs3_shape_ten_million_reads_64_models_in_pieces(gpu, oracle, tmp_path):
    """BASELINE configs[3]: 10M reads x 64 Phymm ICMs.  Ten pieces of 1M x 500 bp, 64 DISTINCT period-1 models per call (SURVEY 8d:
    the six sample-run ICMs + 58 trained on disjoint slices of NC_000915.fna, tests/models64.py): determinism (first piece twice),
    no two models agree on a read, sampled (read, model) pairs of every piece -- incl. the last read of the job and every one of the
    64 models at least once -- equal the oracle reading the same .icm files."""
    import models64
    n_piece, L, pieces = 1_000_000, 500, 10
    pairs = models64.period1_models(gpu, tmp_path, 64)
    models = [m for m, _ in pairs]
    o_models = [oracle.read(p) for _, p in pairs]
    rng = np.random.default_rng(9)
    seen = set()
    for pc in range(pieces):
        packed, off = gpu.synth.packed_reads_range(pc * n_piece * L, n_piece * L, L, 41)
        reads = gpu.Reads(packed, off)
        got = gpu.score_reads_strings(models, reads)
        assert got.shape == (64, n_piece, 2)
        if pc == 0:
            assert got.tobytes() == gpu.score_reads_strings(models, reads).tobytes()
            assert len({got[k, 0, 0] for k in range(64)}) == 64          # 64 different tables
        ks = [int(x) for x in rng.integers(0, 64, 3)] + [k for k in range(pc * 7, min(pc * 7 + 7, 64))]
        for r, k in zip([0, n_piece - 1] + [int(x) for x in rng.integers(0, n_piece, len(ks) - 2)], ks):
            s_ = gpu.synth.unpack_ascii(packed, r * L, L).decode()
            assert got[k, r, 0] == oracle.score_string(o_models[k], s_, 0)
            assert got[k, r, 1] == oracle.score_string(o_models[k], revcomp(s_), 0)
            seen.add(k)
        del reads, got
    assert len(seen) == 64
