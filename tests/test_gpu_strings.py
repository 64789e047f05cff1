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
