"""Pins the oracle's restatement of the Score_Orfs inner loop (src/Glimmer/glimmer3.cc:1275-1552) to the
real reference: oracle/_ref/ref_orfs pulls the reference's glimmer3.cc in whole, runs its own Find_Orfs
and Score_Orfs on seqs.fa and dumps every start list handed to Add_Events_* (tests/golden/orfs_*.npz).
Exact equality, including the double scores.  CPU only."""
import os

import numpy as np
import pytest

from conftest import DATA, GOLD

CASES = {
    "orfs_default": dict(),
    "orfs_X": dict(allow_truncated=True),
    "orfs_g90_first": dict(min_gene_len=90, use_first_start=True),
}


# Ignore_Score_Len: glimmer3 derives it from the GC content (Set_Ignore_Score_Len, glimmer_base.cc); for these
# 500-bp reads it is far above every ORF length, so the default "never" (INT_MAX) reproduces the goldens.

@pytest.mark.parametrize("name", sorted(CASES))
def test_score_orfs_matches_reference_start_lists(oracle, seqs_fa, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    nc = oracle.read(os.path.join(DATA, "NC_000915.icm"))
    gc = float(np.load(os.path.join(GOLD, "frames_nc.npz"))["gc"])
    indep = oracle.indep(gc)
    reads = [oracle.filter_lower(s) for s in seqs_fa[1]]
    prm = oracle.orf_params(**CASES[name])
    accepted = {int(o): i for i, o in enumerate(g["gene_orf"])}
    n_checked = 0
    for oi, (r, frame, stop, ln) in enumerate(g["orfs"]):
        n, out, starts = oracle.score_orf(nc, indep, reads[r], int(frame), int(stop), int(ln), prm)
        if oi in accepted:
            gi = accepted[oi]
            assert n >= 0 and out.is_tentative_gene
            assert out.gene_score == g["gene_score"][gi]
            assert out.best_j + 1 == g["gene_len"][gi]
            b, cnt = int(g["gene_start_begin"][gi]), int(g["gene_nstarts"][gi])
            assert n == cnt
            for s, gi_row, gs in zip(starts, g["start_int"][b:b + cnt], g["start_score"][b:b + cnt]):
                assert (s.j, s.pos, s.which, s.truncated, s.first) == tuple(int(x) for x in gi_row)
                assert s.score == gs
            n_checked += 1
        else:
            assert n < 0 or not out.is_tentative_gene
    assert n_checked == len(g["gene_orf"]) > 200


# ---- Find_Orfs in full: ignore regions and circular sequences (glimmer_base.cc:638-817, 2793-2900) -----------------------
FIND_ORFS_GENERAL = {                                   # golden name -> (mg_params keywords, circular); glimmer3 defaults: -g 75, no -X
    "ignore": (dict(min_gene_len=75, allow_truncated=False), False),
    "ignore_X_g60": (dict(min_gene_len=60, allow_truncated=True), False),
    "circular": (dict(min_gene_len=75, allow_truncated=False), True),
    "circular_X_Z2": (dict(min_gene_len=75, allow_truncated=True, stop_codons=("taa", "tag")), True),
    "circular_ignore": (dict(min_gene_len=75, allow_truncated=False), True),
    "plain_g60": (dict(min_gene_len=60, allow_truncated=False), False),
}


def genome_slices(oracle):
    """the slices the goldens were made on, as the callers hand them to Find_Orfs: tolower (Filter ()) (one of them holds a 'k')"""
    g = np.load(os.path.join(GOLD, "find_orfs_general.npz"))
    genome = "".join(line.strip() for line in open(os.path.join(DATA, "NC_000915.fna")) if not line.startswith(">"))
    return g, [oracle.filter_lower(genome[int(a):int(a) + int(n)]).decode() for a, n in g["slices"]]


@pytest.mark.parametrize("name", sorted(FIND_ORFS_GENERAL))
def test_find_orfs_with_ignore_regions_and_circular_sequences(oracle, name):
    """every Orf_t the reference's Find_Orfs makes on six genome slices (30 kb .. 95 bases), with the ignore regions of -i (regions at the
    very start, overlapping, swapped, reaching past a sequence's end) and with Genome_Is_Circular: oracle/_ref/ref_orfs orfs[-circular]"""
    g, slices = genome_slices(oracle)
    kw, circular = FIND_ORFS_GENERAL[name]
    prm = oracle.mg_params(**kw)
    regions = [tuple(int(x) for x in r) for r in g[name + "_regions"]]
    want = g[name + "_orfs"]
    for k, s in enumerate(slices):
        got = oracle.find_orfs_general(s, prm, circular=circular, regions=regions)
        assert got is not None
        assert np.array_equal(got, want[want[:, 0] == k][:, 1:]), (name, k)
    assert len(want) > 400
    if not circular and not regions:                    # the plain case is the old entry point's too
        for k, s in enumerate(slices):
            assert np.array_equal(oracle.find_orfs(s, prm), want[want[:, 0] == k][:, 1:])
