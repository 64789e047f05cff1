"""Pins the oracle's restatement of glimmer-mg's front half -- Find_Orfs (src/Glimmer/glimmer_base.cc:638-779),
Save_Prev_Stops (src/Glimmer/glimmer-mg.cc:675-729), Score_Orf_Starts / Score_Orfs_Errors (:1605-1861) -- to the
real reference: oracle/_ref/ref_mg_orfs pulls the reference's glimmer-mg.cc in whole, runs its own Find_Orfs and
Score_Orfs_Errors on seqs.fa and dumps every ORF and every start list handed to Add_Events_*
(tests/golden/mg_orfs_*.npz).  Exact equality, including the double scores.  CPU only."""
import math
import os

import numpy as np
import pytest

from conftest import DATA, GOLD

CASES = {
    "mg_orfs_default": dict(),
    "mg_orfs_g120": dict(min_gene_len=120),
    "mg_orfs_Z2": dict(stop_codons=("taa", "tag")),
}


def ignore_score_len(gc, stops):
    """Set_Ignore_Score_Len (glimmer_base.cc:2597-2631)"""
    lam = 0.0
    for s in stops:
        x = 1.0
        for ch in s:
            x *= gc / 2.0 if ch in "cg" else (1.0 - gc) / 2.0
        lam += x
    return int(math.floor(3.0 * math.log(2.0 * 1000000 * lam) / lam))


def sorted_starts(starts):
    """the reference sorts by pos with an unstable sort; ties (truncated + real start of one codon) by which"""
    rows = [(s.pos, s.which, s.j, s.truncated, s.first, s.score) for s in starts]
    return sorted(rows, key=lambda r: (r[0], r[1]))


def golden_rows(g, b, cnt):
    si, ss = g["start_int"][b:b + cnt], g["start_score"][b:b + cnt]
    rows = [(int(r[1]), int(r[2]), int(r[0]), int(r[3]), int(r[4]), float(s)) for r, s in zip(si, ss)]
    return sorted(rows, key=lambda r: (r[0], r[1]))


def mg_case(oracle, name):
    kw = dict(CASES[name])
    gc = float(np.load(os.path.join(GOLD, "frames_nc.npz"))["gc"])
    stops = kw.get("stop_codons", ("taa", "tag", "tga"))
    kw["ignore_score_len"] = ignore_score_len(gc, stops)
    return oracle.mg_params(**kw), oracle.indep(gc, stops), kw


@pytest.mark.parametrize("name", sorted(CASES))
def test_mg_front_half_matches_reference(oracle, seqs_fa, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    nc = oracle.read(os.path.join(DATA, "NC_000915.icm"))
    prm, indep, _ = mg_case(oracle, name)
    reads = [oracle.filter_lower(s) for s in seqs_fa[1]]
    gold_orfs = g["orfs"]
    accepted = {int(o): i for i, o in enumerate(g["gene_orf"])}
    base = 0
    n_checked = 0
    for r, seq in enumerate(reads):
        orfs, scored = oracle.mg_read(nc, indep, seq, prm)
        want = gold_orfs[gold_orfs[:, 0] == r][:, 1:]
        assert np.array_equal(orfs, want), "Find_Orfs differs on read %d" % r
        for k, (out, starts) in enumerate(scored):
            oi = base + k
            if oi in accepted:
                gi = accepted[oi]
                assert out.accepted
                assert sorted_starts(starts) == golden_rows(g, int(g["gene_start_begin"][gi]), int(g["gene_nstarts"][gi]))
                n_checked += 1
            else:
                assert not out.accepted
        base += len(orfs)
    assert base == len(gold_orfs)
    assert n_checked == len(accepted) > 200


@pytest.mark.parametrize("name,trunc", [("orfs_default", False), ("orfs_X", True)])
def test_find_orfs_matches_glimmer3_goldens(oracle, seqs_fa, name, trunc):
    """the same Find_Orfs serves glimmer3: its goldens cover Allow_Truncated_Orfs = false as well"""
    g = np.load(os.path.join(GOLD, name + ".npz"))["orfs"]          # read, frame, stop_position, orf_len
    prm = oracle.mg_params(min_gene_len=75, allow_truncated=trunc)
    reads = [oracle.filter_lower(s) for s in seqs_fa[1]]
    for r, seq in enumerate(reads):
        orfs = oracle.find_orfs(seq, prm)
        assert np.array_equal(orfs[:, [0, 1, 3]], g[g[:, 0] == r][:, 1:]), "read %d" % r


def test_prev_stop_tables_match_a_plain_scan(oracle, seqs_fa):
    """Save_Prev_Stops against its definition: last forward stop at or before i in i's class / next reverse stop"""
    prm = oracle.mg_params()
    seq = oracle.filter_lower(seqs_fa[1][3])
    n = len(seq)
    fwd, rev = oracle.save_prev_stops(seq, prm)
    s = seq.decode()
    stops, rstops = ("taa", "tag", "tga"), ("tta", "cta", "tca")
    for i in range(n):
        want = [0, 1, -1][i % 3]
        for e in range(i, 1, -3):
            if s[e - 2:e + 1] in stops:
                want = e
                break
        assert fwd[i] == want
        want = [n - 1, n - 2, n][(n - 1 - i) % 3]
        for b in range(i, n - 2, 3):
            if s[b:b + 3] in rstops:
                want = b
                break
        assert rev[i] == want


# ---- the error branch (-i indels / -s substitutions): Score_Indels + the recursive Score_Orf_Starts ----
ERR_CASES = {
    "mg_err_indel": (dict(), dict(allow_indels=True), False),
    "mg_err_sub": (dict(), dict(allow_subs=True), False),
    "mg_err_indel_q": (dict(), dict(allow_indels=True), True),
    "mg_err_indel_g90": (dict(min_gene_len=90, stop_codons=("taa", "tag")), dict(allow_indels=True), False),
}


def read_fasta_file(path):
    recs = open(path).read().split(">")[1:]
    return [r.split("\n", 1)[1].replace("\n", "") for r in recs]


def read_qual_file(path):
    recs = open(path).read().split(">")[1:]
    return [np.array(r.split("\n", 1)[1].split(), np.int32) for r in recs]


def err_case(oracle, name):
    kw, ekw, with_q = ERR_CASES[name]
    kw = dict(kw)
    reads = [oracle.filter_lower(s) for s in read_fasta_file(os.path.join(DATA, "seqs80.fa"))]
    ct = sum(s.count(b"g") + s.count(b"c") for s in reads)
    gc = float(ct) / sum(len(s) for s in reads)                     # Set_GC_Fraction on the file the reference was given
    stops = kw.get("stop_codons", ("taa", "tag", "tga"))
    kw["ignore_score_len"] = ignore_score_len(gc, stops)
    quals = read_qual_file(os.path.join(DATA, "seqs80.qual")) if with_q else [None] * len(reads)
    return reads, quals, oracle.mg_params(**kw), oracle.mg_err_params(**ekw), oracle.indep(gc, stops), gc, kw, ekw


def err_rows(starts):
    return [(s.s.j, s.s.pos, s.s.which, s.s.truncated, s.s.first, s.n_errors, s.err_pos[0], s.err_type[0],
             s.err_pos[1], s.err_type[1], s.s.score) for s in starts]


def err_golden_rows(g, b, cnt):
    return [tuple(int(x) for x in r) + (float(s),) for r, s in zip(g["start_int"][b:b + cnt], g["start_score"][b:b + cnt])]


@pytest.mark.parametrize("name", sorted(ERR_CASES))
def test_mg_error_branch_matches_reference_push_order(oracle, name):
    """every accepted ORF's start list, IN THE ORDER Score_Orf_Starts pushed it (the reference's list right before its
    sort, oracle/ref_drivers/ref_mg_orfs.cc), with the Error_t entries and the double scores: exact equality"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    nc = oracle.read(os.path.join(DATA, "NC_000915.icm"))
    reads, quals, prm, ep, indep, _, _, _ = err_case(oracle, name)
    gold_orfs = g["orfs"]
    accepted = {int(o): i for i, o in enumerate(g["gene_orf"])}
    base = n_checked = n_amb = 0
    for r, seq in enumerate(reads):
        orfs, _, scored = oracle.mg_read_errors(nc, indep, seq, prm, ep, quals[r])
        want = gold_orfs[gold_orfs[:, 0] == r][:, 1:]
        assert np.array_equal(orfs, want), "Find_Orfs differs on read %d" % r
        for k, (out, starts) in enumerate(scored):
            oi = base + k
            if oi in accepted:
                gi = accepted[oi]
                assert out.accepted in (1, 2)
                assert err_rows(starts) == err_golden_rows(g, int(g["gene_start_begin"][gi]), int(g["gene_nstarts"][gi])), (r, k)
                n_checked += 1
            else:
                assert out.accepted in (0, 2)       # 2: ties on pos leave first_j to the reference's unstable sort
            n_amb += out.accepted == 2
        base += len(orfs)
    assert base == len(gold_orfs)
    assert n_checked == len(accepted) > 20
    assert n_amb * 20 <= base


def test_quality_454_definitions(oracle):
    """Set_Quality_454: 31 inside a homopolymer run, 31 - 5 r (r < 6, else 6) on its last base"""
    q = oracle.quality_454(b"acccgttttttttaag")
    assert list(q) == [26, 31, 31, 16, 26, 31, 31, 31, 31, 31, 31, 31, 6, 31, 21, 26]
    q = oracle.quality_454(b"aacg", user=[0, 5, 40, -3])
    assert list(q) == [19, 5, 40, 1]
