"""glimmer-mg's error branch on the device (gmg_mg_score_reads with GMG_MG_ALLOW_INDELS / GMG_MG_ALLOW_SUBS;
Score_Indels and the recursive Score_Orf_Starts, src/Glimmer/glimmer-mg.cc:1513-1861) against
  * the reference's own start lists in PUSH order with their Error_t entries (tests/golden/mg_err_*.npz: the list as it
    was right before Score_Orfs_Errors' sort, oracle/ref_drivers/ref_mg_orfs.cc) -- -i, -s, -i with a quality file,
    -i with other gene length / stop codons;
  * the oracle on seeded random reads of ragged lengths, EVERY ORF (also the rejected ones).
Integer fields, error lists, order and double scores must be equal bit for bit."""
import os

import numpy as np
import pytest

from conftest import DATA, GOLD
from test_oracle_mg import ERR_CASES, err_case, err_golden_rows, err_rows, ignore_score_len

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nc(gpu):
    return gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))


def dev_err_rows(starts, errs):
    return [(int(s["j"]), int(s["pos"]), int(s["which"]), int(s["truncated"]), int(s["first"]), int(e["n"]),
             int(e["pos"][0]), int(e["type"][0]), int(e["pos"][1]), int(e["type"][1]), float(s["score"]))
            for s, e in zip(starts, errs)]


def quality_array(quals):
    return None if quals[0] is None else np.concatenate(quals).astype(np.uint8)


@pytest.mark.parametrize("name", sorted(ERR_CASES))
def test_error_branch_matches_reference_push_order(gpu, oracle, nc, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    seqs, quals, _, _, _, gc, kw, ekw = err_case(oracle, name)
    stops = kw.get("stop_codons", ("taa", "tag", "tga"))
    reads = gpu.Reads.from_strings([s.decode() for s in seqs])
    orfs, starts, off, errs = gpu.mg_score_reads(nc, gpu.Icm.indep(gc, stops), reads, quality=quality_array(quals), **kw, **ekw)
    got = np.stack([orfs["read"].astype(np.int32), orfs["frame"], orfs["stop_position"], orfs["gene_len"], orfs["orf_len"]], 1)
    assert np.array_equal(got, g["orfs"])                               # Find_Orfs with the Min_Indel_ORF_Len rule
    accepted = np.zeros(len(orfs), bool)
    accepted[g["gene_orf"]] = True
    sure = orfs["accepted"] != 2                                        # 2: left to the caller's sort (ties on pos)
    assert np.array_equal((orfs["accepted"] == 1)[sure], accepted[sure])
    assert (~sure).sum() * 20 <= len(orfs)
    for oi, b, cnt in zip(g["gene_orf"], g["gene_start_begin"], g["gene_nstarts"]):
        o = orfs[oi]
        assert o["n_starts"] == cnt
        sl = slice(o["start_begin"], o["start_begin"] + o["n_starts"])
        assert dev_err_rows(starts[sl], errs[sl]) == err_golden_rows(g, int(b), int(cnt)), oi


@pytest.mark.parametrize("ekw,with_q", [
    (dict(allow_indels=True), False),
    (dict(allow_indels=True), True),
    (dict(allow_indels=True, indel_max=1, indel_quality_threshold=21, indel_suffix_score_threshold=-6.0), False),
    # every base may branch (threshold above all qualities; the penalty table beyond its LDS copy); a tight suffix threshold bounds the tree
    (dict(allow_indels=True, indel_quality_threshold=70, indel_suffix_score_threshold=-2.5), True),
    (dict(allow_subs=True), False),
])
@pytest.mark.parametrize("kw", [dict(), dict(allow_truncated=False, min_gene_len=60), dict(ignore_score_len=150, start_codons=("atg", "rtg"))])
@pytest.mark.parametrize("path", ["wave", "wave-walk", "wave-mixed", "wave-overflow", "wave-table", "tile", "tile-stage", "tile-overflow", "level", "level-q0", "flat", "level-overflow", "level-grow"])
def test_error_branch_every_orf_vs_oracle(gpu, oracle, nc, kw, ekw, with_q, path, monkeypatch, request_finalizers):
    """path: one wave per (read, strand) with running sums and masks in its LDS (k_mg_err_wcount: both passes without walks, breadth
    first; the default; wave-walk: k_mg_err_wave, a stack of calls per wave, for both passes; wave-mixed: the stack walker as the
    write pass only; the 1300- and 2100-bp
    reads, longer than a wave takes, go to the per-ORF kernel), the same with the stack walker as the count pass, with a stack of 5
    entries (the call repeats on the level kernels) and on a batch without the long reads (the fp32 gene rows instead of the fp64 table),
    tile by tile with the running sums in LDS, one lane per event (option mg_err_tile; the 2100-bp read, longer than a tile,
    goes to the per-ORF kernel), the same with staging arrays too small (the kernel repeats with what it asked for) and
    with slabs too small (everything repeats on the level kernels), level by level with
    one lane per call on the tables in HBM (the default; the 2100-bp read goes to the per-ORF kernel; -s: the one-value-per-base
    table -- q0: the three-row table instead), the per-ORF kernel alone, the level
    kernels with call arrays too small (everything repeats on the per-ORF kernel), and the same with the arrays allowed to
    grow (the count pass repeats with larger ones)"""
    opts = {"wave": {"mg_err_wave": 1, "mg_err_tile": 0}, "wave-walk": {"mg_err_wave": 2, "mg_err_tile": 0}, "wave-mixed": {"mg_err_wave": 3, "mg_err_tile": 0},
            "wave-overflow": {"mg_err_wave": 1, "mg_err_tile": 0, "mg_err_wave_q": 5}, "wave-table": {"mg_err_wave": 1, "mg_err_tile": 0},
            "tile": {"mg_err_tile": 1}, "tile-stage": {"mg_err_tile": 1, "mg_err_tile_q": -1}, "tile-overflow": {"mg_err_tile": 1, "mg_err_tile_q": 3},
            "level": {"mg_err_tile": 0, "mg_err_wave": 0}, "level-q0": {"mg_err_tile": 0, "mg_err_wave": 0, "mg_err_qonly": 0},
            "flat": {"mg_err_flat": 1}, "level-overflow": {"mg_err_tile": 0, "mg_err_wave": 0, "mg_err_calls": 7},
            "level-grow": {"mg_err_tile": 0, "mg_err_wave": 0, "mg_err_calls": 7, "mg_err_calls_grow": 1}}[path]
    for k, v in opts.items():
        old = gpu.get_option(k)
        gpu.set_option(k, v)
        request_finalizers.append(lambda k=k, old=old: gpu.set_option(k, old))
    rng = np.random.default_rng(99)
    lengths = [0, 1, 5, 14, 15, 16, 17, 18, 33, 74, 75, 76, 99, 150, 231, 300, 301, 302, 400, 523, 700]
    seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]
    seqs.append("acg" * 120)                                            # no stop codon at all
    seqs.append("a" * 40 + "".join("acgt"[c] for c in rng.integers(0, 4, size=200)) + "tttttttt" + "gggg" * 9)   # long runs
    seqs.append("".join("acgt"[c] for c in rng.integers(0, 4, size=1300)))
    seqs.append("".join("acgt"[c] for c in rng.integers(0, 4, size=2100)))      # too long for the order keys
    if path == "wave-table":                            # every read fits a wave: the call's own table is the fp32 gene rows
        seqs = seqs[:-2] + ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in (959, 960)]
    quals = [np.where(rng.random(len(s)) < 0.12, rng.integers(0, 19, len(s)), rng.integers(19, 41, len(s))).astype(np.int32)
             for s in seqs] if with_q else [None] * len(seqs)
    gc, stops = 0.5, kw.get("stop_codons", ("taa", "tag", "tga"))
    reads = gpu.Reads.from_strings(seqs)
    okw = {k: v for k, v in ekw.items() if k != "min_indel_orf_len"}
    prm, ep = oracle.mg_params(**kw), oracle.mg_err_params(**okw)
    o_nc, o_indep = oracle.read(os.path.join(DATA, "NC_000915.icm")), oracle.indep(gc, stops)
    orfs, starts, off, errs = gpu.mg_score_reads(nc, gpu.Icm.indep(gc, stops), reads,
                                                 quality=np.concatenate(quals).astype(np.uint8) if with_q else None, **kw, **ekw)
    n_starts = n_children = 0
    for r, s in enumerate(seqs):
        want_orfs, _, scored = oracle.mg_read_errors(o_nc, o_indep, s.encode(), prm, ep, quals[r])
        mine = orfs[int(off[r]):int(off[r + 1])]
        assert np.array_equal(np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1).reshape(-1, 4), want_orfs)
        for o, (out, want) in zip(mine, scored):
            sl = slice(o["start_begin"], o["start_begin"] + o["n_starts"])
            assert dev_err_rows(starts[sl], errs[sl]) == err_rows(want), (r, o)
            assert (int(o["lo"]), int(o["hi"]), int(o["accepted"])) == (out.lo, out.hi, out.accepted)
            if out.accepted:
                assert float(o["best_score"]) == out.best_score and int(o["first_j"]) == out.first_j
            n_starts += len(want)
            n_children += sum(1 for w in want if w.n_errors)
    assert n_starts > 50 and n_children > 10


def test_error_branch_accepted_only_and_flag_errors(gpu, oracle, nc):
    seqs, _, _, _, _, gc, kw, _ = err_case(oracle, "mg_err_indel")
    reads = gpu.Reads.from_strings([s.decode() for s in seqs])
    full = gpu.mg_score_reads(nc, gpu.Icm.indep(gc), reads, allow_indels=True, **kw)
    kept = gpu.mg_score_reads(nc, gpu.Icm.indep(gc), reads, allow_indels=True, accepted_only=True, **kw)
    keep = full[0]["accepted"] != 0
    assert len(kept[0]) == keep.sum() > 20
    for a, b in zip(full[0][keep], kept[0]):
        sa = slice(a["start_begin"], a["start_begin"] + a["n_starts"])
        sb = slice(b["start_begin"], b["start_begin"] + b["n_starts"])
        assert (a["read"], a["frame"], a["stop_position"], a["n_starts"]) == (b["read"], b["frame"], b["stop_position"], b["n_starts"])
        assert np.array_equal(full[1][sa], kept[1][sb]) and np.array_equal(full[3][sa], kept[3][sb])
    assert np.array_equal(kept[2], np.concatenate([[0], np.cumsum(keep)])[full[2].astype(np.int64)])
    with pytest.raises(gpu.GmgError):                                   # glimmer-mg.cc:952: not both
        gpu.mg_score_reads(nc, gpu.Icm.indep(gc), reads, allow_indels=True, allow_subs=True, **kw)
    with pytest.raises(gpu.GmgError):
        gpu.mg_score_reads(nc, gpu.Icm.indep(gc), reads, allow_indels=True, indel_max=3, **kw)
    # the plain mode is untouched by the new fields, and find_orfs follows the flag
    plain = gpu.mg_score_reads(nc, gpu.Icm.indep(gc), reads, **kw)
    assert len(plain[0]) < len(full[0])


def test_error_branch_full_size_properties(gpu, oracle, nc):
    """BASELINE configs[4] shape: 1M reads of ~400 bp (N(400, 60^2) clipped to 100..700), Set_Quality_454 qualities, -i.
    (1) determinism: two calls give identical bytes although slots inside an ORF's slice are handed out by atomics (the
    sort by key restores the push order); (2) bookkeeping: accepted ORFs only, slices back to back and exactly
    sum(n_starts) long, error lists consistent with their length; (3) the per-ORF fallback kernel gives the same bytes on
    a 20k-read slice; (4) sampled reads equal the oracle, every accepted ORF with its list in push order."""
    n = 1_000_000
    lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    packed, _ = gpu.synth.packed_reads(1, int(off[-1]), 7)
    reads = gpu.Reads(packed, off)
    indep = gpu.Icm.indep(0.5)
    orfs, starts, first, errs = gpu.mg_score_reads(nc, indep, reads, allow_indels=True, accepted_only=True)
    orfs2, starts2, first2, errs2 = gpu.mg_score_reads(nc, indep, reads, allow_indels=True, accepted_only=True)
    assert orfs.tobytes() == orfs2.tobytes() and starts.tobytes() == starts2.tobytes() and errs.tobytes() == errs2.tobytes()
    assert np.array_equal(first, first2)
    del orfs2, starts2, errs2
    with gpu.option("mg_err_tile", 1):                  # tile by tile, one lane per event: the same bytes as the wave kernel's at full size
        orfs2, starts2, first2, errs2 = gpu.mg_score_reads(nc, indep, reads, allow_indels=True, accepted_only=True)
    assert orfs.tobytes() == orfs2.tobytes() and starts.tobytes() == starts2.tobytes() and errs.tobytes() == errs2.tobytes()
    assert np.array_equal(first, first2)
    del orfs2, starts2, errs2
    with gpu.option("mg_err_tile", 0), gpu.option("mg_err_wave", 0):    # ... and the level kernels on the running sums in HBM
        orfs2, starts2, first2, errs2 = gpu.mg_score_reads(nc, indep, reads, allow_indels=True, accepted_only=True)
    assert orfs.tobytes() == orfs2.tobytes() and starts.tobytes() == starts2.tobytes() and errs.tobytes() == errs2.tobytes()
    assert np.array_equal(first, first2)
    del orfs2, starts2, errs2
    assert len(orfs) == first[-1] > n // 10 and np.all(orfs["accepted"] != 0)
    assert np.array_equal(orfs["start_begin"], np.concatenate([[0], np.cumsum(orfs["n_starts"].astype(np.int64))[:-1]]))
    assert int(orfs["n_starts"].astype(np.int64).sum()) == len(starts) == len(errs)
    assert errs["n"].min() >= 0 and errs["n"].max() == 2
    assert np.all(errs["type"][errs["n"] == 0] == 0) and np.all(errs["pos"][errs["n"] < 2][:, 1] == 0)
    assert np.all(np.diff(orfs["read"].astype(np.int64)) >= 0)

    m = 20_000                                          # a slice on its own, default kernels and the exact per-ORF fallback
    sub_packed, _ = gpu.synth.packed_reads(1, int(off[m]), 7)       # same generator, same seed: the first off[m] bases
    sub = gpu.Reads(sub_packed, off[:m + 1].copy())
    a = gpu.mg_score_reads(nc, indep, sub, allow_indels=True, accepted_only=True)
    with gpu.option("mg_err_flat", 1):
        b = gpu.mg_score_reads(nc, indep, sub, allow_indels=True, accepted_only=True)
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    k = int(first[m])
    assert a[0].tobytes() == orfs[:k].tobytes() and a[1].tobytes() == starts[:len(a[1])].tobytes()

    prm, ep = oracle.mg_params(), oracle.mg_err_params(allow_indels=True)
    o_nc, o_indep = oracle.read(os.path.join(DATA, "NC_000915.icm")), oracle.indep(0.5)
    checked = 0
    for r in (0, 1, 4242, 31337, 555555, n - 1):
        seq = gpu.synth.unpack_ascii(packed, int(off[r]), int(off[r + 1] - off[r]))
        want_orfs, _, scored = oracle.mg_read_errors(o_nc, o_indep, seq, prm, ep)
        acc = [(o, out, st) for o, (out, st) in zip(want_orfs, scored) if out.accepted]
        mine = orfs[int(first[r]):int(first[r + 1])]
        assert len(mine) == len(acc)
        for g, (o, out, st) in zip(mine, acc):
            assert (int(g["frame"]), int(g["stop_position"]), int(g["accepted"])) == (int(o[0]), int(o[1]), out.accepted)
            sl = slice(g["start_begin"], g["start_begin"] + g["n_starts"])
            assert dev_err_rows(starts[sl], errs[sl]) == err_rows(st)
            checked += 1
    assert checked >= 1


@pytest.mark.parametrize("name,kw", [
    ("indel", dict(allow_indels=True)),
    ("indel_short_genes", dict(allow_indels=True, min_gene_len=45)),
    ("indel_long_genes", dict(allow_indels=True, min_gene_len=150)),
    ("indel_one_level", dict(allow_indels=True, indel_max=1)),
    ("indel_no_truncated", dict(allow_indels=True, allow_truncated=False)),
    ("sub", dict(allow_subs=True)),
    ("sub_long_genes", dict(allow_subs=True, min_gene_len=120)),
])
def test_accepted_only_is_the_full_result_filtered(gpu, nc, name, kw):
    """GMG_MG_ACCEPTED_ONLY does not expand ORFs that cannot reach Min_Gene_Len before their read ends (no path gets further from the
    ORF's end than the read does): what it returns must be exactly the accepted ORFs of the full result -- 30,000 ragged reads
    (0 .. 1,500 bp, many ORFs near the read ends), every record, start and error list"""
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    lens = np.clip(rng.normal(300, 200, 30_000).round(), 0, 1500).astype(np.int64)
    reads = gpu.Reads.from_strings(["".join("acgt"[c] for c in rng.integers(0, 4, size=int(n))) for n in lens])
    indep = gpu.Icm.indep(0.5)
    full = gpu.mg_score_reads(nc, indep, reads, **kw)
    kept = gpu.mg_score_reads(nc, indep, reads, accepted_only=True, **kw)
    with gpu.option("mg_err_tile", 1):                  # (the tile kernel: every record, start and error list of both forms)
        for want, acc in ((full, False), (kept, True)):
            got = gpu.mg_score_reads(nc, indep, reads, accepted_only=acc, **kw)
            for x, y in zip(want, got):
                assert x.tobytes() == y.tobytes()
    with gpu.option("mg_err_tile", 0), gpu.option("mg_err_wave", 2):    # (... the stack walker for both passes)
        for want, acc in ((full, False), (kept, True)):
            got = gpu.mg_score_reads(nc, indep, reads, accepted_only=acc, **kw)
            for x, y in zip(want, got):
                assert x.tobytes() == y.tobytes()
    with gpu.option("mg_err_tile", 0), gpu.option("mg_err_wave", 0):    # (... and the level kernels)
        for want, acc in ((full, False), (kept, True)):
            got = gpu.mg_score_reads(nc, indep, reads, accepted_only=acc, **kw)
            for x, y in zip(want, got):
                assert x.tobytes() == y.tobytes()
    keep = full[0]["accepted"] != 0
    assert len(kept[0]) == keep.sum() > 20
    fo = full[0][keep]
    for f in ("read", "frame", "stop_position", "n_starts", "accepted", "lo", "hi", "first_j"):
        assert np.array_equal(fo[f], kept[0][f]), f
    assert fo["best_score"].tobytes() == kept[0]["best_score"].tobytes()
    idx = np.concatenate([np.arange(b, b + c) for b, c in zip(fo["start_begin"].astype(np.int64), fo["n_starts"].astype(np.int64))])
    assert full[1][idx].tobytes() == kept[1].tobytes() and full[3][idx].tobytes() == kept[3].tobytes()
    assert np.array_equal(kept[2], np.concatenate([[0], np.cumsum(keep)])[full[2].astype(np.int64)])


def test_long_start_lists_keep_the_reference_order(gpu, oracle, nc):
    """k_mg_order_starts puts an ORF's starts into push order by counting: up to 64 starts with the keys in the lanes, beyond that through
    LDS 512 at a time, eight starts per lane and round.  Reads rich in G + C (ORFs that span them) with a quality file that lets every
    third base branch give lists of several thousand starts: every tier, against the oracle entry by entry."""
    rng = np.random.default_rng(3)

    def rr(n, at):
        return "".join("acgt"[c] for c in rng.choice(4, size=n, p=[at, .5 - at, .5 - at, at]))
    seqs = [rr(900, 0.12) for _ in range(5)] + [rr(600, 0.2) for _ in range(5)] + [rr(300, 0.25) for _ in range(5)]
    reads = gpu.Reads.from_strings(seqs)
    quals = [rng.integers(0, 41, size=len(s)).astype(np.int32) for s in seqs]
    ekw = dict(allow_indels=True, indel_quality_threshold=30, indel_suffix_score_threshold=-12.0)
    orfs, starts, off, errs = gpu.mg_score_reads(nc, gpu.Icm.indep(0.5), reads, quality=np.concatenate(quals).astype(np.uint8), **ekw)
    ns = orfs["n_starts"]
    assert int((ns <= 64).sum()) > 0 and int(((ns > 64) & (ns <= 512)).sum()) > 0 and int((ns > 1024).sum()) > 0, int(ns.max())
    prm, ep = oracle.mg_params(), oracle.mg_err_params(**ekw)
    o_nc, o_indep = oracle.read(os.path.join(DATA, "NC_000915.icm")), oracle.indep(0.5, ("taa", "tag", "tga"))
    checked = 0
    for r, s in enumerate(seqs):
        want_orfs, _, scored = oracle.mg_read_errors(o_nc, o_indep, s.encode(), prm, ep, quals[r])
        mine = orfs[int(off[r]):int(off[r + 1])]
        assert len(mine) == len(scored)
        for o, (out, want) in zip(mine, scored):
            sl = slice(o["start_begin"], o["start_begin"] + o["n_starts"])
            assert dev_err_rows(starts[sl], errs[sl]) == err_rows(want), (r, int(o["n_starts"]))
            checked += len(want)
    assert checked == int(ns.sum())


def test_level_lists_drained_on_demand(gpu, oracle, nc):
    """k_mg_err_wcount holds 128 calls per level and works a list off whenever the children of the next 64 pairs would not fit (then it
    evaluates those 64 pairs again).  Reads of every length class with 100 - 155 low-quality bases each (up to the 160 a wave takes) and
    a suffix threshold that lets most candidates through: hundreds of level-1 and thousands of level-2 calls per strand, i.e. many drains
    in both passes.  Every ORF against the oracle on the short reads, and the level kernels on all of them, byte for byte."""
    rng = np.random.default_rng(41)
    lengths = [200, 260, 330, 384, 385, 420, 448, 449, 500, 512, 513, 600, 704, 705, 800, 960]
    seqs = ["".join("acgt"[c] for c in rng.choice(4, size=n, p=[.2, .3, .3, .2])) for n in lengths for _ in range(3)]
    quals = []
    for s in seqs:
        q = np.full(len(s), 35, np.int32)
        q[rng.choice(len(s), size=int(rng.integers(100, 156)), replace=False)] = rng.integers(0, 19, size=1)[0]
        quals.append(q)
    reads = gpu.Reads.from_strings(seqs)
    ekw = dict(allow_indels=True, indel_suffix_score_threshold=-40.0)
    qual = np.concatenate(quals).astype(np.uint8)
    with gpu.option("mg_err_wave", 1), gpu.option("mg_err_tile", 0):
        wave = gpu.mg_score_reads(nc, gpu.Icm.indep(0.5), reads, quality=qual, **ekw)
    with gpu.option("mg_err_wave", 0), gpu.option("mg_err_tile", 0):
        level = gpu.mg_score_reads(nc, gpu.Icm.indep(0.5), reads, quality=qual, **ekw)
    for x, y in zip(wave, level):
        assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes()
    orfs, starts, off, errs = wave
    assert int(orfs["n_starts"].max()) > 2000 and int((errs["n"] == 2).sum()) > 100000
    prm, ep = oracle.mg_params(), oracle.mg_err_params(**ekw)
    o_nc, o_indep = oracle.read(os.path.join(DATA, "NC_000915.icm")), oracle.indep(0.5, ("taa", "tag", "tga"))
    checked = 0
    for r in (0, 4, 9, 13):                              # 200, 260, 384 and 420 bases
        want_orfs, _, scored = oracle.mg_read_errors(o_nc, o_indep, seqs[r].encode(), prm, ep, quals[r])
        mine = orfs[int(off[r]):int(off[r + 1])]
        assert len(mine) == len(scored)
        for o, (out, want) in zip(mine, scored):
            sl = slice(o["start_begin"], o["start_begin"] + o["n_starts"])
            assert dev_err_rows(starts[sl], errs[sl]) == err_rows(want), (r, int(o["n_starts"]))
            checked += len(want)
    assert checked > 5000
