"""Per-read null models in ONE call (glimmer-mg's classification mode: Update_Meta_Null_ICM,
glimmer-mg.cc:2050-2068, inside the ICM-grouped loop :361-451) and the GENE32 form of the front half's table
(option mg_gene32: fp32 gene rows, the null model applied where the running sums are built):
  * gmg_frame_score6_nulls against the real reference's tables for 8 GC values mixed in one batch
    (tests/golden/frames_multigc.npz) and against the oracle on ragged reads with 120 distinct GCs;
  * gmg_mg_score_reads with gmg_mg_params.nulls / read_null / read_ignore_score_len against the oracle read by read
    (default mode through the GENE32 kernels and through the fp64 table; the -i error branch);
  * GENE32 on or off: the same bytes out, on uniform and ragged batches."""
import os

import numpy as np
import pytest

from conftest import DATA, GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nc(gpu):
    return gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))


def ragged(rng, lengths):
    return ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]


def test_frame6_nulls_golden_eight_gc_values_in_one_batch(gpu, nc, seqs_fa):
    g = np.load(os.path.join(GOLD, "frames_multigc.npz"))
    n = len(g["frames"])
    reads = gpu.Reads.from_strings(seqs_fa[1][:n])
    ns = gpu.NullSet([gpu.Icm.indep(float(gc)) for gc in g["gcs"]])
    out = gpu.frame_score6(nc, ns, reads, read_null=g["read_null"])
    assert np.array_equal(out.reshape(6, n, 500).transpose(1, 0, 2), g["frames"])
    with pytest.raises(gpu.GmgError):
        gpu.frame_score6(nc, ns, reads, read_null=np.full(n, 8, np.uint32))     # model 8 of 8


def test_frame6_nulls_ragged_reads_120_gcs_vs_oracle(gpu, nc, oracle):
    rng = np.random.default_rng(21)
    lens = [0, 1, 2, 3, 5, 11, 12, 13, 40, 500, 499, 501, 1025, 2049] + [int(x) for x in rng.integers(1, 900, 186)]
    seqs = ragged(rng, lens)
    gcs = np.linspace(0.2, 0.8, 120)
    read_null = rng.integers(0, len(gcs), len(seqs)).astype(np.uint32)
    read_null[:3] = [119, 0, 57]
    reads = gpu.Reads.from_strings(seqs)
    ns = gpu.NullSet([gpu.Icm.indep(float(gc)) for gc in gcs])
    out = gpu.frame_score6(nc, ns, reads, read_null=read_null)
    o_nc = oracle.read(os.path.join(DATA, "NC_000915.icm"))
    o_nulls = [oracle.indep(float(gc)) for gc in gcs]
    for r, s in enumerate(seqs):
        lo, hi = int(reads.offsets[r]), int(reads.offsets[r + 1])
        assert np.array_equal(out[:, lo:hi], oracle.score_all_frames(o_nc, o_nulls[int(read_null[r])], s)), (r, len(s))
    # one model for everybody through the per-read entry = the plain entry
    same = gpu.frame_score6(nc, ns, reads, read_null=np.full(len(seqs), 57, np.uint32))
    assert np.array_equal(same, gpu.frame_score6(nc, gpu.Icm.indep(float(gcs[57])), reads))


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("gene32", [2, 0])
@pytest.mark.parametrize("uniform", [False, True, 500, 300, 151])
def test_mg_per_read_null_and_ignore_score_len_vs_oracle(gpu, nc, oracle, gene32, uniform, fused):
    """every read against its own Indep_Model and its own Ignore_Score_Len, one call: ORFs, start lists (scores bit for bit),
    best score and the accepted flag equal the oracle's, read by read.  Uniform batches: 400 / 500 bp = two reads per two-wave tile (the
    tile kernel's form with two null tables in LDS), 300 = three, 151 = six (the form with six)"""
    rng = np.random.default_rng(5 + int(uniform))
    ulen = 400 if uniform is True else int(uniform)
    lens = [ulen] * 150 if uniform else [0, 2, 14, 75, 76, 300, 512, 513, 700, 1504, 1505, 2100] + [int(x) for x in rng.integers(60, 900, 140)]
    seqs = ragged(rng, lens)
    gcs = np.linspace(0.25, 0.75, 101)
    read_null = rng.integers(0, len(gcs), len(seqs)).astype(np.uint32)
    read_isl = rng.choice([2 ** 31 - 1, 150, 300, 90], len(seqs)).astype(np.int32)
    reads = gpu.Reads.from_strings(seqs)
    ns = gpu.NullSet([gpu.Icm.indep(float(gc)) for gc in gcs])
    kw = dict(min_gene_len=60)
    with gpu.option("mg_gene32", gene32), gpu.option("mg_fused", fused):     # (fused: the tile's reads' tables in LDS; else the sequential kernels)
        orfs, starts, first = gpu.mg_score_reads(nc, ns, reads, read_null=read_null, read_ignore_score_len=read_isl, **kw)
    o_nc = oracle.read(os.path.join(DATA, "NC_000915.icm"))
    o_nulls = [oracle.indep(float(gc)) for gc in gcs]
    n_starts = n_boost = 0
    for r, s in enumerate(seqs):
        prm = oracle.mg_params(ignore_score_len=int(read_isl[r]), **kw)
        want_orfs, scored = oracle.mg_read(o_nc, o_nulls[int(read_null[r])], s.encode(), prm)
        mine = orfs[int(first[r]):int(first[r + 1])]
        assert np.array_equal(np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1).reshape(-1, 4), want_orfs), r
        for o, (out, want) in zip(mine, scored):
            st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
            assert [(s_["j"], s_["pos"], s_["which"], s_["truncated"], s_["first"], s_["score"]) for s_ in st] == \
                   [(w.j, w.pos, w.which, w.truncated, w.first, w.score) for w in want], r
            assert (o["first_j"], bool(o["accepted"]), o["best_score"]) == (out.first_j, bool(out.accepted), out.best_score)
            n_starts += len(want)
            n_boost += sum(1 for w in want if w.score == 0.0)
    assert n_starts > 500 and n_boost > 0                # the per-read Ignore_Score_Len boost did fire


@pytest.mark.parametrize("shape", ["uniform500", "ragged", "long", "tiny"])
def test_mg_gene32_and_fp64_table_give_the_same_bytes(gpu, nc, shape):
    rng = np.random.default_rng(17)
    lens = {"uniform500": [500] * 3000, "ragged": [int(x) for x in rng.integers(0, 700, 3000)],
            "long": [int(x) for x in rng.integers(400, 2600, 400)], "tiny": [int(x) for x in rng.integers(0, 40, 3000)]}[shape]
    reads = gpu.Reads.from_strings(ragged(rng, lens))
    indep = gpu.Icm.indep(0.46)
    res = []
    for g32 in (2, 0):
        with gpu.option("mg_gene32", g32):
            res.append(gpu.mg_score_reads(nc, indep, reads, min_gene_len=45, ignore_score_len=200))
    for a, b in zip(*res):
        assert a.tobytes() == b.tobytes()
    assert len(res[0][1]) > 0 or shape == "tiny"


def test_mg_error_branch_with_per_read_nulls_vs_oracle(gpu, nc, oracle):
    """-i with one null model per read: the table the walks read is gmg_frame_score6_nulls'"""
    rng = np.random.default_rng(31)
    lens = [int(x) for x in rng.integers(60, 600, 40)]
    seqs = ragged(rng, lens)
    gcs = np.linspace(0.3, 0.7, 9)
    read_null = rng.integers(0, len(gcs), len(seqs)).astype(np.uint32)
    read_isl = rng.choice([2 ** 31 - 1, 200], len(seqs)).astype(np.int32)
    reads = gpu.Reads.from_strings(seqs)
    ns = gpu.NullSet([gpu.Icm.indep(float(gc)) for gc in gcs])
    orfs, starts, first, errs = gpu.mg_score_reads(nc, ns, reads, read_null=read_null, read_ignore_score_len=read_isl, allow_indels=True)
    o_nc = oracle.read(os.path.join(DATA, "NC_000915.icm"))
    o_nulls = [oracle.indep(float(gc)) for gc in gcs]
    ep = oracle.mg_err_params(allow_indels=True)
    total = 0
    for r, s in enumerate(seqs):
        prm = oracle.mg_params(ignore_score_len=int(read_isl[r]))
        _, _, scored = oracle.mg_read_errors(o_nc, o_nulls[int(read_null[r])], s.encode(), prm, ep)
        mine = orfs[int(first[r]):int(first[r + 1])]
        assert len(mine) == len(scored)
        for o, (out, want) in zip(mine, scored):
            sl = slice(o["start_begin"], o["start_begin"] + o["n_starts"])
            got = [(int(s_["j"]), int(s_["pos"]), float(s_["score"]), int(e["n"])) for s_, e in zip(starts[sl], errs[sl])]
            assert got == [(w.s.j, w.s.pos, w.s.score, w.n_errors) for w in want], r
            assert int(o["accepted"]) == out.accepted
            total += len(want)
    assert total > 1000
