"""glimmer-mg's front half on the device (include/gmg.h: gmg_mg_score_reads) against
  * the reference's own Find_Orfs + Score_Orfs_Errors on seqs.fa (tests/golden/mg_orfs_*.npz, written by
    oracle/_ref/ref_mg_orfs, which pulls the reference's glimmer-mg.cc in whole),
  * the oracle on seeded random reads of ragged lengths (edge cases: reads shorter than Min_Gene_Len, no stop at
    all, truncated ORFs off, the Ignore_Score_Len boost, IUPAC start codons, two stop codons),
  * the reference CLI: integration/glimmer-mg_gpu = glimmer-mg's own events / DP / trace-back around
    gmg_mg_score_reads calls must write byte-identical .predict files -- as one process, in several batches, and as one
    process per shard (forked, the job's GC fraction summed from the shards' counts).
Integer fields and double scores must be equal bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, GOLD, ROOT, built_binary
from test_oracle_mg import CASES, golden_rows, ignore_score_len, mg_case, sorted_starts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nc(gpu):
    return gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))


def dev_rows(starts):
    rows = [(int(s["pos"]), int(s["which"]), int(s["j"]), int(s["truncated"]), int(s["first"]), float(s["score"]))
            for s in starts]
    return sorted(rows, key=lambda r: (r[0], r[1]))


@pytest.mark.parametrize("name", sorted(CASES))
def test_mg_front_half_matches_reference_goldens(gpu, oracle, nc, seqs_fa, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    _, _, kw = mg_case(oracle, name)
    gc = float(np.load(os.path.join(GOLD, "frames_nc.npz"))["gc"])
    stops = kw.get("stop_codons", ("taa", "tag", "tga"))
    reads = gpu.Reads.from_strings(seqs_fa[1])
    orfs, starts, off = gpu.mg_score_reads(nc, gpu.Icm.indep(gc, stops), reads, **kw)
    got = np.stack([orfs["read"].astype(np.int32), orfs["frame"], orfs["stop_position"], orfs["gene_len"],
                    orfs["orf_len"]], 1)
    assert np.array_equal(got, g["orfs"])                               # Find_Orfs, every read, in order
    assert np.array_equal(off, np.searchsorted(g["orfs"][:, 0], np.arange(len(seqs_fa[1]) + 1)))
    accepted = np.zeros(len(orfs), bool)
    accepted[g["gene_orf"]] = True
    assert np.array_equal(orfs["accepted"] != 0, accepted)              # what goes to Add_Events_*
    for oi, b, cnt in zip(g["gene_orf"], g["gene_start_begin"], g["gene_nstarts"]):
        o = orfs[oi]
        assert o["n_starts"] == cnt
        st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
        assert dev_rows(st) == golden_rows(g, int(b), int(cnt))


def random_reads(rng, lengths):
    return ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]


@pytest.mark.parametrize("kw", [
    dict(),
    dict(allow_truncated=False),
    dict(min_gene_len=90, ignore_score_len=150),
    dict(min_gene_len=30, start_codons=("atg", "rtg", "ttg", "ctg"), stop_codons=("taa", "tag"), start_threshold=-2.0),
    dict(min_gene_len=4, ignore_score_len=10, start_threshold=-1e300),
])
def test_mg_front_half_matches_oracle_on_random_ragged_reads(gpu, oracle, nc, kw):
    rng = np.random.default_rng(20260102)
    lengths = list(rng.integers(1, 700, size=300)) + [1, 2, 3, 4, 5, 29, 30, 74, 75, 76, 89, 90, 91, 500, 1500]
    seqs = random_reads(rng, lengths)
    seqs.append("acg" * 200)                                            # no stop codon in any forward frame
    seqs.append("taa" * 100 + "a")                                      # stops back to back
    seqs.append("atg" + "gct" * 150 + "taa" + "cc")                     # one clean forward gene
    seqs.append("gg" + "tta" + "agc" * 150 + "cat" + "g")               # its reverse-strand twin
    stops = kw.get("stop_codons", ("taa", "tag", "tga"))
    o_nc = oracle.read(os.path.join(DATA, "NC_000915.icm"))
    o_indep = oracle.indep(0.45, stops)
    prm = oracle.mg_params(**kw)
    reads = gpu.Reads.from_strings(seqs)
    orfs, starts, off = gpu.mg_score_reads(nc, gpu.Icm.indep(0.45, stops), reads, **kw)
    n_acc = n_trunc = n_boost = 0
    for r, seq in enumerate(seqs):
        want_orfs, scored = oracle.mg_read(o_nc, o_indep, seq.encode(), prm)
        mine = orfs[int(off[r]):int(off[r + 1])]
        got = np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1).reshape(-1, 4)
        assert np.array_equal(got, want_orfs), "Find_Orfs differs on read %d (len %d)" % (r, len(seq))
        assert np.all(mine["read"] == r)
        for o, (out, want) in zip(mine, scored):
            assert (o["lo"], o["hi"], o["orf_is_truncated"]) == (out.lo, out.hi, out.orf_is_truncated)
            st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
            assert len(st) == len(want)
            for s, w in zip(st, want):                                  # push order, field by field
                assert (s["j"], s["pos"], s["which"], s["truncated"], s["first"]) == (w.j, w.pos, w.which, w.truncated, w.first)
                assert s["score"] == w.score
                n_boost += int(w.j > prm.ignore_score_len and w.score == 0.0)
            assert (o["first_j"], o["accepted"] != 0) == (out.first_j, bool(out.accepted))
            assert o["best_score"] == out.best_score
            n_acc += int(out.accepted)
            n_trunc += out.orf_is_truncated
    assert n_acc > 10
    if kw.get("allow_truncated", True):
        assert n_trunc > 50
    if "ignore_score_len" in kw:
        assert n_boost > 0


def test_mg_frame_scores_are_handed_back_and_empty_batches_work(gpu, nc, seqs_fa):
    reads = gpu.Reads.from_strings(seqs_fa[1][:40])
    indep = gpu.Icm.indep(0.5)
    buf = gpu.api._DeviceBuffer(6 * reads.total_bases * 8)
    orfs, starts, off = gpu.mg_score_reads(nc, indep, reads, frame_scores=buf)
    table = buf.to_host(np.float64, 6 * reads.total_bases).reshape(6, -1)
    assert np.array_equal(table, gpu.frame_score6(nc, indep, reads))
    buf.free()
    assert len(orfs) == off[-1] > 100
    short = gpu.Reads.from_strings(["acgt", "ac"])                      # nothing reaches Min_Gene_Len
    orfs, starts, off = gpu.mg_score_reads(nc, indep, short)
    assert len(orfs) == 0 and len(starts) == 0 and list(off) == [0, 0, 0]
    with pytest.raises(gpu.GmgError):
        gpu.mg_score_reads(gpu.Icm.open(os.path.join(DATA, "cluster-4.icm")), indep, reads)     # periodicity 1


@pytest.mark.parametrize("flags,golden,fasta", [
    ([], "glimmer-mg.default.predict", "seqs.fa"), (["-g", "120"], "glimmer-mg.g120.predict", "seqs.fa"),
    (["-Z", "taa,tag"], "glimmer-mg.Z2.predict", "seqs.fa"),
    # the error branch: indels (predictions with I: / D: lists), substitutions (S:), indels with a quality file
    (["-i"], "glimmer-mg.indel.predict", "seqs.fa"), (["-s"], "glimmer-mg.sub.predict", "seqs.fa"),
    (["-i", "-q", os.path.join(DATA, "seqs80.qual")], "glimmer-mg.indel_q80.predict", "seqs80.fa")])
def test_glimmer_mg_with_device_front_half_is_byte_identical(gpu, tmp_path, flags, golden, fasta):
    """integration/glimmer-mg_gpu: glimmer-mg's own Add_Events / Process_Events / Trace_Back around ONE
    gmg_mg_score_reads call that replaces Score_All_Frames, Find_Orfs and Score_Orfs_Errors of all 999 reads."""
    exe = built_binary("integration", "_build", "glimmer-mg_gpu")
    tag = str(tmp_path / "out")
    cmd = [exe, *flags, "-m", os.path.join(DATA, "NC_000915.icm"), os.path.join(DATA, fasta), tag]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert open(tag + ".predict", "rb").read() == open(os.path.join(GOLD, "predict", golden), "rb").read()


@pytest.mark.parametrize("how", [["--batch-bytes", "50000"], ["--shards", "2"], ["--shards", "3", "--batch-bytes", "40000"],
                                 ["--shards", "4", "--gpus", "1"]])    # (at most 6 processes may hold the GPU of a test box: 4 + this one)
@pytest.mark.parametrize("flags,golden", [([], "glimmer-mg.default.predict"), (["-i"], "glimmer-mg.indel.predict")])
def test_glimmer_mg_gpu_in_batches_and_shards_is_byte_identical(gpu, tmp_path, how, flags, golden):
    """the same file in several batches per process and / or one forked process per shard (all on the one GPU of the box):
    the null model's GC fraction is the whole file's (summed from the shards' {gc, total}), the parts are concatenated in
    shard order -- the bytes of the reference's single run."""
    exe = built_binary("integration", "_build", "glimmer-mg_gpu")
    tag = str(tmp_path / "out")
    cmd = [exe, *how, *flags, "-m", os.path.join(DATA, "NC_000915.icm"), os.path.join(DATA, "seqs.fa"), tag]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert open(tag + ".predict", "rb").read() == open(os.path.join(GOLD, "predict", golden), "rb").read()
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]


@pytest.mark.parametrize("how", [["--shards", "3"], ["--shards", "2", "--batch-bytes", "9000"]])
def test_glimmer_mg_gpu_shards_with_a_quality_file(gpu, tmp_path, how):
    """-i -q with --shards: every shard passes over the quality records of the reads in front of its byte range (as many as there are
    header lines there) and reads its own in order -- the bytes of the reference's single run"""
    exe = built_binary("integration", "_build", "glimmer-mg_gpu")
    tag = str(tmp_path / "out")
    cmd = [exe, *how, "-i", "-q", os.path.join(DATA, "seqs80.qual"), "-m", os.path.join(DATA, "NC_000915.icm"), os.path.join(DATA, "seqs80.fa"), tag]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert open(tag + ".predict", "rb").read() == open(os.path.join(GOLD, "predict", "glimmer-mg.indel_q80.predict"), "rb").read()
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]


def test_shards_quality_file_and_records_that_begin_inside_a_line(gpu, tmp_path):
    """Fasta_Read (src/Common/fasta.cc:236-286) starts a record at ANY '>' outside a header line, also in the middle of a sequence line;
    gmg_fasta_ingest follows that rule, so a shard behind the first must count the records in front of it the same way when it passes
    over their quality records (counting header LINES gave every later shard the wrong records).  The reference's single
    run against --shards 3 / 4 on a file whose every fifth record begins inside a line."""
    rng = np.random.default_rng(77)
    ref, dev = built_binary("oracle", "_ref", "glimmer-mg"), built_binary("integration", "_build", "glimmer-mg_gpu")
    genome = "".join(line.strip() for line in open(os.path.join(DATA, "NC_000915.fna")) if not line.startswith(">")).lower()
    fa, ql = str(tmp_path / "in.fa"), str(tmp_path / "in.qual")
    with open(fa, "w") as f, open(ql, "w") as q:
        for i in range(90):
            n, at = int(rng.integers(200, 520)), int(rng.integers(0, 1_600_000))
            s = genome[at:at + n]
            inline, next_inline = i % 5 == 3, (i + 1) % 5 == 3
            f.write(">r%d%s\n" % (i, " inside a line" if inline else ""))       # (an inline header follows the last base of the record before it)
            lines = [s[k:k + 60] for k in range(0, n, 60)]
            f.write("\n".join(lines) + ("" if next_inline else "\n"))
            vals = np.where(rng.random(n) < 0.1, rng.integers(0, 19, n), rng.integers(19, 41, n))
            q.write(">r%d\n" % i)
            for k in range(0, n, 25):
                q.write(" ".join(str(int(v)) for v in vals[k:k + 25]) + "\n")
    assert "t>r3 inside" in open(fa).read() or "a>r3 inside" in open(fa).read() or "c>r3 inside" in open(fa).read() or "g>r3 inside" in open(fa).read()
    icm = os.path.join(DATA, "NC_000915.icm")
    res = subprocess.run([ref, "-i", "-q", ql, "-m", icm, fa, str(tmp_path / "ref")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    want = open(str(tmp_path / "ref") + ".predict", "rb").read()
    assert want.count(b">") == 90 and b">r3 inside a line" in want and want.count(b"orf") >= 60
    for shards in ("3", "4"):
        tag = str(tmp_path / ("dev" + shards))
        res = subprocess.run([dev, "--shards", shards, "-i", "-q", ql, "-m", icm, fa, tag], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        assert open(tag + ".predict", "rb").read() == want


def test_reads_select_groups_like_classification_mode(gpu, nc):
    """gmg_reads_select: the reads of one group gathered on the device (glimmer-mg -c scores every ICM's reads with that ICM
    and the classes' null model, glimmer-mg.cc:361-375); a group scored alone gives each read's records of the full batch"""
    rng = np.random.default_rng(5)
    lens = [0, 17, 300, 1, 512, 499, 0, 76, 1025, 33, 250, 250]
    seqs = random_reads(rng, lens)
    reads = gpu.Reads.from_strings(seqs)
    idx = [8, 2, 2, 0, 11, 5, 6, 4]
    sub = reads.select(idx)
    packed, off = sub.download()
    want = gpu.Reads.from_strings([seqs[i] for i in idx]).download()
    assert np.array_equal(off, want[1]) and np.array_equal(packed, want[0])
    assert reads.select([]).n_reads == 0
    with pytest.raises(gpu.GmgError):
        reads.select([3, len(seqs)])
    full = gpu.mg_score_reads(nc, gpu.Icm.indep(0.47), reads, min_gene_len=60)
    part = gpu.mg_score_reads(nc, gpu.Icm.indep(0.47), sub, min_gene_len=60)
    for k, i in enumerate(idx):
        a = full[0][int(full[2][i]):int(full[2][i + 1])]
        b = part[0][int(part[2][k]):int(part[2][k + 1])]
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert (x["frame"], x["stop_position"], x["n_starts"], x["accepted"]) == (y["frame"], y["stop_position"], y["n_starts"], y["accepted"])
            assert np.array_equal(full[1][x["start_begin"]:x["start_begin"] + x["n_starts"]], part[1][y["start_begin"]:y["start_begin"] + y["n_starts"]])


def test_mg_full_size_properties(gpu, oracle, nc):
    """BASELINE configs[1] shape: 1M x 500 bp (24 GB of Frame_Scores + 8 GB of running sums on the device, 0.8 GB of
    results).  Properties: (1) determinism: two calls give identical bytes; (2) bookkeeping: ORFs are stored read by
    read, start lists are back to back and exactly sum(n_starts) long, bounds lie inside the reads; (3) locality: a
    read scored alone gives the same records as its slice of the batch; (4) sampled reads equal the oracle."""
    n, L, seed = 1_000_000, 500, 7
    packed, off = gpu.synth.packed_reads(n, L, seed)
    reads = gpu.Reads(packed, off)
    indep = gpu.Icm.indep(0.5)
    orfs, starts, first = gpu.mg_score_reads(nc, indep, reads)
    orfs2, starts2, first2 = gpu.mg_score_reads(nc, indep, reads)
    assert orfs.tobytes() == orfs2.tobytes() and starts.tobytes() == starts2.tobytes() and np.array_equal(first, first2)
    del orfs2, starts2
    assert len(orfs) == first[-1] > 5 * n and np.all(np.diff(first.astype(np.int64)) >= 0)
    assert np.array_equal(orfs["read"], np.repeat(np.arange(n, dtype=np.uint32), np.diff(first.astype(np.int64))))
    assert int(orfs["n_starts"].sum()) == len(starts)
    assert np.array_equal(orfs["start_begin"], np.concatenate([[0], np.cumsum(orfs["n_starts"], dtype=np.uint64)[:-1]]))
    assert np.all(orfs["lo"] >= 0) and np.all(orfs["hi"] <= L + 1) and np.all(orfs["hi"] - orfs["lo"] >= 0)
    assert np.all(orfs["accepted"][orfs["n_starts"] == 0] == 0)
    o_nc, o_indep, prm = oracle.read(os.path.join(DATA, "NC_000915.icm")), oracle.indep(0.5), oracle.mg_params()
    rng = np.random.default_rng(11)
    sample = [0, 1, n - 1] + [int(x) for x in rng.integers(0, n, 40)]
    alone = gpu.Reads.from_strings([gpu.synth.unpack_ascii(packed, r * L, L).decode() for r in sample])
    a_orfs, a_starts, a_first = gpu.mg_score_reads(nc, indep, alone)
    for k, r in enumerate(sample):
        mine = orfs[int(first[r]):int(first[r + 1])]
        solo = a_orfs[int(a_first[k]):int(a_first[k + 1])]
        cols = ["frame", "stop_position", "orf_len", "gene_len", "lo", "hi", "first_j", "n_starts", "accepted", "best_score"]
        assert all(np.array_equal(mine[c], solo[c]) for c in cols)
        want_orfs, scored = oracle.mg_read(o_nc, o_indep, gpu.synth.unpack_ascii(packed, r * L, L), prm)
        assert np.array_equal(np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1), want_orfs)
        for o, so, (out, want) in zip(mine, solo, scored):
            st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
            assert st.tobytes() == a_starts[so["start_begin"]:so["start_begin"] + so["n_starts"]].tobytes()
            assert [(s["j"], s["pos"], s["which"], s["truncated"], s["first"], s["score"]) for s in st] == \
                   [(w.j, w.pos, w.which, w.truncated, w.first, w.score) for w in want]
            assert (o["first_j"], bool(o["accepted"]), o["best_score"]) == (out.first_j, bool(out.accepted), out.best_score)


@pytest.mark.parametrize("L", [7, 60, 100, 150, 512, 513, 1504, 1505])
def test_mg_uniform_read_lengths_exercise_every_tile_shape(gpu, oracle, nc, L):
    """uniform batches take whole reads per tile (several short reads per tile, one 512-bp read, 1,504-base tiles, and the
    per-lane kernel beyond that); every start score must still equal the oracle's"""
    rng = np.random.default_rng(L)
    n = 300 if L < 600 else 60
    seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=L)) for _ in range(n)]
    kw = dict(min_gene_len=30 if L >= 30 else 4)
    orfs, starts, off = gpu.mg_score_reads(nc, gpu.Icm.indep(0.5), gpu.Reads.from_strings(seqs), **kw)
    o_nc, o_indep, prm = oracle.read(os.path.join(DATA, "NC_000915.icm")), oracle.indep(0.5), oracle.mg_params(**kw)
    n_starts = 0
    for r in range(0, n, 7):
        want_orfs, scored = oracle.mg_read(o_nc, o_indep, seqs[r].encode(), prm)
        mine = orfs[int(off[r]):int(off[r + 1])]
        assert np.array_equal(np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1).reshape(-1, 4), want_orfs)
        for o, (out, want) in zip(mine, scored):
            st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
            assert [(s["j"], s["pos"], s["which"], s["score"]) for s in st] == [(w.j, w.pos, w.which, w.score) for w in want]
            assert o["best_score"] == out.best_score and bool(o["accepted"]) == bool(out.accepted)
            n_starts += len(want)
    assert n_starts > 0 or L < 30


@pytest.mark.parametrize("name,trunc", [("orfs_default", False), ("orfs_X", True)])
def test_find_orfs_alone_matches_glimmer3_goldens(gpu, seqs_fa, name, trunc):
    """gmg_find_orfs against the ORF lists the reference's own Find_Orfs produced inside glimmer3 (tests/golden/orfs_*.npz)"""
    g = np.load(os.path.join(GOLD, name + ".npz"))["orfs"]          # read, frame, stop_position, orf_len
    orfs, off = gpu.find_orfs(gpu.Reads.from_strings(seqs_fa[1]), min_gene_len=75, allow_truncated=trunc)
    got = np.stack([orfs["read"].astype(np.int32), orfs["frame"], orfs["stop_position"], orfs["orf_len"]], 1)
    assert np.array_equal(got, g)
    assert np.array_equal(off, np.searchsorted(g[:, 0], np.arange(len(seqs_fa[1]) + 1)))
    assert np.all(orfs["n_starts"] == 0)


def test_mg_accepted_only_is_the_filtered_full_result(gpu, nc, seqs_fa):
    """GMG_MG_ACCEPTED_ONLY: the same records and start lists as the full result restricted to accepted ORFs, same order,
    start lists re-packed, read offsets counting the kept ORFs"""
    rng = np.random.default_rng(9)
    seqs = list(seqs_fa[1][:200]) + ["".join("acgt"[c] for c in rng.integers(0, 4, size=int(n))) for n in rng.integers(1, 800, size=150)]
    reads = gpu.Reads.from_strings(seqs)
    indep = gpu.Icm.indep(0.42)
    orfs, starts, first = gpu.mg_score_reads(nc, indep, reads)
    k_orfs, k_starts, k_first = gpu.mg_score_reads(nc, indep, reads, accepted_only=True)
    keep = orfs["accepted"] != 0
    assert 0 < keep.sum() == len(k_orfs) < len(orfs)
    cols = ["read", "frame", "stop_position", "orf_len", "gene_len", "lo", "hi", "first_j", "n_starts", "accepted", "orf_is_truncated", "best_score"]
    assert all(np.array_equal(orfs[c][keep], k_orfs[c]) for c in cols)
    assert np.array_equal(k_orfs["start_begin"], np.concatenate([[0], np.cumsum(k_orfs["n_starts"], dtype=np.uint64)[:-1]]))
    assert len(k_starts) == int(k_orfs["n_starts"].sum())
    for o, k in zip(orfs[keep], k_orfs):
        assert starts[o["start_begin"]:o["start_begin"] + o["n_starts"]].tobytes() == \
               k_starts[k["start_begin"]:k["start_begin"] + k["n_starts"]].tobytes()
    assert np.array_equal(k_first, np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(np.int64), first[:-1].astype(np.int64))
                                                                   * (np.diff(first.astype(np.int64)) > 0))]).astype(np.uint64))


def test_empty_reads_inside_a_batch_are_harmless(gpu, oracle, nc):
    """FASTA files do contain empty records (Fasta_Read returns them): zero-length reads anywhere in a batch must not
    disturb their neighbours in any of the batch entry points"""
    rng = np.random.default_rng(4)
    body = ["".join("acgt"[c] for c in rng.integers(0, 4, size=int(n))) for n in (300, 520, 90, 700)]
    seqs = ["", body[0], "", "", body[1], body[2], "", body[3], ""]
    reads = gpu.Reads.from_strings(seqs)
    indep = gpu.Icm.indep(0.5)
    dense = gpu.Reads.from_strings(body)
    a = gpu.frame_score6(nc, indep, reads)
    assert np.array_equal(a, gpu.frame_score6(nc, indep, dense))                 # same bases, same table
    orfs, starts, first = gpu.mg_score_reads(nc, indep, reads, min_gene_len=60)
    d_orfs, d_starts, d_first = gpu.mg_score_reads(nc, indep, dense, min_gene_len=60)
    assert len(orfs) == len(d_orfs) > 10 and starts.tobytes() == d_starts.tobytes()
    assert np.array_equal(orfs["read"], np.array([1, 4, 5, 7])[d_orfs["read"]])
    for c in ("frame", "stop_position", "lo", "hi", "n_starts", "accepted", "best_score"):
        assert np.array_equal(orfs[c], d_orfs[c])
    assert list(np.diff(first.astype(np.int64))[[0, 2, 3, 6, 8]]) == [0, 0, 0, 0, 0]
    m = gpu.Icm.open(os.path.join(DATA, "cluster-1.icm"))
    s = gpu.score_reads_strings([m], reads)[0]
    assert np.all(s[[0, 2, 3, 6, 8]] == 0.0) and np.array_equal(s[[1, 4, 5, 7]], gpu.score_reads_strings([m], dense)[0])
    f_orfs, f_first = gpu.find_orfs(reads, min_gene_len=60, allow_truncated=True)
    assert np.array_equal(f_orfs["stop_position"], orfs["stop_position"]) and np.array_equal(f_first, first)


@pytest.mark.parametrize("shape,kw", [
    ("uniform 500", dict()),
    ("uniform 500", dict(allow_truncated=False, min_gene_len=90, ignore_score_len=150)),
    ("uniform 100", dict(min_gene_len=30)),
    ("uniform 9", dict(min_gene_len=4, start_threshold=-1e300)),
    ("uniform 567", dict()),
    ("uniform 568", dict(min_gene_len=4, ignore_score_len=10)),
    ("uniform 1134", dict()),
    ("uniform 2268", dict(min_gene_len=30, start_codons=("atg", "rtg", "ttg", "ctg"), stop_codons=("taa", "tag"))),
    ("uniform 2269", dict()),
    ("ragged 400", dict()),
    ("ragged 400", dict(min_gene_len=4, ignore_score_len=10, start_threshold=-1e300)),
    ("ragged 900", dict()),
    ("ragged 900", dict(min_gene_len=198)),                            # lowest j = 195: the last shape whose starts the ORF scan counts itself
    ("ragged 900", dict(min_gene_len=199, allow_truncated=False)),     # ... and the first that needs the count pass
    ("uniform 500", dict(min_gene_len=300)),
    ("ragged 30", dict(min_gene_len=4)),
    ("codons", dict(min_gene_len=4)),
    ("ragged 400", dict(min_gene_len=4, start_codons=("nnn",), stop_codons=("taa", "tag", "tga", "tta"))),   # start set and stop set overlap
])
def test_mg_fused_kernel_equals_the_sequential_kernels(gpu, nc, shape, kw):
    """k_mg_tile_starts (running sums as a parallel scan + start lists, option mg_fused = 1, the default when the models'
    values make every sum exact) against k_mg_cum_tiled / k_mg_cum + k_mg_starts (sequential sums in the reference's
    order): every byte of the ORF records and of the start lists, for every tile shape (mg_tile = 1, 2, 4 waves) and both
    forms of the table it reads"""
    kind, _, arg = shape.partition(" ")
    rng = np.random.default_rng(len(shape) * 1000 + len(kw))
    if kind == "uniform":
        L = int(arg)
        n = max(40, min(4000, 400_000 // L))
        packed, off = gpu.synth.packed_reads(n, L, 100 + L)
        reads = gpu.Reads(packed, off)
    elif kind == "ragged":
        mean = int(arg)
        lens = np.clip(rng.normal(mean, mean * 0.3, 1500).round(), 0, 2600).astype(np.int64)
        lens[::97] = 0
        lens[5::131] = 2500                                             # beyond every tile: the per-lane kernels
        reads = gpu.Reads.from_strings(random_reads(rng, lens))
    else:                                                               # nothing but start codons / stops back to back / no stop at all
        seqs = ["atg" * 170, "ttg" * 60 + "c", "cat" * 150 + "aa", "taa" * 100 + "a", "acg" * 200, "atgtaa" * 80,
                "g" + "ttacat" * 90, "atg" * 400, "cat" * 700] + random_reads(rng, [600, 3, 2, 1])
        reads = gpu.Reads.from_strings(seqs)
    stops = kw.get("stop_codons", ("taa", "tag", "tga"))
    indep = gpu.Icm.indep(0.45, stops)
    with gpu.option("mg_fused", 0), gpu.option("mg_orfs_events", 0):       # (and the ORF scan that visits every position)
        want = gpu.mg_score_reads(nc, indep, reads, **kw)
    assert len(want[0]) > 0
    for tile in (0, 1, 2, 4):
        if kind == "uniform" and tile and int(arg) > 567 * tile:
            continue
        for g32 in (1, 0):                                              # the call's own table as fp32 gene rows (the default) / as doubles
            with gpu.option("mg_tile", tile), gpu.option("mg_gene32", g32):
                got = gpu.mg_score_reads(nc, indep, reads, **kw)
            assert np.array_equal(got[2], want[2])
            for c in want[0].dtype.names:
                assert np.array_equal(got[0][c], want[0][c]), (shape, tile, g32, c)
            assert got[1].tobytes() == want[1].tobytes(), (shape, tile, g32)


@pytest.mark.parametrize("doctor", ["tiny", "zero"])
def test_mg_models_whose_sums_could_round_take_the_sequential_kernels(gpu, oracle, tmp_path, doctor):
    """the fused kernel and the event-only walks of the error branch change the ORDER of the additions, which is harmless only
    while every sum is exact; a gene model with a value of 3e-33 (exponent spread beyond the bound) or the logarithm of a zero
    probability (-FLT_MAX, icm.cc:1345-1349) must run the reference's order -- either way the starts are the oracle's, bit for bit"""
    src = os.path.join(DATA, "NC_000915.icm")
    raw = bytearray(open(src, "rb").read())
    _, prob = oracle.tables(oracle.read(src))
    at = bytes(raw).find(prob[0, 0].astype("<f4").tobytes())             # the root of sub-model 0: its four values
    assert at > 0
    vals = prob[0, 0].astype("<f4").copy()
    vals[1] = np.float32(-3.0e-33) if doctor == "tiny" else np.float32(-3.4028234663852886e38)
    raw[at:at + 16] = vals.tobytes()
    path = tmp_path / (doctor + ".icm")
    path.write_bytes(bytes(raw))
    rng = np.random.default_rng(9)
    seqs = random_reads(rng, list(rng.integers(30, 700, size=120)) + [500] * 30)
    reads = gpu.Reads.from_strings(seqs)
    gene, o_gene = gpu.Icm.open(str(path)), oracle.read(str(path))
    indep, o_indep = gpu.Icm.indep(0.5), oracle.indep(0.5)
    kw = dict(min_gene_len=30)
    orfs, starts, off = gpu.mg_score_reads(gene, indep, reads, **kw)
    prm = oracle.mg_params(**kw)
    n_starts = 0
    for r, seq in enumerate(seqs):
        _, scored = oracle.mg_read(o_gene, o_indep, seq.encode(), prm)
        mine = orfs[int(off[r]):int(off[r + 1])]
        for o, (out, want) in zip(mine, scored):
            st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
            assert [(s["j"], s["pos"], s["which"], s["score"]) for s in st] == [(w.j, w.pos, w.which, w.score) for w in want], (doctor, r)
            assert o["best_score"] == out.best_score
            n_starts += len(want)
    assert n_starts > 100
    # the error branch on the same model: every start list against the oracle's recursion
    e_orfs, e_starts, e_off, e_errs = gpu.mg_score_reads(gene, indep, reads, allow_indels=True, **kw)
    ep = oracle.mg_err_params(allow_indels=True)
    from test_gpu_mg_err import dev_err_rows, err_rows
    for r in range(0, len(seqs), 5):
        _, _, scored = oracle.mg_read_errors(o_gene, o_indep, seqs[r].encode(), prm, ep)
        mine = e_orfs[int(e_off[r]):int(e_off[r + 1])]
        assert len(mine) == len(scored)
        for o, (out, want) in zip(mine, scored):
            sl = slice(o["start_begin"], o["start_begin"] + o["n_starts"])
            assert dev_err_rows(e_starts[sl], e_errs[sl]) == err_rows(want), (doctor, "-i", r)


# ---- Find_Orfs in full on the device: ignore regions (glimmer3 -i) and circular sequences (glimmer-mg -r) ----------------------
@pytest.mark.parametrize("name", ["ignore", "ignore_X_g60", "circular", "circular_X_Z2", "circular_ignore", "plain_g60"])
def test_find_orfs_ignore_regions_and_circular_on_the_device(gpu, oracle, name):
    """gmg_find_orfs with gmg_mg_params.n_ignore_regions / .circular (k_find_orfs_general) against the reference's own Find_Orfs on six
    genome slices (tests/golden/find_orfs_general.npz: oracle/_ref/ref_orfs orfs | orfs-circular) and against the oracle on random
    sequences with random regions: every Orf_t field, in the reference's order"""
    from test_oracle_orfs import FIND_ORFS_GENERAL, genome_slices
    g, slices = genome_slices(oracle)
    kw, circular = FIND_ORFS_GENERAL[name]
    regions = [tuple(int(x) for x in r) for r in g[name + "_regions"]]
    reads = gpu.Reads.from_strings(slices)
    orfs, off = gpu.find_orfs(reads, circular=circular, ignore_regions=regions, **kw)
    got = np.stack([orfs["read"].astype(np.int32), orfs["frame"], orfs["stop_position"], orfs["gene_len"], orfs["orf_len"]], 1)
    assert np.array_equal(got, g[name + "_orfs"])
    assert int(off[-1]) == len(orfs) > 400
    # random sequences (0 .. 3,000 bases), random disjoint regions, both modes: the oracle, sequence by sequence
    rng = np.random.default_rng(len(name) * 7 + len(regions))
    seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=int(n))) for n in rng.integers(0, 3000, 40)]
    cuts = np.sort(rng.choice(np.arange(1, 2900), size=8, replace=False))
    rnd_regions = [(int(cuts[k]), int(cuts[k + 1])) for k in range(0, 8, 2)] if regions else []
    prm = oracle.mg_params(**kw)
    want = [oracle.find_orfs_general(s, prm, circular=circular, regions=rnd_regions) for s in seqs]
    keep = [k for k, w in enumerate(want) if w is not None]          # (None: the reference's assert in Wrap_Around_Back)
    reads = gpu.Reads.from_strings([seqs[k] for k in keep])
    orfs, off = gpu.find_orfs(reads, circular=circular, ignore_regions=rnd_regions, **kw)
    for j, k in enumerate(keep):
        mine = orfs[int(off[j]):int(off[j + 1])]
        assert np.array_equal(np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1).reshape(-1, 4), want[k]), (name, k)
    for k, w in enumerate(want):                            # where the reference aborts the call refuses (and says why)
        if w is None:
            with pytest.raises(gpu.GmgError) as e:
                gpu.find_orfs(gpu.Reads.from_strings([seqs[k]]), circular=circular, ignore_regions=rnd_regions, **kw)
            assert "Wrap_Around_Back" in str(e.value)
            break
    # the front half itself does not take such calls
    with pytest.raises(gpu.GmgError):
        gpu.find_orfs(reads, ignore_regions=[(50, 40)], **kw)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 1023, 1024, 4095, 4096, 4097, 8191, 8193, 65537, 300001])
def test_prefix_sums_at_tile_and_chunk_edges(gpu, n):
    """gmg_scan.h (tiles of 4,096 items, 16-byte chunks of four; 32-bit and 64-bit items): the offsets of a selection of n reads
    (64-bit lengths -> offsets) and the ORF offsets of gmg_find_orfs (32-bit counts -> 64-bit offsets) against numpy's cumsum"""
    rng = np.random.default_rng(n)
    base_lens = rng.integers(0, 120, size=257)
    seqs = random_reads(rng, base_lens)
    reads = gpu.Reads.from_strings(seqs)
    idx = rng.integers(0, len(seqs), size=n)
    sub = reads.select(idx)
    _, off = sub.download()
    assert np.array_equal(off, np.concatenate([[0], np.cumsum(base_lens[idx])]).astype(np.uint64))
    orfs, orf_off = gpu.find_orfs(sub, min_gene_len=33, allow_truncated=True)
    counts = np.bincount(orfs["read"].astype(np.int64), minlength=n) if len(orfs) else np.zeros(n, np.int64)
    assert np.array_equal(orf_off, np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)) and int(orf_off[-1]) == len(orfs)
    assert np.all(np.diff(orfs["read"].astype(np.int64)) >= 0)
