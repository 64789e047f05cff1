"""glimmer-mg's classification mode (-c, SURVEY 8(f) #3: the ICM-grouped loop, glimmer-mg.cc:361-451) end to end on the device:

  * integration/glimmer-mg_gpu -c writes <tag>.predict byte for byte as the REAL reference did (tests/golden/predict/classes.*:
    the reference's own main() over the synthetic .genomeData tree, six option sets incl. -i, -s, chunks of 250 reads, and
    -m together with -c);
  * the product's loop in Python -- gmg_classes_plan -> per ICM group and stop-codon set ONE gmg_mg_score_reads with a null
    model per distinct GC (gmg_null_set_build) and Ignore_Score_Len per read -- gives, for every processed read in the
    reference's order, the ORFs of the reference's Find_Orfs and, for every ORF the reference handed to Add_Events_*, its start
    list in the order Score_Orf_Starts pushed it (scores bit for bit, Error_t lists with -i / -s);
  * the same loop against the oracle (its own plan, its own scoring) for EVERY ORF of every read."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA, GOLD, built_binary

sys.path.insert(0, GOLD)
import make_genome_data  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def genome_data(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("genomeData"))
    mixed = make_genome_data.write_class_variants(d)
    make_genome_data.build(os.path.join(d, ".genomeData"), [os.path.join(DATA, "seqs.class.txt"), mixed])
    return d


@pytest.fixture(autouse=True)
def in_genome_data_dir(genome_data, monkeypatch):
    monkeypatch.chdir(genome_data)                      # ICM_dir = ".genomeData", as the goldens were made (the order depends on the name)


CLI_CASES = [
    ("default", [], "seqs.class.txt", None), ("indel", ["-i"], "seqs.class.txt", None), ("g90", ["-g", "90"], "seqs.class.txt", None),
    ("mixed_chunks", [], "mixed.class.txt", 250), ("mixed_sub", ["-s"], "mixed.class.txt", None),
    ("user_icm", ["-m", os.path.join(DATA, "NC_000915.icm")], "seqs.class.txt", None),
]


@pytest.mark.parametrize("name,flags,cls,chunk", CLI_CASES)
def test_glimmer_mg_gpu_classification_mode_is_byte_identical(gpu, tmp_path, name, flags, cls, chunk):
    exe = built_binary("integration", "_build", "glimmer-mg_gpu")
    tag = str(tmp_path / "out")
    own = ["--icm-dir", ".genomeData"] + (["--chunk-reads", str(chunk)] if chunk else [])
    cmd = [exe, *own, *flags, "-c", os.path.join(DATA, cls), os.path.join(DATA, "seqs.fa"), tag]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert open(tag + ".predict", "rb").read() == open(os.path.join(GOLD, "predict", "classes.%s.predict" % name), "rb").read()


@pytest.mark.parametrize("name,flags,cls,own", [
    ("default", [], "seqs.class.txt", ["--shards", "2"]),
    ("mixed_chunks", [], "mixed.class.txt", ["--shards", "3", "--chunk-reads", "250"]),
    ("indel", ["-i"], "seqs.class.txt", ["--shards", "4", "--gpus", "1"]),
    ("user_icm", ["-m", os.path.join(DATA, "NC_000915.icm")], "seqs.class.txt", ["--shards", "2"])])
def test_glimmer_mg_gpu_classification_mode_in_shards_is_byte_identical(gpu, tmp_path, name, flags, cls, own):
    """--shards N with -c: every forked shard ingests the whole file, plans every chunk and takes its run of each chunk's visiting
    order; the parent concatenates chunk by chunk, shard by shard -- the reference's bytes, no part file left behind"""
    exe = built_binary("integration", "_build", "glimmer-mg_gpu")
    tag = str(tmp_path / "out")
    cmd = [exe, "--icm-dir", ".genomeData", *own, *flags, "-c", os.path.join(DATA, cls), os.path.join(DATA, "seqs.fa"), tag]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert open(tag + ".predict", "rb").read() == open(os.path.join(GOLD, "predict", "classes.%s.predict" % name), "rb").read()
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]


def chunks(n, size):
    return [(b, min(n, b + size)) for b in range(0, n, size)] if size else [(0, n)]


def score_groups(gpu, cls, hdrs, seqs, chunk, models, grouped=True, **kw):
    """the product's -c loop: yields, per processed read in the reference's order,
    (chunk-global read index, gc, stops, isl, orfs, starts, errs-or-None).  grouped: ONE gmg_mg_score_groups call per chunk and
    stop-codon set (every ICM group a consecutive range of the gathered batch); else one gmg_mg_score_reads per (ICM, stop set)"""
    err = bool(kw.get("allow_indels") or kw.get("allow_subs"))
    for b, e in chunks(len(hdrs), chunk):
        reads = gpu.Reads.from_strings(seqs[b:e])
        order, icm_begin, gc, transl = cls.plan(hdrs[b:e])
        group_of = np.searchsorted(icm_begin, np.arange(len(order)), side="right") - 1
        for f in set(group_of.tolist()):
            path = os.path.realpath(cls.icm_file(f))
            if path not in models:
                models[path] = gpu.Icm.open(path)
        per_read = {}
        for code in sorted(set(transl.tolist())):
            stops = gpu.api.stop_codons_by_code(code)
            jobs = []                                   # [(positions k, groups or model)]
            ks_all = [k for k in range(len(order)) if transl[k] == code]
            if grouped:
                groups, r = [], 0
                for f in sorted(set(group_of[ks_all].tolist())):
                    n = int(np.sum(group_of[ks_all] == f))
                    groups.append((models[os.path.realpath(cls.icm_file(f))], r, r + n))
                    r += n
                jobs.append((ks_all, groups))
            else:
                for f in sorted(set(group_of[ks_all].tolist())):
                    jobs.append(([k for k in ks_all if group_of[k] == f], models[os.path.realpath(cls.icm_file(f))]))
            for ks, what in jobs:
                ugc, inv = np.unique(gc[ks], return_inverse=True)
                isl = np.array([gpu.api.ignore_score_len(gc[k], stops) for k in ks], np.int32)
                batch = reads.select([int(order[k]) for k in ks])
                extra = dict(groups=what) if grouped else {}
                res = gpu.mg_score_reads(None if grouped else what, gpu.NullSet.build(ugc, stops), batch, read_null=inv.astype(np.uint32),
                                         read_ignore_score_len=isl, stop_codons=stops, **extra, **kw)
                for r, k in enumerate(ks):
                    o = res[0][int(res[2][r]):int(res[2][r + 1])]
                    per_read[k] = (b + int(order[k]), float(gc[k]), stops, int(isl[r]), o, res[1], res[3] if err else None)
        for k in range(len(order)):
            yield per_read[k]


@pytest.mark.parametrize("grouped", [True, False])
@pytest.mark.parametrize("name", ["default", "indel", "g90", "mixed_chunks", "mixed_sub"])
def test_grouped_scoring_equals_the_reference_read_by_read(gpu, seqs_fa, genome_data, name, grouped):
    g = np.load(os.path.join(GOLD, "classes_%s.npz" % name))
    flags = str(g["flags"]).split()
    kw = dict(allow_indels="-i" in flags, allow_subs="-s" in flags)
    if "-g" in flags:
        kw["min_gene_len"] = int(flags[flags.index("-g") + 1])
    hdrs, seqs = seqs_fa
    cls = gpu.api.Classes(open(os.path.join(DATA, str(g["class_file"]))).read(), ".genomeData")
    acc_of = {}
    for a, r in enumerate(g["acc_read"]):
        acc_of.setdefault(int(r), []).append(a)
    n_lists = n_starts = 0
    k = -1
    for k, (ri, gc, stops, isl, orfs, starts, errs) in enumerate(score_groups(gpu, cls, hdrs, seqs, int(g["chunk"]), {}, grouped=grouped, **kw)):
        assert hdrs[ri].split()[0] == str(g["reads"][k])
        assert (gc, ",".join(stops), isl) == (float(g["gc"][k]), str(g["stops"][k]), int(g["isl"][k]))
        want = g["orfs"][int(g["orf_off"][k]):int(g["orf_off"][k + 1])]
        got = np.stack([orfs["frame"], orfs["stop_position"], orfs["gene_len"], orfs["orf_len"]], 1).reshape(-1, 4)
        assert np.array_equal(got, want), (name, k)                           # Find_Orfs with this read's stop codons
        if k >= int(g["list_reads"]):
            continue
        mine = [o for o in orfs if o["accepted"]]
        assert [(int(o["frame"]), int(o["stop_position"]), int(o["n_starts"])) for o in mine] == \
               [tuple(int(x) for x in g["acc"][a]) for a in acc_of.get(k, [])], (name, k)
        for o, a in zip(mine, acc_of.get(k, [])):
            sl = slice(int(o["start_begin"]), int(o["start_begin"]) + int(o["n_starts"]))
            gl = slice(int(g["st_off"][a]), int(g["st_off"][a + 1]))
            st = starts[sl]
            assert np.array_equal(st["j"], g["st_j"][gl]) and np.array_equal(st["pos"], g["st_pos"][gl])
            assert np.array_equal(st["score"], g["st_score"][gl])                  # doubles, bit for bit
            assert np.array_equal(st["which"], g["st_which"][gl]) and np.array_equal(st["truncated"], g["st_trunc"][gl])
            assert np.array_equal(st["first"], g["st_first"][gl])
            if errs is not None:
                e = errs[sl]
                assert np.array_equal(e["n"], g["st_nerr"][gl])
                for c in range(2):
                    used = g["st_nerr"][gl] > c
                    assert np.array_equal(e["pos"][:, c][used], g["st_epos"][gl][:, c][used])
                    assert np.array_equal(e["type"][:, c][used], g["st_etype"][gl][:, c][used])
            n_lists += 1
            n_starts += gl.stop - gl.start
    assert k + 1 == len(g["reads"]) and n_lists == len(g["acc"]) and n_starts == len(g["st_j"])


def test_grouped_scoring_equals_the_oracle_for_every_orf(gpu, oracle, seqs_fa, genome_data):
    """the oracle end to end: its own plan (orc_classes_*), its own null model per read, its own Score_Orf_Starts -- against the
    product's grouped calls, every ORF of every processed read of mixed.class.txt in chunks of 300"""
    hdrs, seqs = seqs_fa
    text = open(os.path.join(DATA, "mixed.class.txt")).read()
    cls = gpu.api.Classes(text, ".genomeData")
    oc = oracle.classes_load(text, ".genomeData")
    o_files = oracle.classes_icm_files(oc)
    want_order = []
    for b, e in chunks(len(hdrs), 300):
        order, icm_begin, gc, transl = oracle.classes_plan(oc, hdrs[b:e])
        for f in range(len(o_files)):
            for k in range(int(icm_begin[f]), int(icm_begin[f + 1])):
                want_order.append((b + int(order[k]), float(gc[k]), int(transl[k]), os.path.realpath(o_files[f])))
    o_models, n_orfs, n_starts, n_acc = {}, 0, 0, 0
    got = list(score_groups(gpu, cls, hdrs, seqs, 300, {}))
    assert len(got) == len(want_order) > 800
    for (ri, gc, stops, isl, orfs, starts, _), (w_ri, w_gc, w_tt, w_path) in zip(got, want_order):
        assert (ri, gc) == (w_ri, w_gc) and stops == oracle.stop_codons_by_code(w_tt)
        assert isl == oracle.ignore_score_len(w_gc, stops)
        if w_path not in o_models:
            o_models[w_path] = oracle.read(w_path)
        prm = oracle.mg_params(ignore_score_len=isl, stop_codons=stops)
        want_orfs, scored = oracle.mg_read(o_models[w_path], oracle.indep(w_gc, stops), seqs[ri].encode(), prm)
        assert np.array_equal(np.stack([orfs["frame"], orfs["stop_position"], orfs["gene_len"], orfs["orf_len"]], 1).reshape(-1, 4), want_orfs)
        for o, (out, want) in zip(orfs, scored):
            st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
            assert [(s["j"], s["pos"], s["which"], s["truncated"], s["first"], s["score"]) for s in st] == \
                   [(w.j, w.pos, w.which, w.truncated, w.first, w.score) for w in want]
            assert (o["first_j"], bool(o["accepted"]), o["best_score"]) == (out.first_j, bool(out.accepted), out.best_score)
            n_starts += len(want)
            n_acc += int(out.accepted)
        n_orfs += len(orfs)
    assert n_orfs > 5000 and n_starts > 10000 and n_acc > 500
    oracle.L.orc_classes_free(oc)


def test_null_set_build_equals_uploaded_models(gpu, seqs_fa):
    """gmg_null_set_build (tables made on the host, one copy) = gmg_null_set_upload of the same models one by one"""
    nc = gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))
    reads = gpu.Reads.from_strings(seqs_fa[1][:64])
    gcs = np.linspace(0.21, 0.79, 300)
    rn = (np.arange(64) * 7 % 300).astype(np.uint32)
    for stops in (("taa", "tag", "tga"), ("taa", "tag"), ("tga",)):
        a = gpu.frame_score6(nc, gpu.NullSet.build(gcs, stops), reads, read_null=rn)
        b = gpu.frame_score6(nc, gpu.NullSet([gpu.Icm.indep(float(x), stops) for x in gcs]), reads, read_null=rn)
        assert a.tobytes() == b.tobytes()


def ragged(rng, lengths):
    return ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]


GICMS = ["NC_000915.icm", "seqs.cluster-0.run1.filt.gicm", "seqs.cluster-2.run1.filt.gicm", "seqs.cluster-4.run1.filt.gicm",
         "seqs.cluster-5.run1.filt.gicm"]


@pytest.mark.parametrize("shape", ["uniform500", "ragged", "tiny_groups", "long_reads", "one_group", "other_depth"])
@pytest.mark.parametrize("mode", ["default", "indel", "fp64_table"])
def test_score_groups_equals_one_call_per_group(gpu, shape, mode):
    """gmg_mg_score_groups (ONE six-frame launch that swaps the groups' tables in LDS) = gmg_mg_score_reads group by group,
    bytes of every record and start: group edges inside chunks, groups smaller than a chunk, empty groups, reads longer than a
    chunk, a model of another depth among the groups (the any-shape kernel), the error branch and the fp64 table form"""
    import zlib
    rng = np.random.default_rng(zlib.crc32((shape + mode).encode()))
    if shape == "uniform500":
        lens, cuts = [500] * 6000, [0, 1, 700, 701, 2500, 2500, 4097, 6000]
    elif shape == "ragged":
        lens = [int(x) for x in rng.integers(0, 900, 5000)]
        cuts = [0, 900, 1800, 1803, 1803, 3333, 5000]
    elif shape == "tiny_groups":
        lens = [int(x) for x in rng.integers(30, 400, 600)]
        cuts = list(range(0, 600, 3)) + [600]
    elif shape == "long_reads":
        lens = [int(x) for x in rng.integers(1500, 9000, 300)]
        cuts = [0, 50, 51, 200, 300]
    elif shape == "one_group":
        lens, cuts = [int(x) for x in rng.integers(100, 700, 3000)], [0, 3000]
    else:
        lens, cuts = [int(x) for x in rng.integers(100, 700, 3000)], [0, 1000, 2000, 3000]
    if mode == "indel":
        lens, cuts = lens[:len(lens) // 6], [c // 6 for c in cuts]
    seqs = ragged(rng, lens)
    reads = gpu.Reads.from_strings(seqs)
    files = list(GICMS)
    if shape == "other_depth":
        files = ["NC_000915.icm", os.path.join("..", "train", "syn_d4.icm"), "seqs.cluster-5.run1.filt.gicm"]
    models = [gpu.Icm.open(os.path.join(DATA, f)) for f in files]
    groups = [(models[g % len(models)], cuts[g], cuts[g + 1]) for g in range(len(cuts) - 1)]
    gcs = np.linspace(0.3, 0.7, 37)
    nulls = gpu.NullSet.build(gcs)
    read_null = rng.integers(0, len(gcs), len(seqs)).astype(np.uint32)
    read_isl = rng.choice([2 ** 31 - 1, 200, 90], len(seqs)).astype(np.int32)
    kw = dict(min_gene_len=60, allow_indels=mode == "indel")
    with gpu.option("mg_gene32", 0 if mode == "fp64_table" else 1):
        whole = gpu.mg_score_reads(None, nulls, reads, read_null=read_null, read_ignore_score_len=read_isl, groups=groups, **kw)
        n_orfs = n_starts = 0
        for m, b, e in groups:
            if b == e:
                continue
            part = gpu.mg_score_reads(m, nulls, reads.select(np.arange(b, e, dtype=np.uint64)), read_null=read_null[b:e],
                                      read_ignore_score_len=read_isl[b:e], **kw)
            o0, o1 = int(whole[2][b]), int(whole[2][e])
            mine = whole[0][o0:o1].copy()
            assert len(mine) == len(part[0])
            if len(mine) == 0:
                continue
            s0 = int(mine["start_begin"][0])
            s1 = int(mine["start_begin"][-1]) + int(mine["n_starts"][-1])
            mine["read"] -= b                            # the two fields that count from the batch's start
            mine["start_begin"] -= s0
            assert mine.tobytes() == part[0].tobytes(), (shape, b, e)
            assert whole[1][s0:s1].tobytes() == part[1].tobytes(), (shape, b, e)
            if mode == "indel":
                assert whole[3][s0:s1].tobytes() == part[3].tobytes()
            n_orfs += len(mine)
            n_starts += s1 - s0
    assert n_orfs == len(whole[0]) and n_starts == len(whole[1]) and n_starts > 100


def test_score_groups_refuses_bad_group_lists(gpu):
    rng = np.random.default_rng(1)
    reads = gpu.Reads.from_strings(ragged(rng, [300] * 10))
    m = gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))
    nulls = gpu.NullSet.build([0.5])
    rn = np.zeros(10, np.uint32)
    for bad in ([(m, 0, 5)], [(m, 0, 5), (m, 6, 10)], [(m, 1, 10)], [(m, 0, 11)], [],
                [(m, 0, 5), (gpu.Icm.open(os.path.join(DATA, "cluster-4.icm")), 5, 10)]):      # ... a periodicity-1 model
        with pytest.raises(gpu.GmgError):
            gpu.mg_score_reads(None, nulls, reads, read_null=rn, groups=bad)
    with pytest.raises(gpu.GmgError):                   # groups need the per-read null models
        gpu.mg_score_reads(None, gpu.Icm.indep(0.5), reads, groups=[(m, 0, 10)])


def test_score_groups_full_size_with_error_branch(gpu, oracle, tmp_path):
    """BASELINE configs[4] as glimmer-mg.py runs it: -c together with -i on 1M reads of ~400 bp -- 64 ICM groups under 64 DISTINCT
    3-periodic gene models (the five sample files + 59 trained on disjoint slices of NC_000915.fna, tests/models64.py: 64 MB of
    tables against 4 MB of L2 per XCD), 100 GC values (a null model per read), Ignore_Score_Len per read, accepted ORFs only, ONE
    gmg_mg_score_groups call.
    (1) two calls give the same bytes; (2) the records are back to back; (3) three whole groups equal gmg_mg_score_reads on the
    gathered group alone; (4) sampled reads of other groups equal the oracle with the group's ICM and the read's null model."""
    from test_gpu_mg_err import dev_err_rows, err_rows
    n, n_groups, n_gc = 1_000_000, 64, 100
    lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    packed, _ = gpu.synth.packed_reads(1, int(off[-1]), 7)
    reads = gpu.Reads(packed, off)
    rng = np.random.default_rng(5)
    cuts = np.concatenate([[0], np.sort(rng.choice(np.arange(1, n), n_groups - 1, replace=False)), [n]]).astype(np.int64)
    import models64
    pairs = models64.gene_models(gpu, tmp_path, n_groups)
    models = [m for m, _ in pairs]
    groups = [(models[g], int(cuts[g]), int(cuts[g + 1])) for g in range(n_groups)]
    gcs = np.linspace(0.3, 0.7, n_gc)
    nulls = gpu.NullSet.build(gcs)
    read_null = rng.integers(0, n_gc, n).astype(np.uint32)
    read_isl = rng.choice([2 ** 31 - 1, 300, 150], n).astype(np.int32)
    kw = dict(allow_indels=True, accepted_only=True)
    orfs, starts, first, errs = gpu.mg_score_reads(None, nulls, reads, read_null=read_null, read_ignore_score_len=read_isl, groups=groups, **kw)
    again = gpu.mg_score_reads(None, nulls, reads, read_null=read_null, read_ignore_score_len=read_isl, groups=groups, **kw)
    for x, y in zip((orfs, starts, first, errs), again):
        assert x.tobytes() == y.tobytes()
    del again
    assert len(orfs) == first[-1] > n // 10 and np.all(orfs["accepted"] != 0)
    assert np.array_equal(orfs["start_begin"], np.concatenate([[0], np.cumsum(orfs["n_starts"].astype(np.int64))[:-1]]))
    assert int(orfs["n_starts"].astype(np.int64).sum()) == len(starts) == len(errs)
    assert np.all(np.diff(orfs["read"].astype(np.int64)) >= 0)
    for g in (0, 31, 63):
        m, b, e = groups[g]
        part = gpu.mg_score_reads(m, nulls, reads.select(np.arange(b, e, dtype=np.uint64)), read_null=read_null[b:e],
                                  read_ignore_score_len=read_isl[b:e], **kw)
        o0, o1 = int(first[b]), int(first[e])
        mine = orfs[o0:o1].copy()
        assert len(mine) == len(part[0]) > 0
        s0, s1 = int(mine["start_begin"][0]), int(mine["start_begin"][-1]) + int(mine["n_starts"][-1])
        mine["read"] -= b
        mine["start_begin"] -= s0
        assert mine.tobytes() == part[0].tobytes() and starts[s0:s1].tobytes() == part[1].tobytes() and errs[s0:s1].tobytes() == part[3].tobytes()
    ep = oracle.mg_err_params(allow_indels=True)
    checked = 0
    for g in (1, 2, 3, 4, 17, 40, 62):
        o_model = oracle.read(pairs[g][1])              # (the group's own table, from the file the device model was written to)
        for r in (groups[g][1], groups[g][2] - 1):      # the group's first and last read: the model changes right there
            seq = gpu.synth.unpack_ascii(packed, int(off[r]), int(off[r + 1] - off[r]))
            prm = oracle.mg_params(ignore_score_len=int(read_isl[r]))
            want_orfs, _, scored = oracle.mg_read_errors(o_model, oracle.indep(float(gcs[read_null[r]])), seq, prm, ep)
            acc = [(o, out, st) for o, (out, st) in zip(want_orfs, scored) if out.accepted]
            mine = orfs[int(first[r]):int(first[r + 1])]
            assert len(mine) == len(acc)
            for d, (o, out, st) in zip(mine, acc):
                assert (int(d["frame"]), int(d["stop_position"])) == (int(o[0]), int(o[1]))
                sl = slice(d["start_begin"], d["start_begin"] + d["n_starts"])
                assert dev_err_rows(starts[sl], errs[sl]) == err_rows(st)
                checked += 1
    assert checked >= 5


def test_score_groups_with_more_groups_than_65535(gpu):
    """a chunk of glimmer-mg -c may meet one ICM file per read (500,000 reads per chunk): 70,000 groups of one read each, five tables
    in turn -- every read's records must be those of a single-ICM call with its group's table"""
    rng = np.random.default_rng(77)
    n = 70_000
    seqs = ragged(rng, [int(x) for x in rng.integers(40, 300, n)])
    reads = gpu.Reads.from_strings(seqs)
    models = [gpu.Icm.open(os.path.join(DATA, f)) for f in GICMS]
    groups = [(models[g % len(models)], g, g + 1) for g in range(n)]
    gcs = np.linspace(0.35, 0.65, 11)
    nulls = gpu.NullSet.build(gcs)
    read_null = rng.integers(0, len(gcs), n).astype(np.uint32)
    read_isl = np.full(n, 2 ** 31 - 1, np.int32)
    kw = dict(min_gene_len=60, read_null=read_null, read_ignore_score_len=read_isl)
    whole = gpu.mg_score_reads(None, nulls, reads, groups=groups, **kw)
    fields = [f for f in whole[0].dtype.names if f not in ("read", "start_begin")]
    seen = 0
    for m in range(len(models)):
        one = gpu.mg_score_reads(models[m], nulls, reads, **kw)
        for r in range(m, n, len(models) * 97):         # a sample of this table's reads
            a = whole[0][int(whole[2][r]):int(whole[2][r + 1])]
            b = one[0][int(one[2][r]):int(one[2][r + 1])]
            assert len(a) == len(b)
            for f in fields:
                assert a[f].tobytes() == b[f].tobytes(), (r, f)
            for x, y in zip(a, b):
                sa = whole[1][int(x["start_begin"]):int(x["start_begin"]) + int(x["n_starts"])]
                sb = one[1][int(y["start_begin"]):int(y["start_begin"]) + int(y["n_starts"])]
                assert sa.tobytes() == sb.tobytes(), r
                seen += len(sa)
    assert seen > 100


@pytest.mark.parametrize("own", [[], ["--shards", "1", "--chunk-reads", "30"]])
def test_classification_mode_with_a_quality_file_equals_the_reference_run_here(gpu, tmp_path, own):
    """-c -i -q: the quality file is read chunk by chunk in FILE order (glimmer-mg.cc:334-359) while the reads are visited in the
    ICM groups' order -- the reference's own main() (oracle/_ref/ref_mg_classes, run here) against glimmer-mg_gpu, 80 reads with
    Phred values, also in chunks of 30 reads"""
    ref = built_binary("oracle", "_ref", "ref_mg_classes")
    dev = built_binary("integration", "_build", "glimmer-mg_gpu")
    common = ["-i", "-q", os.path.join(DATA, "seqs80.qual"), "-c", os.path.join(DATA, "seqs.class.txt"), os.path.join(DATA, "seqs80.fa")]
    chunk = own[own.index("--chunk-reads") + 1] if "--chunk-reads" in own else None
    env = dict(os.environ, GMG_REF_ICM_DIR=".genomeData", GMG_REF_QUIET="1", **({"GMG_REF_CHUNK": chunk} if chunk else {}))
    res = subprocess.run([ref, *common, str(tmp_path / "ref")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    res = subprocess.run([dev, "--icm-dir", ".genomeData", *own, *common, str(tmp_path / "dev")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    want = open(str(tmp_path / "ref") + ".predict", "rb").read()
    assert open(str(tmp_path / "dev") + ".predict", "rb").read() == want and want.count(b"orf") > 20


OPTION_SETS = [
    ["-Z", "taa,tag"], ["-z", "4"], ["-b", os.path.join(DATA, "seqs.cluster-2.run1.filt.motif")],
    ["-f", os.path.join(DATA, "NC_000915.run1.features.txt")], ["-g", "60", "-o", "50"], ["-u", "2.5", "-s"],
    ["-i", "-Z", "tga", "-g", "99"],
]


@pytest.mark.parametrize("mode", ["classes", "one_icm"])
@pytest.mark.parametrize("opts", OPTION_SETS, ids=["_".join(o[:2]).replace(",", "").replace("/", "") [:24] for o in OPTION_SETS])
def test_driver_options_against_the_reference_run_here(gpu, tmp_path, opts, mode):
    """every glimmer-mg option that changes what is scored or how the events are weighed (-Z / -z stop codons for all reads, -b RBS
    matrix, -f feature file, -g, -o, -u, -s, -i) with -c and with -m: the reference binary run in the test against glimmer-mg_gpu on
    the first 80 reads"""
    fa = os.path.join(DATA, "seqs80.fa")
    if mode == "classes":
        ref = built_binary("oracle", "_ref", "ref_mg_classes")
        sel, own = ["-c", os.path.join(DATA, "seqs.class.txt")], ["--icm-dir", ".genomeData"]
    else:
        ref = built_binary("oracle", "_ref", "glimmer-mg")
        sel, own = ["-m", os.path.join(DATA, "NC_000915.icm")], []
    dev = built_binary("integration", "_build", "glimmer-mg_gpu")
    env = dict(os.environ, GMG_REF_ICM_DIR=".genomeData", GMG_REF_QUIET="1")
    res = subprocess.run([ref, *opts, *sel, fa, str(tmp_path / "ref")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    res = subprocess.run([dev, *own, *opts, *sel, fa, str(tmp_path / "dev")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    want = open(str(tmp_path / "ref") + ".predict", "rb").read()
    assert open(str(tmp_path / "dev") + ".predict", "rb").read() == want and want.count(b">") == 80
