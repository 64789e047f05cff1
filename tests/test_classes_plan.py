"""glimmer-mg -c bookkeeping of the product (gmg_classes_*, host only -- no GPU needed) and of the oracle (orc_classes_*)
against what the REAL reference did (tests/golden/classes_*.npz, written by oracle/gen_golden_classes.py from the reference's
own main() running over the synthetic .genomeData tree of tests/golden/make_genome_data.py):
the order in which reads are processed (ICM by ICM, hash-table order, chunk by chunk), every read's ICM file, Indep_GC_Frac,
Genbank_Xlate_Code, stop codons and Ignore_Score_Len."""
import os
import sys

import numpy as np
import pytest

from conftest import DATA, GOLD

sys.path.insert(0, GOLD)
import make_genome_data  # noqa: E402


@pytest.fixture(scope="module")
def genome_data(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("genomeData"))
    mixed = make_genome_data.write_class_variants(d)
    assert open(mixed).read() == open(os.path.join(DATA, "mixed.class.txt")).read()      # the committed copy is what the script writes
    make_genome_data.build(os.path.join(d, ".genomeData"), [os.path.join(DATA, "seqs.class.txt"), mixed])
    return d


@pytest.fixture(autouse=True)
def in_genome_data_dir(genome_data, monkeypatch):
    """the order of the ICM files depends on the hash of their whole NAME, ICM_dir included: the goldens were made with
    ICM_dir = ".genomeData" relative to the working directory"""
    monkeypatch.chdir(genome_data)


def chunks(n, size):
    return [(b, min(n, b + size)) for b in range(0, n, size)] if size else [(0, n)]


CASES = ["default", "indel", "g90", "mixed_chunks", "mixed_sub"]


@pytest.mark.parametrize("case", CASES)
def test_product_plan_is_the_reference_order(gmg, seqs_fa, genome_data, case):
    g = np.load(os.path.join(GOLD, "classes_%s.npz" % case))
    hdrs = seqs_fa[0]
    cls = gmg.api.Classes(open(os.path.join(DATA, str(g["class_file"]))).read(), ".genomeData")
    assert cls.n_icms >= len(g["icm_files"])                                    # ICM files of reads that are in no input never show
    files = [os.path.relpath(cls.icm_file(k), ".genomeData") for k in range(cls.n_icms)]
    got_reads, got_icm, got_gc, got_tt = [], [], [], []
    for b, e in chunks(len(hdrs), int(g["chunk"])):
        order, icm_begin, gc, transl = cls.plan(hdrs[b:e])
        assert icm_begin[0] == 0 and icm_begin[-1] == len(order) and np.all(np.diff(icm_begin.astype(np.int64)) >= 0)
        for f in range(cls.n_icms):
            for k in range(int(icm_begin[f]), int(icm_begin[f + 1])):
                got_reads.append(hdrs[b + int(order[k])].split()[0])
                got_icm.append(files[f])
        got_gc += gc.tolist()
        got_tt += transl.tolist()
    assert got_reads == g["reads"].tolist()                                     # the order <tag>.predict is written in
    assert got_icm == [str(g["icm_files"][i]) for i in g["icm"]]
    assert np.array_equal(np.array(got_gc), g["gc"])                            # bit for bit
    assert got_tt == g["transl"].tolist()
    # the stop codons and Ignore_Score_Len every read was scored with
    for k in range(len(got_reads)):
        stops = gmg.api.stop_codons_by_code(got_tt[k])
        assert ",".join(stops) == str(g["stops"][k])
        assert gmg.api.ignore_score_len(got_gc[k], stops) == int(g["isl"][k])
    assert cls.n_missing_gc >= 12 and len(set(got_gc)) > 400 and len(set(got_tt)) == 2


def test_classes_load_refuses_what_the_reference_would_crash_on(gmg, genome_data):
    for bad in ("read1\n", "read1 A|B\n\nread2 A|B\n", "read1 NoBarHere\n", "read1 A|B NoBar\n"):
        with pytest.raises(gmg.GmgError):
            gmg.api.Classes(bad, ".genomeData")
    c = gmg.api.Classes("read1 A|B", ".genomeData")                              # no final newline; unknown class: default GC and table
    order, icm_begin, gc, transl = c.plan(["read0 x", "read1 some text", "", "   ", "read1"])
    assert order.tolist() == [4] and gc.tolist() == [0.5] and transl.tolist() == [11]      # the later read of a key wins
    assert c.icm_file(0) == ".genomeData/A/B.gicm"
    with pytest.raises(gmg.GmgError):
        gmg.api.stop_codons_by_code(7)
    assert gmg.api.stop_codons_by_code(2) == ("taa", "tag", "aga", "agg")


@pytest.mark.parametrize("case", CASES)
def test_oracle_plan_is_the_reference_order(oracle, seqs_fa, genome_data, case):
    """the oracle's restatement (its own SGI hash table in C) against the same reference dump"""
    g = np.load(os.path.join(GOLD, "classes_%s.npz" % case))
    hdrs = seqs_fa[0]
    c = oracle.classes_load(open(os.path.join(DATA, str(g["class_file"]))).read(), ".genomeData")
    files = [os.path.relpath(f, ".genomeData") for f in oracle.classes_icm_files(c)]
    got_reads, got_icm, got_gc, got_tt = [], [], [], []
    for b, e in chunks(len(hdrs), int(g["chunk"])):
        order, icm_begin, gc, transl = oracle.classes_plan(c, hdrs[b:e])
        for f in range(len(files)):
            for k in range(int(icm_begin[f]), int(icm_begin[f + 1])):
                got_reads.append(hdrs[b + int(order[k])].split()[0])
                got_icm.append(files[f])
        got_gc += gc.tolist()
        got_tt += transl.tolist()
    assert got_reads == g["reads"].tolist()
    assert got_icm == [str(g["icm_files"][i]) for i in g["icm"]]
    assert np.array_equal(np.array(got_gc), g["gc"])
    assert got_tt == g["transl"].tolist()
    for k in range(len(got_reads)):
        stops = oracle.stop_codons_by_code(got_tt[k])
        assert ",".join(stops) == str(g["stops"][k])
        assert oracle.ignore_score_len(got_gc[k], stops) == int(g["isl"][k])
    oracle.L.orc_classes_free(c)


def test_oracle_and_product_agree_on_a_table_that_rehashes(gmg, oracle, genome_data):
    """5,000 classified reads over 700 ICM files: both hash tables grow through several bucket counts (193 -> 389 -> ... -> 6151)"""
    rng = np.random.default_rng(3)
    names = ["r%d_%s" % (i, "".join(rng.choice(list("abcXYZ09_-"), 6))) for i in range(5000)]
    lines = ["%s S%d|N%d S%d|N%d" % (n, rng.integers(0, 700), rng.integers(0, 3), rng.integers(0, 50), rng.integers(0, 2)) for n in names]
    text = "\n".join(lines) + "\n"
    hdrs = [n + " desc" for n in names]
    rng.shuffle(hdrs)
    c = oracle.classes_load(text, ".genomeData")
    p = gmg.api.Classes(text, ".genomeData")
    assert oracle.classes_icm_files(c) == [p.icm_file(k) for k in range(p.n_icms)]
    for part in (hdrs, hdrs[:1234], hdrs[1234:]):
        a, b = oracle.classes_plan(c, part), p.plan(part)
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x, np.float64), np.asarray(y, np.float64))
    assert oracle.classes_load("read1\n", ".genomeData") is None
    oracle.L.orc_classes_free(c)
