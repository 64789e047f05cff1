#!/usr/bin/env python3
"""Differential stress run of gmg_score_orfs (glimmer3's Score_Orfs inner loop, src/Glimmer/glimmer3.cc:1275-1552): random batch
shapes (ragged reads 0 .. 1,500 bp, uniform batches, tiny reads), the ORFs Find_Orfs gives for random codon sets / Min_Gene_Len /
truncation, random Ignore_Score_Len, threshold, first-start rule and start codon subsets -- the events path (running sums per class,
one lane per ORF at its start codons), the fused path (one lane walks its ORF) and the any-model path (two cumulative-score launches)
must return the same bytes, and a sample of the ORFs of every configuration must equal the CPU oracle's start lists.
usage: stress_orfs.py [configurations] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gmg_pkg  # noqa: E402
import oracle_py  # noqa: E402

gmg = _gmg_pkg.load()
gmg.init(0)
orc = oracle_py.load()
n_conf = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DATA = os.path.join(ROOT, "tests", "golden", "data")
MODELS = ["NC_000915.icm", "seqs.cluster-0.run1.filt.gicm", "seqs.cluster-4.run1.filt.gicm"]
t0 = time.perf_counter()
tot_orfs = tot_starts = checked = 0
for conf in range(n_conf):
    shape = rng.integers(0, 4)
    n = int(rng.integers(200, 4000))
    if shape == 0:
        lens = np.full(n, int(rng.integers(60, 900)))
    elif shape == 1:
        lens = rng.integers(0, 1500, n)
    elif shape == 2:
        lens = rng.integers(0, 120, n)
    else:
        lens = np.clip(rng.normal(400, 150, n).round(), 0, 1500).astype(np.int64)
    seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=int(L))) for L in lens]
    reads = gmg.Reads.from_strings(seqs)
    mfile = MODELS[int(rng.integers(0, len(MODELS)))]
    gene, gc = gmg.Icm.open(os.path.join(DATA, mfile)), float(rng.uniform(0.3, 0.7))
    indep = gmg.Icm.indep(gc)
    starts_all = ("atg", "gtg", "ttg", "ctg")
    start_codons = tuple(sorted(rng.choice(starts_all, size=int(rng.integers(1, 5)), replace=False)))
    stop_codons = (("taa", "tag", "tga"), ("taa", "tag"), ("tga",))[int(rng.integers(0, 3))]
    find_kw = dict(min_gene_len=int(rng.choice([30, 60, 75, 90, 150])), allow_truncated=bool(rng.integers(0, 2)),
                   start_codons=start_codons, stop_codons=stop_codons)
    orfs, _ = gmg.find_orfs(reads, **find_kw)
    rows = np.stack([orfs["read"], orfs["frame"], orfs["stop_position"], orfs["orf_len"]], 1).astype(np.int64) if len(orfs) else np.zeros((0, 4), np.int64)
    kw = dict(min_gene_len=find_kw["min_gene_len"], allow_truncated=find_kw["allow_truncated"], use_first_start=bool(rng.integers(0, 2)),
              ignore_score_len=int(rng.choice([2 ** 31 - 1, 300, 120])), start_threshold=float(rng.choice([-6.0, -2.0, 0.0, -30.0])),
              start_codons=start_codons)
    out = {}
    for path in (0, 2, 1):
        with gmg.option("orfs_exact_path", path):
            out[path] = gmg.score_orfs(gene, indep, reads, rows, **kw)
    total = int(out[0][0]["start_begin"][-1]) + int(out[0][0]["n_starts"][-1]) if len(rows) else 0
    for path in (2, 1):
        assert out[0][0].tobytes() == out[path][0].tobytes(), (conf, path, "results")
        assert out[0][1][:total].tobytes() == out[path][1][:total].tobytes(), (conf, path, "starts")
    # a sample of the ORFs against the oracle (its own buffers, its own cumulative scores)
    o_gene, o_indep, o_prm = orc.read(os.path.join(DATA, mfile)), orc.indep(gc), orc.orf_params(**kw)
    for i in rng.choice(len(rows), size=min(40, len(rows)), replace=False) if len(rows) else []:
        r, frame, stop_pos, orf_len = (int(x) for x in rows[i])
        cnt, o_out, want = orc.score_orf(o_gene, o_indep, seqs[r], frame, stop_pos, orf_len, o_prm)
        got = out[0][0][i]
        assert (int(got["first_j"]), int(got["best_j"]), int(got["best_pos"]), int(got["orf_is_truncated"])) == \
               (o_out.first_j, o_out.best_j, o_out.best_pos, o_out.orf_is_truncated), (conf, i)
        assert float(got["best_score"]) == o_out.best_score, (conf, i)
        if cnt < 0:
            assert got["n_starts"] == 0 and not got["is_tentative_gene"], (conf, i)
        else:
            st = out[0][1][int(got["start_begin"]):int(got["start_begin"]) + cnt]
            assert int(got["n_starts"]) == cnt and bool(got["is_tentative_gene"]) == bool(o_out.is_tentative_gene), (conf, i)
            assert [(int(a["j"]), int(a["pos"]), int(a["which"]), int(a["truncated"]), int(a["first"]), float(a["score"])) for a in st] == \
                   [(w.j, w.pos, w.which, w.truncated, w.first, w.score) for w in want], (conf, i)
        checked += 1
    tot_orfs += len(rows)
    tot_starts += total
    if (conf + 1) % 10 == 0:
        print("%d configurations, %d ORFs, %d starts: three paths identical, %d ORFs equal to the oracle (%.0f s)" %
              (conf + 1, tot_orfs, tot_starts, checked, time.perf_counter() - t0), flush=True)
print("all %d configurations identical: %d ORFs, %d starts on three paths; %d sampled ORFs equal to the oracle" % (n_conf, tot_orfs, tot_starts, checked))
