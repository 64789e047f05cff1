#!/usr/bin/env python3
"""Differential stress run of glimmer-mg's front half (default mode and the error branch): N ragged random reads, several parameter sets, EVERY ORF's start list
(push order, errors, scores), verdict and bounds against the CPU oracle.  usage: stress_mg.py [reads] [seed]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
model = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")
gene, o_gene = gmg.Icm.open(model), orc.read(model)
lens = np.clip(rng.normal(300, 150, n).round(), 0, 1500).astype(np.int64)
seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=int(L))) for L in lens]
# some homopolymer-rich reads (Set_Quality_454 makes the last base of a run a branch point)
for i in range(0, n, 7):
    s = list(seqs[i])
    for k in range(0, len(s) - 6, 17):
        s[k:k + 4] = s[k] * 4
    seqs[i] = "".join(s)
reads = gmg.Reads.from_strings(seqs)
quals = [np.where(rng.random(len(s)) < 0.1, rng.integers(0, 19, len(s)), rng.integers(19, 41, len(s))).astype(np.int32) for s in seqs]
cases = [
    ("default", dict(), dict(), None),
    ("default2", dict(min_gene_len=45, allow_truncated=False, ignore_score_len=200), dict(), None),
    ("-i", dict(), dict(allow_indels=True), None),
    ("-i -q", dict(min_gene_len=60), dict(allow_indels=True), quals),
    ("-s", dict(allow_truncated=False), dict(allow_subs=True), None),
    ("-i max1", dict(ignore_score_len=120), dict(allow_indels=True, indel_max=1, indel_suffix_score_threshold=-8.0), None),
]
gc = 0.46
indep, o_indep = gmg.Icm.indep(gc), orc.indep(gc)
for name, kw, ekw, q in cases:
    t0 = time.perf_counter()
    res = gmg.mg_score_reads(gene, indep, reads, quality=np.concatenate(q).astype(np.uint8) if q else None, **kw, **ekw)
    t_dev = time.perf_counter() - t0
    prm, ep = orc.mg_params(**kw), orc.mg_err_params(**ekw)
    if not ekw:                                         # the default mode: running-sum kernels + start scan, lists in push order
        orfs, starts, off = res
        n_orf = n_start = n_acc = 0
        t0 = time.perf_counter()
        for r, s in enumerate(seqs):
            want_orfs, scored = orc.mg_read(o_gene, o_indep, s.encode(), prm)
            mine = orfs[int(off[r]):int(off[r + 1])]
            assert np.array_equal(np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1).reshape(-1, 4), want_orfs), (name, r)
            for o, (out, want) in zip(mine, scored):
                st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
                assert [(int(a["j"]), int(a["pos"]), int(a["which"]), int(a["truncated"]), int(a["first"]), float(a["score"])) for a in st] == \
                       [(w.j, w.pos, w.which, w.truncated, w.first, w.score) for w in want], (name, r)
                assert (int(o["lo"]), int(o["hi"]), int(o["accepted"]), int(o["first_j"])) == (out.lo, out.hi, out.accepted, out.first_j), (name, r)
                n_orf += 1; n_start += len(want); n_acc += out.accepted != 0
        print("%-8s reads %d  ORFs %d  starts %d  accepted %d  device %.1f ms  oracle %.1f s  -- identical" %
              (name, n, n_orf, n_start, n_acc, t_dev * 1e3, time.perf_counter() - t0))
        continue
    orfs, starts, off, errs = res
    n_orf = n_start = n_acc = 0
    t0 = time.perf_counter()
    for r, s in enumerate(seqs):
        want_orfs, _, scored = orc.mg_read_errors(o_gene, o_indep, s.encode(), prm, ep, q[r] if q else None)
        mine = orfs[int(off[r]):int(off[r + 1])]
        assert np.array_equal(np.stack([mine["frame"], mine["stop_position"], mine["gene_len"], mine["orf_len"]], 1).reshape(-1, 4), want_orfs), (name, r)
        for o, (out, want) in zip(mine, scored):
            sl = slice(o["start_begin"], o["start_begin"] + o["n_starts"])
            st, er = starts[sl], errs[sl]
            assert len(st) == len(want), (name, r, o)
            for a, e, w in zip(st, er, want):
                assert (int(a["j"]), int(a["pos"]), int(a["which"]), int(a["truncated"]), int(a["first"]), float(a["score"]), int(e["n"]),
                        int(e["pos"][0]), int(e["type"][0]), int(e["pos"][1]), int(e["type"][1])) == \
                       (w.s.j, w.s.pos, w.s.which, w.s.truncated, w.s.first, w.s.score, w.n_errors, w.err_pos[0], w.err_type[0], w.err_pos[1], w.err_type[1]), (name, r)
            assert (int(o["lo"]), int(o["hi"]), int(o["accepted"])) == (out.lo, out.hi, out.accepted), (name, r)
            if out.accepted:
                assert float(o["best_score"]) == out.best_score
            n_orf += 1; n_start += len(want); n_acc += out.accepted != 0
    print("%-8s reads %d  ORFs %d  starts %d  accepted %d  device %.1f ms  oracle %.1f s  -- identical" %
          (name, n, n_orf, n_start, n_acc, t_dev * 1e3, time.perf_counter() - t0))
