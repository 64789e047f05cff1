#!/usr/bin/env python3
"""k_mg_err_wave / k_mg_err_wcount against the level kernels on the read set of tests/test_gpu_mg_err.py::test_error_branch_every_orf_vs_oracle"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gmg_pkg
gmg = _gmg_pkg.load(); gmg.init(0)
DATA = os.path.join(ROOT, "tests", "golden", "data")
nc = gmg.Icm.open(os.path.join(DATA, "NC_000915.icm"))
rng = np.random.default_rng(99)
lengths = [0, 1, 5, 14, 15, 16, 17, 18, 33, 74, 75, 76, 99, 150, 231, 300, 301, 302, 400, 523, 700]
seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]
seqs.append("acg" * 120)
seqs.append("a" * 40 + "".join("acgt"[c] for c in rng.integers(0, 4, size=200)) + "tttttttt" + "gggg" * 9)
seqs.append("".join("acgt"[c] for c in rng.integers(0, 4, size=1300)))
seqs.append("".join("acgt"[c] for c in rng.integers(0, 4, size=2100)))
reads = gmg.Reads.from_strings(seqs)
KW = [dict(), dict(allow_truncated=False, min_gene_len=60), dict(ignore_score_len=150, start_codons=("atg", "rtg"))]
EKW = [dict(allow_indels=True), dict(allow_indels=True, indel_max=1, indel_quality_threshold=21, indel_suffix_score_threshold=-6.0), dict(allow_subs=True)]
indep = gmg.Icm.indep(0.5)
def run(kw, ekw, **opts):
    olds = {k: gmg.get_option(k) for k in opts}
    for k, v in opts.items(): gmg.set_option(k, v)
    try:
        return gmg.mg_score_reads(nc, indep, reads, **kw, **ekw)
    finally:
        for k, v in olds.items(): gmg.set_option(k, v)
wave = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for ki, kw in enumerate(KW):
    for ei, ekw in enumerate(EKW):
        a = run(kw, ekw, mg_err_wave=0, mg_err_tile=0)
        b = run(kw, ekw, mg_err_wave=wave, mg_err_tile=0)
        oa, ob = a[0], b[0]
        bad = [i for i in range(len(oa)) if oa[i].tobytes() != ob[i].tobytes()]
        print("kw%d ekw%d: orfs %d, differing records %d, starts %d vs %d" % (ki, ei, len(oa), len(bad), len(a[1]), len(b[1])))
        for i in bad[:6]:
            print("   ORF", i, "read", oa[i]["read"], "len", len(seqs[int(oa[i]["read"])]), "frame", oa[i]["frame"], "stop", oa[i]["stop_position"],
                  {f: (oa[i][f], ob[i][f]) for f in oa.dtype.names if oa[i][f] != ob[i][f]})
