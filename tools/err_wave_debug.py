#!/usr/bin/env python3
"""k_mg_err_wave against the level kernels on the 80-read golden case: the first differing ORF records / start lists, field by field"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gmg_pkg
gmg = _gmg_pkg.load(); gmg.init(0)
DATA = os.path.join(ROOT, "tests", "golden", "data")
nc = gmg.Icm.open(os.path.join(DATA, "NC_000915.icm"))
hdrs, seqs = gmg.read_fasta(os.path.join(DATA, "seqs80.fa"))
mode = sys.argv[1] if len(sys.argv) > 1 else "indel"
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 80
seqs = [s.decode() if isinstance(s, bytes) else s for s in seqs][:nreads]
reads = gmg.Reads.from_strings([s.lower() for s in seqs])
kw = dict(allow_indels=True) if mode == "indel" else dict(allow_subs=True)
indep = gmg.Icm.indep(0.39)
def run(**opts):
    olds = {k: gmg.get_option(k) for k in opts}
    for k, v in opts.items(): gmg.set_option(k, v)
    try:
        return gmg.mg_score_reads(nc, indep, reads, **kw)
    finally:
        for k, v in olds.items(): gmg.set_option(k, v)
a = run(mg_err_wave=0, mg_err_tile=0)
b = run(mg_err_wave=1, mg_err_tile=0)
oa, ob = a[0], b[0]
print("orfs", len(oa), len(ob), "starts", len(a[1]), len(b[1]))
nd = 0
for i in range(min(len(oa), len(ob))):
    if oa[i].tobytes() != ob[i].tobytes():
        print("ORF", i, "read", oa[i]["read"], "len", len(seqs[int(oa[i]["read"])]))
        for f in oa.dtype.names:
            if oa[i][f] != ob[i][f]: print("   ", f, oa[i][f], ob[i][f])
        print("    frame", oa[i]["frame"], "stop", oa[i]["stop_position"], "lo/hi", oa[i]["lo"], oa[i]["hi"], "n_starts", oa[i]["n_starts"], ob[i]["n_starts"])
        nd += 1
        if nd >= 12: break
print("differing ORF records shown:", nd)
if len(a[1]) == len(b[1]):
    sa, sb, ea, eb = a[1], b[1], a[3], b[3]
    bad = [k for k in range(len(sa)) if sa[k].tobytes() != sb[k].tobytes() or ea[k].tobytes() != eb[k].tobytes()]
    print("differing starts:", len(bad), "of", len(sa))
    for k in bad[:12]:
        print("  start", k, sa[k], sb[k], ea[k], eb[k])
