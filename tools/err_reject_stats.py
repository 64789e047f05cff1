#!/usr/bin/env python3
"""Why the error branch rejects what it rejects: 100 k reads of ~400 bp with -i, all ORFs out; shares of the ORFs (and of the starts they
produce) that are too short whatever their score, long enough but below the score threshold, accepted.  Diagnostic (gpurun)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import _gmg_pkg
gmg = _gmg_pkg.load(); gmg.init(0)
n = 100000
lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
reads = gmg.Reads(packed, off)
nc = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
orfs, starts, first, errs = gmg.mg_score_reads(nc, gmg.Icm.indep(0.5), reads, allow_indels=True)
print("orfs", len(orfs), "starts", len(starts), "accepted", int((orfs["accepted"] != 0).sum()))
ns = orfs["n_starts"].astype(np.int64); sb = orfs["start_begin"].astype(np.int64)
has = ns > 0
print("orfs with starts", int(has.sum()))
# max j per ORF
seg = np.repeat(np.arange(len(orfs)), ns)
maxj = np.full(len(orfs), -1, np.int64)
np.maximum.at(maxj, seg, starts["j"].astype(np.int64))
short = maxj + 1 < 75
print("no start at all: %.3f" % (1 - has.mean()))
print("ORFs with starts whose longest one is still below Min_Gene_Len: %.3f of ORFs" % (short & has).mean())
rej_score = (~short) & (orfs["accepted"] == 0)
print("long enough but rejected: %.3f of ORFs, %.3f of starts" % (rej_score.mean(), ns[rej_score].sum() / ns.sum()))
acc = orfs["accepted"] != 0
print("accepted: %.3f of ORFs, %.3f of starts" % (acc.mean(), ns[acc].sum() / ns.sum()))
# how far from the read end
rd = orfs["read"].astype(np.int64); L = (off[1:] - off[:-1]).astype(np.int64)[rd]
fwd = orfs["frame"] > 0
endp = np.where(fwd, orfs["stop_position"] - 1, orfs["stop_position"] + 3).astype(np.int64)
avail = np.where(fwd, endp, L - endp + 1)
print("avail + 12 < 75: %.3f of ORFs, %.3f of starts" % ((avail + 12 < 75).mean(), ns[avail + 12 < 75].sum() / ns.sum()))
for t in (100, 150, 200):
    m = maxj + 1 < t
    print("max j + 1 < %d: %.3f" % (t, m.mean()))
