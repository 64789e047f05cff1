#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out/r05p10; mkdir -p $R; rm -f $R/*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 - > $R/nstarts_hist.txt 2>&1 <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import _gmg_pkg
gmg = _gmg_pkg.load(); gmg.init(0)
n = 1000000
lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
reads = gmg.Reads(packed, off)
gene = gmg.Icm.open(os.path.join("tests", "golden", "data", "NC_000915.icm")); indep = gmg.Icm.indep(0.5)
for kw in (dict(allow_indels=True), dict(allow_subs=True)):
    res = gmg.mg_score_reads(gene, indep, reads, accepted_only=True, **kw)
    ns = res[0]["n_starts"]
    print(kw, "ORFs", len(ns), "starts", int(ns.sum()), "max", int(ns.max()))
    for lo, hi in ((1, 8), (9, 16), (17, 32), (33, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 4096), (4097, 1 << 30)):
        m = (ns >= lo) & (ns <= hi)
        print("  %5d - %-6d ORFs %7d  starts %8d" % (lo, hi, int(m.sum()), int(ns[m].sum())))
P
cat $R/nstarts_hist.txt
echo done
