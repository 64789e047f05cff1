#!/usr/bin/env python3
"""Where k_mg_err_wcount<count> (error branch, one wave per (read, strand)) spends its cycles: runs a GMG_EW_STAMPS build
(GMG_LIB_PATH) on 1M ragged ~400-bp reads with -i or -s and prints the share of every phase, summed over all waves.  Diagnostic only."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
gmg.init(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "indel"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
reads = gmg.Reads(packed, off)
gene = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
indep = gmg.Icm.indep(0.5)
lib = gmg.capi.lib()
lib.gmg_debug_ew_stamps.argtypes = [C.c_void_p, C.c_int]
kw = dict(allow_indels=True) if mode == "indel" else dict(allow_subs=True)
for rep in range(3):
    assert lib.gmg_debug_ew_stamps(None, 1) == 0
    t0 = time.perf_counter()
    gmg.mg_score_reads(gene, indep, reads, accepted_only=True, **kw)
    dt = time.perf_counter() - t0
buf = np.zeros(8, np.uint64)
assert lib.gmg_debug_ew_stamps(buf.ctypes.data, 0) == 0
st = buf.astype(np.float64)
names = ["block header (reads, ORF ranges, first loads)", "next pair's loads issued", "sums, masks, lists (waits for the loads)", "level 0 (ORFs)",
         "level 2 (the calls' own starts)", "level 0 -> 1, level 1 and its pairs", "verdicts", "waiting for the pair's own loads (asked for one pair earlier)"]
tot = st.sum()
print("%s: cycles over all waves: %.3e (whole call incl. fetch %.1f ms)" % (mode, tot, dt * 1e3))
for nm, v in zip(names, st):
    if v:
        print("  %-48s %5.1f %%" % (nm, 100 * v / tot))
