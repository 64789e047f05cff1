#!/usr/bin/env python3
"""Where k_mg_tile_starts (glimmer-mg front half, fused running sums + start lists) spends its cycles: runs a GMG_MT_STAMPS
build (GMG_LIB_PATH) on 1M x 500 bp and prints the share of every stage, summed over all waves.  Diagnostic only."""
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
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 500
if len(sys.argv) > 3 and sys.argv[3] == "ragged":         # the lengths of tests/bench/bench_mg.py's ragged batch
    lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
else:
    packed, off = gmg.synth.packed_reads(n, L, 7)
reads = gmg.Reads(packed, off)
gene = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
indep = gmg.Icm.indep(0.5)
lib = gmg.capi.lib()
lib.gmg_debug_mt_stamps.argtypes = [C.c_void_p, C.c_int]
for rep in range(3):
    assert lib.gmg_debug_mt_stamps(None, 1) == 0
    t0 = time.perf_counter()
    gmg.mg_score_reads_counts(gene, indep, reads) if hasattr(gmg, "mg_score_reads_counts") else gmg.mg_score_reads(gene, indep, reads)
    dt = time.perf_counter() - t0
buf = np.zeros(8, np.uint64)
assert lib.gmg_debug_mt_stamps(buf.ctypes.data, 0) == 0
st = buf.astype(np.float64)
names = ["stage 0 (commit, waits for loads)", "next tile's loads issued", "stage 1 (flags, ORFs)", "stage 2 (scan)", "stages 3, 4 (starts, ORF records)",
         "", "barrier at the top"]
tot = st.sum()
print("cycles over all waves: %.3e (whole call incl. fetch %.1f ms)" % (tot, dt * 1e3))
for i, nm in enumerate(names):
    if nm:
        print("  %-36s %6.1f %%" % (nm, 100 * st[i] / tot))
