#!/usr/bin/env python3
"""Where k_orf_walk_sums8 (gmg_score_orfs, the running sums) spends its cycles: runs a GMG_OW_STAMPS build (GMG_LIB_PATH) on 1M x 500 bp
and prints the share of every phase, summed over all waves.  Diagnostic only (the stamped build waits for its loads before the steps)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
gmg.init(0)
n, L = 1_000_000, 500
reads = gmg.Reads(*gmg.synth.packed_reads(n, L, 7))
gene = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
indep = gmg.Icm.indep(0.5)
orfs, off = gmg.find_orfs(reads, min_gene_len=90, allow_truncated=True)
rows = np.stack([orfs["read"].astype(np.int32), orfs["frame"], orfs["stop_position"], orfs["orf_len"]], 1).astype(np.int32)
lib = gmg.capi.lib()
lib.gmg_debug_ow_stamps.argtypes = [C.c_void_p, C.c_int]
for rep in range(3):
    assert lib.gmg_debug_ow_stamps(None, 1) == 0
    gmg.score_orfs(gene, indep, reads, rows, min_gene_len=90)
buf = np.zeros(8, np.uint64)
assert lib.gmg_debug_ow_stamps(buf.ctypes.data, 0) == 0
st = buf.astype(np.float64)
names = ["item's offsets, loop overhead", "unit's loads issued + waited for", "the eight steps", "the three wave scans", "Q values, need mask", "stores issued"]
tot = st.sum()
print("cycles over all waves: %.3e (%d ORFs)" % (tot, len(rows)))
for nm, v in zip(names, st):
    print("  %-36s %5.1f %%" % (nm, 100 * v / tot))
