#!/usr/bin/env python3
"""Where k_mg_err_tile (glimmer-mg's error branch tile by tile) spends its cycles: runs a GMG_ET_STAMPS build (GMG_LIB_PATH) on
1M ragged reads with -i (or -s) and prints the share of every stage, wave 0 of every work-group.  Diagnostic only."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "indel"
lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
reads = gmg.Reads(packed, off)
gene = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
indep = gmg.Icm.indep(0.5)
lib = gmg.capi.lib()
lib.gmg_debug_et_stamps.argtypes = [C.c_void_p, C.c_int]
kw = dict(allow_indels=True) if mode == "indel" else dict(allow_subs=True)
for rep in range(3):
    assert lib.gmg_debug_et_stamps(None, 1) == 0
    t0 = time.perf_counter()
    gmg.mg_score_reads(gene, indep, reads, accepted_only=True, **kw)
    dt = time.perf_counter() - t0
buf = np.zeros(16, np.uint64)
assert lib.gmg_debug_et_stamps(buf.ctypes.data, 0) == 0
st = buf.astype(np.float64)
names = {0: "top of the item + stage A", 1: "stage B (sums, flags), own work", 2: "... waiting for the other waves", 3: "stage C (event lists)",
         4: "barrier + next item's loads issued", 5: "ORF records staged", 6: "event counts fetched + scanned", 7: "waiting behind the items",
         8: "items, level 0", 9: "items, level 1", 10: "items, level 2", 11: "verdict, staging", 12: "end-of-item barrier"}
tot = st.sum()
print("cycles, wave 0 of every work-group: %.3e (whole call incl. fetch %.1f ms)" % (tot, dt * 1e3))
for i, nm in names.items():
    print("  %-40s %6.1f %%" % (nm, 100 * st[i] / tot))
