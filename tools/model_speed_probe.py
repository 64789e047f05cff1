#!/usr/bin/env python3
"""Does a call's time depend on WHICH table it gets?  One model at a time: a sample file, a table trained here on a 26-kb slice
(tests/models64.py), the same file with its bases renamed -- gmg_mg_score_reads (default and -i) and gmg_score_reads_strings."""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gmg_pkg  # noqa: E402
import models64  # noqa: E402

gmg = _gmg_pkg.load()
gmg.init(0)
api, capi = gmg.api, gmg.capi
lib = capi.lib()
n = 1_000_000
lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
reads = gmg.Reads(packed, off)
indep = gmg.Icm.indep(0.5)
t = tempfile.mkdtemp()
g3 = models64.gene_models(gmg, t, 8)
p1 = models64.period1_models(gmg, t, 9)
rl = models64.relabeled_models(gmg, t, models64.GENE_FILES, 8)


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def mg(model, flags):
    prm = capi.MgParams(75, 1, 2**31 - 1, 3, 3, flags, -6.0)
    prm.min_indel_orf_len, prm.indel_quality_threshold, prm.indel_max, prm.indel_suffix_score_threshold = 15, 18, 2, -12.0
    for i, c in enumerate(("atg", "gtg", "ttg")):
        prm.start_codon[i].value = c.encode()
    for i, c in enumerate(("taa", "tag", "tga")):
        prm.stop_codon[i].value = c.encode()

    def call():
        res = C.c_void_p()
        api._ck(lib.gmg_mg_score_reads(model.device(), indep.device(), reads.h, C.byref(prm), None, C.byref(res), None))
        lib.gmg_mg_result_free(res)
    return timed(call)


for name, (m, path) in [("file NC_000915", g3[0]), ("file cluster-4 gicm", g3[3]), ("trained g3_00", g3[5]), ("trained g3_02", g3[7]), ("relabeled", rl[6])]:
    print("%-22s mg %7.2f ms   mg -i (accepted only) %8.2f ms" % (name, mg(m, 0), mg(m, 1 | 2)), flush=True)
out = api._DeviceBuffer(n * 2 * 8)
for name, (m, path) in [("file cluster-0", p1[0]), ("trained p1_00", p1[6]), ("trained p1_02", p1[8])]:
    arr = (C.c_void_p * 1)(m.device())
    print("%-22s strings %7.3f ms per model" % (name, timed(lambda: api._ck(lib.gmg_score_reads_strings(arr, 1, reads.h, out.ptr, None)))), flush=True)
if os.environ.get("PROBE_TIMING"):
    gmg.set_option("mg_timing", 1)
    for name, (m, path) in [("file NC_000915", g3[0]), ("trained g3_00", g3[5])]:
        print("==", name, flush=True)
        mg(m, 0)
    gmg.set_option("mg_timing", 0)
    for name, (m, path) in [("file NC_000915", g3[0]), ("trained g3_00", g3[5])]:
        res = gmg.mg_score_reads(m, indep, gmg.Reads(packed[:int(off[20000]) // 16 + 2], off[:20001].copy()))
        print(name, "ORFs", len(res[0]), "starts", len(res[1]), "accepted", int((res[0]["accepted"] != 0).sum()))
