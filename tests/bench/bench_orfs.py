#!/usr/bin/env python3
"""Secondary measurement: gmg_score_orfs (the scoring part of Score_Orfs, src/Glimmer/glimmer3.cc:1275-1552)
on synthetic 500-bp reads with ~5 ORFs each, timed with the ORFs already uploaded, for both device paths:
  fused  gene-only six-frame pass + one-lane-per-ORF scan   (default for the 12/7/3 model)
  exact  two cumulative-score launches over the ORF buffers + scan   (GMG_ORFS_EXACT_PATH=1, any model)
and checks that the two return the same bytes.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
api, capi = gmg.api, gmg.capi

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L = 500
gmg.init(0)
model = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")
gene, indep = gmg.Icm.open(model), gmg.Icm.indep(0.5)
packed, off = gmg.synth.packed_reads(n_reads, L, 7)
reads = gmg.Reads(packed, off)
rng = np.random.default_rng(1)
per = 5                                           # ~4.7 ORFs per 500-bp read
ln = (rng.integers(30, L // 3, size=(n_reads, per)) * 3).astype(np.int64)          # 90 .. 498, whole codons
lo = (rng.random((n_reads, per)) * (L - ln + 1)).astype(np.int64)
fwd = rng.random((n_reads, per)) < 0.5
o = np.zeros(n_reads * per, api.ORF_DTYPE)
o["read"] = np.repeat(np.arange(n_reads, dtype=np.uint32), per)
o["frame"] = np.where(fwd, 1 + lo % 3, -(1 + lo % 3)).reshape(-1)
o["stop_position"] = np.where(fwd, lo + ln + 1, lo - 2).reshape(-1)               # glimmer3.cc:1322-1343
o["orf_len"] = ln.reshape(-1)
lib = capi.lib()
prm = capi.OrfParams(90, 1, 0, 2**31 - 1, -6.0, 3)
for i, c in enumerate(("atg", "gtg", "ttg")):
    prm.start_codon[i].value = c.encode()
batch, max_starts = C.c_void_p(), C.c_uint64()
api._ck(lib.gmg_orfs_upload(reads.h, api._ptr(o), len(o), C.byref(max_starts), C.byref(batch)))


def run(exact):
    api.set_option("orfs_exact_path", 1 if exact else 0)
    res = np.zeros(len(o), api.ORF_RESULT_DTYPE)
    starts = np.zeros(max(int(max_starts.value), 1), api.START_DTYPE)
    call = lambda: api._ck(lib.gmg_score_orfs(gene.device(), indep.device(), reads.h, batch, C.byref(prm),
                                              api._ptr(res), api._ptr(starts), None))
    call()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    return (time.perf_counter() - t0) / reps, res, starts


t_f, res_f, st_f = run(False)
t_e, res_e, st_e = run(True)
same = res_f.tobytes() == res_e.tobytes()
for r in res_f[res_f["n_starts"] > 0][:20000]:
    b, n = int(r["start_begin"]), int(r["n_starts"])
    same = same and st_f[b:b + n].tobytes() == st_e[b:b + n].tobytes()
lib.gmg_orf_batch_free(batch)
orf_bases = int(ln.sum())
print(json.dumps({"reads": n_reads, "orfs": len(o), "orf_bases": orf_bases, "paths_identical": bool(same),
                  "fused_ms": t_f * 1e3, "fused_morf_bases_per_s": orf_bases / t_f / 1e6,
                  "exact_ms": t_e * 1e3, "exact_morf_bases_per_s": orf_bases / t_e / 1e6,
                  "note": "times include the D2H copy of results + start lists"}))
