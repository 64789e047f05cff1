#!/usr/bin/env python3
"""Secondary measurement: gmg_score_orfs (the scoring part of Score_Orfs, src/Glimmer/glimmer3.cc:1275-1552)
on synthetic 500-bp reads with ~5 ORFs each, timed with the ORFs already uploaded, for the three device paths:
  events gene-only six-frame pass + running sums per strand and class + one lane per ORF that visits its start codons only
         (round 3; default when the models' values make every order of the additions exact)
  fused  gene-only six-frame pass + one lane per ORF walking all of its positions   (orfs_exact_path = 2)
  exact  two cumulative-score launches over the ORF buffers + scan   (orfs_exact_path = 1, any model)
and checks that all return the same bytes.  bench_orfs.py [n_reads] [reps].  Prints one JSON line."""
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


def run(path):
    api.set_option("orfs_exact_path", path)
    res = np.zeros(len(o), api.ORF_RESULT_DTYPE)
    starts = np.zeros(max(int(max_starts.value), 1), api.START_DTYPE)
    call = lambda: api._ck(lib.gmg_score_orfs(gene.device(), indep.device(), reads.h, batch, C.byref(prm),
                                              api._ptr(res), api._ptr(starts), None))
    call()
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    return (time.perf_counter() - t0) / reps, res, starts


t_v, res_v, st_v = run(0)
t_f, res_f, st_f = run(2)
big = n_reads > 400_000                           # the exact path needs 16 B per ORF base of scratch: skipped on the big batch
t_e, res_e, st_e = (float("nan"), res_f, st_f) if big else run(1)
api.set_option("orfs_exact_path", 0)
same = res_f.tobytes() == res_e.tobytes() == res_v.tobytes()
total = int(res_v["start_begin"][-1]) + int(res_v["n_starts"][-1])
same = same and st_f[:total].tobytes() == st_v[:total].tobytes() == st_e[:total].tobytes()
lib.gmg_orf_batch_free(batch)
orf_bases = int(ln.sum())
print(json.dumps({"reads": n_reads, "orfs": len(o), "orf_bases": orf_bases, "starts": total, "paths_identical": bool(same),
                  "events_ms": t_v * 1e3, "events_ms_per_million_orfs": t_v * 1e3 / (len(o) / 1e6),
                  "fused_ms": t_f * 1e3, "fused_ms_per_million_orfs": t_f * 1e3 / (len(o) / 1e6),
                  "exact_ms": t_e * 1e3, "events_morf_bases_per_s": orf_bases / t_v / 1e6,
                  "note": "times include the gene-only six-frame pass over the reads and the D2H copy of results + start lists"}))
