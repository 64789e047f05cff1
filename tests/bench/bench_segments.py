#!/usr/bin/env python3
"""Secondary measurement: the glimmer3-style per-ORF scoring (Score_Orfs inner loop,
src/Glimmer/glimmer3.cc:1346-1347): gene + null Cumulative_Score on ORF buffers of synthetic reads.
Prints Mbases/s of ORF bases scored (both models) and the CPU oracle's rate on a sample."""
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
import ctypes as C  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L = 500
gmg.init(0)
model = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")
gene, indep = gmg.Icm.open(model), gmg.Icm.indep(0.5)
packed, off = gmg.synth.packed_reads(n_reads, L, 7)
reads = gmg.Reads(packed, off)
rng = np.random.default_rng(1)
per = 5                                           # ~4.7 ORFs per 500-bp read
ln = rng.integers(60, L + 1, size=(n_reads, per)).astype(np.uint32)
lo = (rng.random((n_reads, per)) * (L - ln + 1)).astype(np.uint32)
rows = np.empty((n_reads, per, 4), np.uint32)
rows[..., 0] = np.arange(n_reads, dtype=np.uint32)[:, None]
rows[..., 1] = lo
rows[..., 2] = ln
rows[..., 3] = np.where(rng.random((n_reads, per)) < 0.5, gmg.REVERSED, gmg.COMPLEMENTED)
segs = gmg.Segments(reads, rows.reshape(-1, 4))
lib = gmg.capi.lib()
buf = gmg.api._DeviceBuffer(segs.total_len * 8)
buf2 = gmg.api._DeviceBuffer(segs.total_len * 8)


def run():
    gmg.api._ck(lib.gmg_segment_cumscore(gene.device(), reads.h, segs.h, 1, buf.ptr, None))
    gmg.api._ck(lib.gmg_segment_cumscore(indep.device(), reads.h, segs.h, 1, buf2.ptr, None))
    gmg.api._ck(lib.gmg_synchronize(None))


run()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    run()
dt = (time.perf_counter() - t0) / reps
out = {"segments": segs.n, "orf_bases": segs.total_len, "ms": dt * 1e3,
       "mbases_per_s": segs.total_len / dt / 1e6}
# oracle on a sample
import oracle_py  # noqa: E402
orc = oracle_py.load()
og, oi = orc.read(model), orc.indep(0.5)
sample = 2000
t0 = time.perf_counter()
nb = 0
for r, lo_, ln_, orient in segs.rows[:sample]:
    b = orc.buffer(gmg.synth.unpack_ascii(packed, int(r) * L, L), int(lo_), int(ln_), int(orient))
    orc.cumulative_score(og, b, 1)
    orc.cumulative_score(oi, b, 1)
    nb += int(ln_)
out["cpu_port_mbases_per_s"] = nb / (time.perf_counter() - t0) / 1e6
print(json.dumps(out))
