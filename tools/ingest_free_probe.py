#!/usr/bin/env python3
"""What one gmg_fasta_ingest call of 521 MB costs, and what freeing its results costs (tests/bench/bench_extras.py times both together)."""
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
lib = gmg.capi.lib()
if os.environ.get("PROBE_LIKE_EXTRAS"):                 # what bench_extras.py has done before its ingest leg
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    orc = oracle_py.load()
    MODEL = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")
    gene, indep = gmg.Icm.open(MODEL), gmg.Icm.indep(0.5)
    og, oi = orc.read(MODEL), orc.indep(0.5)
L, nr = 500, 1_000_000
rng = np.random.default_rng(1)
bases = np.frombuffer(b"acgt", np.uint8)[rng.integers(0, 4, (nr, L))]
hdr = np.frombuffer(b"".join(b">read%07d\n" % i for i in range(nr)), np.uint8).reshape(nr, -1)
body = np.full((nr, L + 8), ord("\n"), np.uint8)
body[:, np.arange(L) + np.arange(L) // 70] = bases
data = np.ascontiguousarray(np.concatenate([hdr, body], 1).reshape(-1))
lib.gmg_host_register(C.c_void_p(data.ctypes.data), C.c_size_t(data.size))
p = data.ctypes.data_as(C.c_char_p)
for rep in range(6):
    r, ix = C.c_void_p(), C.c_void_p()
    t0 = time.perf_counter()
    assert lib.gmg_fasta_ingest(p, C.c_uint64(data.size), C.byref(r), C.byref(ix)) == 0
    t1 = time.perf_counter()
    lib.gmg_fasta_free(ix)
    t2 = time.perf_counter()
    lib.gmg_reads_free(r)
    t3 = time.perf_counter()
    print("ingest %.2f  gmg_fasta_free %.3f  gmg_reads_free %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
