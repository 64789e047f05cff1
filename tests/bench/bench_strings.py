#!/usr/bin/env python3
"""Secondary measurement (BASELINE configs[3] shape): whole-read Score_String of every read and of its reverse
complement under 64 periodicity-1 ICMs (gmg_score_reads_strings), reads resident in HBM.  The 64 models are the 6
cluster-*.icm of the reference's sample run, repeated (the tables are uploaded as 64 separate device models).
Prints one JSON line: (read, model, strand) string scores per second, base-model pairs per second, the same for the
exact segment kernel (one model) and the CPU oracle (one core, sample)."""
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
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_models = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = 500
gmg.init(0)
data = os.path.join(ROOT, "tests", "golden", "data")
if os.environ.get("BENCH_DISTINCT", "1") == "1":     # 64 DIFFERENT period-1 tables (the six sample ICMs + models trained on slices of NC_000915.fna)
    import tempfile
    import models64
    models = [m for m, _ in models64.period1_models(gmg, tempfile.mkdtemp(), n_models)]
elif os.environ.get("BENCH_DISTINCT") == "relabel":  # 64 different tables with the six files' values (bases renamed)
    import tempfile
    import models64
    models = [m for m, _ in models64.relabeled_models(gmg, tempfile.mkdtemp(), ["cluster-%d.icm" % i for i in range(6)], n_models)]
else:
    models = [gmg.Icm.open(os.path.join(data, "cluster-%d.icm" % (i % 6))) for i in range(n_models)]
ragged = len(sys.argv) > 3 and sys.argv[3] == "ragged"
if ragged:                                              # 454-like lengths ~ N(400, 60^2), clipped (as tests/bench/bench_mg.py)
    lens = np.clip(np.random.default_rng(12).normal(400, 60, n_reads).round(), 100, 700).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
else:
    packed, off = gmg.synth.packed_reads(n_reads, L, 7)
reads = gmg.Reads(packed, off)
total_bases = int(off[-1])
lib = capi.lib()
arr = (C.c_void_p * n_models)(*[m.device() for m in models])
out = api._DeviceBuffer(n_models * n_reads * 2 * 8)


def run():
    api._ck(lib.gmg_score_reads_strings(arr, n_models, reads.h, out.ptr, None))


run()
times = []
for _ in range(3):
    t0 = time.perf_counter()
    run()
    times.append(time.perf_counter() - t0)
dt = sorted(times)[1]
res = {"reads": n_reads, "models": n_models, "ms": dt * 1e3, "ms_per_model": dt * 1e3 / n_models,
       "string_scores_per_s": 2 * n_reads * n_models / dt, "gbase_model_strand_per_s": 2 * total_bases * n_models / dt / 1e9, "ragged": ragged}

# the exact segment kernel, one model, both strands
rows = np.zeros((2 * n_reads, 4), np.uint32)
rows[:, 0] = np.repeat(np.arange(n_reads, dtype=np.uint32), 2)
rows[:, 2] = np.repeat(np.diff(off).astype(np.uint32), 2)
rows[0::2, 3] = gmg.FORWARD
rows[1::2, 3] = gmg.REVCOMP
segs = gmg.Segments(reads, rows)
buf = api._DeviceBuffer(2 * n_reads * 8)
api._ck(lib.gmg_score_string(models[0].device(), reads.h, segs.h, 0, buf.ptr, None))
api._ck(lib.gmg_synchronize(None))
t0 = time.perf_counter()
api._ck(lib.gmg_score_string(models[0].device(), reads.h, segs.h, 0, buf.ptr, None))
api._ck(lib.gmg_synchronize(None))
t_seg = time.perf_counter() - t0
res["segment_kernel_ms_per_model"] = t_seg * 1e3
res["segment_kernel_gbase_model_strand_per_s"] = 2 * total_bases / t_seg / 1e9
fast = out.to_host(np.float64, 2 * n_reads)
assert np.array_equal(fast, buf.to_host(np.float64, 2 * n_reads)), "batched and segment paths differ"

import oracle_py  # noqa: E402
orc = oracle_py.load()
om = orc.read(os.path.join(data, "cluster-0.icm"))
sample = 2000
t0 = time.perf_counter()
for r in range(sample):
    s = gmg.synth.unpack_ascii(packed, int(off[r]), int(off[r + 1] - off[r]))
    orc.score_string(om, s, 0)
    orc.score_string(om, s[::-1].translate(bytes.maketrans(b"acgt", b"tgca")), 0)
res["cpu_port_gbase_model_strand_per_s"] = 2 * sample * L / (time.perf_counter() - t0) / 1e9
print(json.dumps(res))
