#!/usr/bin/env python3
"""Secondary measurement: gmg_mg_score_reads (glimmer-mg's front half: Score_All_Frames + Find_Orfs +
Score_Orf_Starts + the filter of Score_Orfs_Errors) on synthetic 500-bp reads, with the reads
resident in HBM.  Prints one JSON line: Mbases/s of read bases through the whole front half, the sizes of what
leaves the GPU, and the CPU oracle's rate for the same steps on a sample.
BENCH_ERR=indel|sub: the error branch (glimmer-mg -i / -s; BASELINE configs[4] with the ragged reads and
Set_Quality_454's homopolymer qualities), accepted ORFs only."""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = int(os.environ.get("BENCH_READ_LEN", "500"))
gmg.init(0)
model = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")
gene, indep = gmg.Icm.open(model), gmg.Icm.indep(0.5)
ragged = len(sys.argv) > 3 and sys.argv[3] == "ragged"
if ragged:                                              # BASELINE configs[4] shape: lengths ~ N(400, 60^2), clipped
    lens = np.clip(np.random.default_rng(12).normal(400, 60, n_reads).round(), 100, 700).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
else:
    packed, off = gmg.synth.packed_reads(n_reads, L, 7)
reads = gmg.Reads(packed, off)
lib = capi.lib()
err = os.environ.get("BENCH_ERR", "")
prm = capi.MgParams(75, 1, 2**31 - 1, 3, 3, {"": 0, "indel": 1 | 2, "sub": 1 | 4}[err], -6.0)
prm.min_indel_orf_len, prm.indel_quality_threshold, prm.indel_max, prm.indel_suffix_score_threshold = 15, 18, 2, -12.0
for i, c in enumerate(("atg", "gtg", "ttg")):
    prm.start_codon[i].value = c.encode()
for i, c in enumerate(("taa", "tag", "tga")):
    prm.stop_codon[i].value = c.encode()
n_nulls = int(os.environ.get("BENCH_NULLS", "0"))      # glimmer-mg -c: a null model per read, BENCH_NULLS GC values
if n_nulls:
    null_set = gmg.NullSet([gmg.Icm.indep(float(gc)) for gc in np.linspace(0.3, 0.7, n_nulls)])
    read_null = np.random.default_rng(3).integers(0, n_nulls, n_reads).astype(np.uint32)
    prm.nulls, prm.read_null = null_set.h, read_null.ctypes.data
    indep = null_set.icms[0]
fs = api._DeviceBuffer(6 * reads.total_bases * 8)      # the Frame_Scores table stays on the device
fs_ptr = None if os.environ.get("BENCH_OWN_TABLE") or n_nulls else fs.ptr    # (or let the call use its own, row-padded table)


def run():
    res = C.c_void_p()
    api._ck(lib.gmg_mg_score_reads(gene.device(), indep.device(), reads.h, C.byref(prm), fs_ptr, C.byref(res), None))
    n_orfs, n_starts = C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_mg_result_info(res, C.byref(n_orfs), C.byref(n_starts)))
    lib.gmg_mg_result_free(res)
    return n_orfs.value, n_starts.value


n_orfs, n_starts = run()
times = []
for _ in range(reps):
    t0 = time.perf_counter()
    run()
    times.append(time.perf_counter() - t0)
dt = sorted(times)[len(times) // 2]
out = {"reads": n_reads, "read_bases": reads.total_bases, "orfs": n_orfs, "starts": n_starts,
       "ms": dt * 1e3, "ms_all": [round(t * 1e3, 2) for t in times], "mbases_per_s": reads.total_bases / dt / 1e6,
       "result_bytes": n_orfs * 56 + n_starts * 24, "frame_scores_bytes": 6 * reads.total_bases * 8}
# the oracle on a sample: the same steps on one core
import oracle_py  # noqa: E402
orc = oracle_py.load()
og, oi = orc.read(model), orc.indep(0.5)
oprm = orc.mg_params()
oerr = orc.mg_err_params(allow_indels=err == "indel", allow_subs=err == "sub")
sample = 200 if err == "indel" else 2000
t0 = time.perf_counter()
for r in range(sample):
    seq = gmg.synth.unpack_ascii(packed, int(off[r]), int(off[r + 1] - off[r]))
    if err:
        orc.mg_read_errors(og, oi, seq, oprm, oerr)
    else:
        orc.mg_read(og, oi, seq, oprm)
out["cpu_port_mbases_per_s"] = int(off[sample]) / (time.perf_counter() - t0) / 1e6
out["ragged"] = ragged
out["error_branch"] = err or None
out["cpu_sample_reads"] = sample
print(json.dumps(out))
