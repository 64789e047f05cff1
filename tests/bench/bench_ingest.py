#!/usr/bin/env python3
"""Secondary measurement: FASTA bytes on the host -> packed reads in HBM (gmg_fasta_ingest), and the
whole glimmer-mg front half from file bytes to start lists on the host (ingest + gmg_mg_score_reads + fetch), PCIe
included.  Synthetic file: 1M reads x 500 bp, 70 bases per line.  The CPU figure is the oracle's Fasta_Read + filter
loop on a sample (one core), i.e. what the reference's fgetc loop does."""
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
L, width = 500, 70
gmg.init(0)

# the file, built with numpy: ">read%07d\n" + 500 bases in lines of 70
rng = np.random.default_rng(3)
bases = np.frombuffer(b"acgt", np.uint8)[rng.integers(0, 4, size=(n_reads, L), dtype=np.uint8)]
n_lines = (L + width - 1) // width
rec_body = np.full((n_reads, L + n_lines), ord("\n"), np.uint8)       # every line ends with a newline
cols = np.arange(L) + np.arange(L) // width                           # base k sits behind k // width newlines
rec_body[:, cols] = bases
hdr = np.frombuffer(b"".join(b">read%07d\n" % i for i in range(n_reads)), np.uint8).reshape(n_reads, -1)
data_arr = np.ascontiguousarray(np.concatenate([hdr, rec_body], axis=1).reshape(-1))
del bases, rec_body, hdr
data = data_arr.ctypes.data_as(C.c_char_p)             # the file's bytes; page-locked below
n_data = data_arr.size

lib = capi.lib()


def ingest():
    reads, index = C.c_void_p(), C.c_void_p()
    api._ck(lib.gmg_fasta_ingest(data, n_data, C.byref(reads), C.byref(index)))
    return reads, index


def drop(reads, index):
    lib.gmg_fasta_free(index)
    lib.gmg_reads_free(reads)


drop(*ingest())
times = []
for _ in range(3):
    t0 = time.perf_counter()
    r, ix = ingest()
    times.append(time.perf_counter() - t0)
    drop(r, ix)
t_ing_pageable = sorted(times)[len(times) // 2]
api._ck(lib.gmg_host_register(data_arr.ctypes.data, n_data))
times = []
for _ in range(5):
    t0 = time.perf_counter()
    r, ix = ingest()
    times.append(time.perf_counter() - t0)
    drop(r, ix)
t_ing = sorted(times)[len(times) // 2]

# whole front half from file bytes to host-side start lists
gene = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
prm = capi.MgParams(75, 1, 2**31 - 1, 3, 3, 0, -6.0)
for i, c in enumerate(("atg", "gtg", "ttg")):
    prm.start_codon[i].value = c.encode()
for i, c in enumerate(("taa", "tag", "tga")):
    prm.stop_codon[i].value = c.encode()


out_orfs = np.empty(9_000_000 * (n_reads // 1_000_000 + 1), api.MG_ORF_DTYPE)      # reused, page-locked result buffers
out_starts = np.empty(18_000_000 * (n_reads // 1_000_000 + 1), api.START_DTYPE)
api._ck(lib.gmg_host_register(out_orfs.ctypes.data, out_orfs.nbytes))
api._ck(lib.gmg_host_register(out_starts.ctypes.data, out_starts.nbytes))


def front_half(accepted_only=False):
    prm.flags = 1 if accepted_only else 0            # GMG_MG_ACCEPTED_ONLY
    reads, index = ingest()
    n, total, gc = C.c_uint64(), C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_fasta_info(index, C.byref(n), C.byref(total), C.byref(gc)))
    indep = gmg.Icm.indep(gc.value / total.value)                     # Set_GC_Fraction + Build_Indep_WO_Stops
    res = C.c_void_p()
    api._ck(lib.gmg_mg_score_reads(gene.device(), indep.device(), reads, C.byref(prm), None, C.byref(res), None))
    n_orfs, n_starts = C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_mg_result_info(res, C.byref(n_orfs), C.byref(n_starts)))
    assert n_orfs.value <= len(out_orfs) and n_starts.value <= len(out_starts)
    first = np.empty(n.value + 1, np.uint64)
    api._ck(lib.gmg_mg_result_fetch(res, api._ptr(out_orfs), api._ptr(out_starts), api._ptr(first)))
    lib.gmg_mg_result_free(res)
    drop(reads, index)
    return n.value, total.value, n_orfs.value, n_starts.value


front_half()
times = []
for _ in range(5):
    t0 = time.perf_counter()
    stats = front_half()
    times.append(time.perf_counter() - t0)
t_all = sorted(times)[len(times) // 2]
front_half(True)
times = []
for _ in range(5):
    t0 = time.perf_counter()
    stats_acc = front_half(True)
    times.append(time.perf_counter() - t0)
t_acc = sorted(times)[len(times) // 2]

import oracle_py  # noqa: E402
orc = oracle_py.load()
sample = data_arr[:200_000_000].tobytes()
t0 = time.perf_counter()
orc.fasta_all(sample)
t_cpu = time.perf_counter() - t0
print(json.dumps({"file_bytes": n_data, "ingest_pageable_ms": t_ing_pageable * 1e3, "reads": stats[0], "bases": stats[1], "orfs": stats[2], "starts": stats[3],
                  "ingest_ms": t_ing * 1e3, "ingest_GBps": n_data / t_ing / 1e9,
                  "ingest_mbases_per_s": stats[1] / t_ing / 1e6,
                  "file_to_start_lists_ms": t_all * 1e3, "file_to_start_lists_mbases_per_s": stats[1] / t_all / 1e6,
                  "file_to_accepted_lists_ms": t_acc * 1e3, "file_to_accepted_lists_mbases_per_s": stats[1] / t_acc / 1e6,
                  "accepted_orfs": stats_acc[2], "accepted_starts": stats_acc[3],
                  "cpu_port_ingest_GBps": len(sample) / t_cpu / 1e9,
                  "note": "ingest and front half include the H2D copy of the file and the D2H copy of the results (host buffers page-locked with gmg_host_register)"}))
