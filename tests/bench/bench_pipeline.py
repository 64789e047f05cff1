#!/usr/bin/env python3
"""File bytes -> start lists on the host for a file that is processed in pieces (the shape of BASELINE configs[2]:
far more reads per GPU than one batch should hold), with the stages overlapped on separate HIP streams:
  phase 1  gmg_fasta_ingest_on of every piece (stream I): the packed reads stay resident (0.25 B/base), the g/c counts add
           up to the file's GC content, from which the null model is built (Set_GC_Fraction needs the whole file first);
  phase 2  gmg_mg_score_reads of piece i (stream C) while a second host thread copies the results of piece i-1 back
           (gmg_mg_result_fetch_on, stream F) into page-locked buffers.
Prints one JSON line: serial (one stream, stage after stage) vs overlapped wall time, and checks that both give the same
bytes as scoring the whole file as ONE batch."""
import ctypes as C
import json
import os
import queue
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
api, capi = gmg.api, gmg.capi
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
piece_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
L, width = 500, 70
gmg.init(0)
lib = capi.lib()

rng = np.random.default_rng(3)
bases = np.frombuffer(b"acgt", np.uint8)[rng.integers(0, 4, size=(n_reads, L), dtype=np.uint8)]
n_lines = (L + width - 1) // width
rec_body = np.full((n_reads, L + n_lines), ord("\n"), np.uint8)
rec_body[:, np.arange(L) + np.arange(L) // width] = bases
hdr = np.frombuffer(b"".join(b">read%07d\n" % i for i in range(n_reads)), np.uint8).reshape(n_reads, -1)
data_arr = np.ascontiguousarray(np.concatenate([hdr, rec_body], axis=1).reshape(-1))
del bases, rec_body, hdr
n_data = data_arr.size
data_ptr = data_arr.ctypes.data
api._ck(lib.gmg_host_register(data_ptr, n_data))
rec_bytes = n_data // n_reads
cuts = np.zeros(n_reads // piece_reads + 8, np.uint64)
n_pieces = lib.gmg_fasta_split(C.cast(data_ptr, C.c_char_p), n_data, piece_reads * rec_bytes, api._ptr(cuts), len(cuts) - 1)
assert n_pieces >= 1
cuts = [int(c) for c in cuts[:n_pieces + 1]]

gene = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
prm = capi.MgParams(75, 1, 2**31 - 1, 3, 3, 0, -6.0)
for i, c in enumerate(("atg", "gtg", "ttg")):
    prm.start_codon[i].value = c.encode()
for i, c in enumerate(("taa", "tag", "tga")):
    prm.stop_codon[i].value = c.encode()
cap_orfs, cap_starts = 9 * piece_reads + 1024, 18 * piece_reads + 1024
bufs = []
for _ in range(2):                                      # double-buffered, page-locked result arrays
    o, s, f = np.empty(cap_orfs, api.MG_ORF_DTYPE), np.empty(cap_starts, api.START_DTYPE), np.empty(2 * piece_reads + 2, np.uint64)
    for a in (o, s, f):
        api._ck(lib.gmg_host_register(a.ctypes.data, a.nbytes))
    bufs.append((o, s, f))


def new_stream():
    s = C.c_void_p()
    api._ck(lib.gmg_stream_create(C.byref(s)))
    return s


def ingest_piece(k, stream):
    reads, index = C.c_void_p(), C.c_void_p()
    api._ck(lib.gmg_fasta_ingest_on(C.cast(data_ptr + cuts[k], C.c_char_p), cuts[k + 1] - cuts[k], C.byref(reads), C.byref(index), stream))
    n, total, gc = C.c_uint64(), C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_fasta_info(index, C.byref(n), C.byref(total), C.byref(gc)))
    lib.gmg_fasta_free(index)
    return reads, n.value, total.value, gc.value


def score_piece(reads, indep, stream):
    res = C.c_void_p()
    api._ck(lib.gmg_mg_score_reads(gene.device(), indep.device(), reads, C.byref(prm), None, C.byref(res), stream))
    return res


def fetch_piece(res, buf, stream, sink):
    n_orfs, n_starts = C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_mg_result_info(res, C.byref(n_orfs), C.byref(n_starts)))
    assert n_orfs.value <= cap_orfs and n_starts.value <= cap_starts
    api._ck(lib.gmg_mg_result_fetch_on(res, api._ptr(buf[0]), api._ptr(buf[1]), api._ptr(buf[2]), stream))
    lib.gmg_mg_result_free(res)
    if sink is not None:                                # what a consumer would do with the piece: here a checksum of the records
        sink.append((int(n_orfs.value), int(n_starts.value), int(buf[0]["accepted"][:n_orfs.value].sum()),
                     float(buf[1]["score"][:n_starts.value].sum())))


def run(overlap, sink):
    s_i, s_c, s_f = (new_stream(), new_stream(), new_stream()) if overlap else (None, None, None)
    t0 = time.perf_counter()
    pieces = [ingest_piece(k, s_i) for k in range(n_pieces)]
    gc = sum(p[3] for p in pieces) / sum(p[2] for p in pieces)
    indep = gmg.Icm.indep(gc)
    if not overlap:
        for k, p in enumerate(pieces):
            fetch_piece(score_piece(p[0], indep, None), bufs[0], None, sink)
    else:
        q = queue.Queue(maxsize=1)
        free = queue.Queue()
        for b in bufs:
            free.put(b)

        def fetcher():
            while True:
                item = q.get()
                if item is None:
                    return
                b = free.get()
                fetch_piece(item, b, s_f, sink)
                free.put(b)
        th = threading.Thread(target=fetcher)
        th.start()
        for p in pieces:
            q.put(score_piece(p[0], indep, s_c))
        q.put(None)
        th.join()
    dt = time.perf_counter() - t0
    for p in pieces:
        lib.gmg_reads_free(p[0])
    for s in (s_i, s_c, s_f):
        if s is not None:
            lib.gmg_stream_destroy(s)
    return dt, gc


run(False, None)
sink_s, sink_o = [], []
t_serial = min(run(False, sink_s if i == 0 else None)[0] for i in range(3))
t_over = min(run(True, sink_o if i == 0 else None)[0] for i in range(3))
# the whole file as ONE batch
whole = None
if n_reads <= 2_000_000:
    r_all, idx = C.c_void_p(), C.c_void_p()
    api._ck(lib.gmg_fasta_ingest(C.cast(data_ptr, C.c_char_p), n_data, C.byref(r_all), C.byref(idx)))
    n, total, gc = C.c_uint64(), C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_fasta_info(idx, C.byref(n), C.byref(total), C.byref(gc)))
    lib.gmg_fasta_free(idx)
    res = score_piece(r_all, gmg.Icm.indep(gc.value / total.value), None)
    no, ns = C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_mg_result_info(res, C.byref(no), C.byref(ns)))
    o, s, f = np.empty(no.value, api.MG_ORF_DTYPE), np.empty(ns.value, api.START_DTYPE), np.empty(n.value + 1, np.uint64)
    api._ck(lib.gmg_mg_result_fetch(res, api._ptr(o), api._ptr(s), api._ptr(f)))
    lib.gmg_mg_result_free(res)
    lib.gmg_reads_free(r_all)
    whole = (int(no.value), int(ns.value), int(o["accepted"].sum()), float(s["score"].sum()))
tot = lambda sink: (sum(x[0] for x in sink), sum(x[1] for x in sink), sum(x[2] for x in sink))
same = sink_s == sink_o and (whole is None or tot(sink_s) == whole[:3])
print(json.dumps({"reads": n_reads, "pieces": n_pieces, "file_bytes": n_data, "serial_ms": t_serial * 1e3,
                  "overlapped_ms": t_over * 1e3, "overlapped_mbases_per_s": n_reads * L / t_over / 1e6,
                  "pieces_identical_serial_vs_overlapped": sink_s == sink_o,
                  "counts_equal_one_batch": None if whole is None else tot(sink_s) == whole[:3], "ok": bool(same)}))
