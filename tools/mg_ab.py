#!/usr/bin/env python3
"""A/B of gmg_mg_score_reads (glimmer-mg's front half, default mode) between builds of the library in ONE process on ONE GPU
(boxes differ by several per cent):
    python tools/mg_ab.py [ragged] libA.so libB.so ...     -> median / min ms per call of each, calls interleaved
a library may be given as  path:key=value[,key=value]  (gmg_set_option after gmg_init; use copies of one file for A/B of a switch).
The checksum printed per build covers the ORF records and the start lists."""
import ctypes as C
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
synth, capi = gmg.synth, gmg.capi
args = sys.argv[1:]
ragged = bool(args) and args[0] == "ragged"
if ragged:
    args = args[1:]
n, L = 1_000_000, 500
if ragged:
    lens = np.clip(np.random.default_rng(12).normal(400, 60, n).round(), 100, 700).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    packed, _ = synth.packed_reads(1, int(off[-1]), 7)
else:
    packed, off = synth.packed_reads(n, L, 7)
MODEL = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm").encode()
vp = C.c_void_p


def ck(lib, rc):
    if rc != 0:
        lib.gmg_last_error.restype = C.c_char_p
        raise RuntimeError(lib.gmg_last_error().decode())


class Build:
    def __init__(self, spec):
        path, _, opts = spec.partition(":")
        self.name = os.path.basename(spec)
        lib = self.lib = C.CDLL(path)
        ck(lib, lib.gmg_init(0))
        lib.gmg_set_option.argtypes = [C.c_char_p, C.c_longlong]
        for kv in filter(None, opts.split(",")):
            key, val = kv.split("=")
            ck(lib, lib.gmg_set_option(key.encode(), int(val)))
        gene, indep = vp(), vp()
        ck(lib, lib.gmg_icm_open(MODEL, C.byref(gene)))
        ck(lib, lib.gmg_icm_new(3, 2, 3, C.byref(indep)))
        stops = (C.c_char_p * 3)(b"taa", b"tag", b"tga")
        lib.gmg_icm_build_indep.argtypes = [vp, C.c_double, vp, C.c_int]
        ck(lib, lib.gmg_icm_build_indep(indep, 0.5, stops, 3))
        self.gene, self.indep = vp(), vp()
        ck(lib, lib.gmg_icm_device_model(gene, C.byref(self.gene)))
        ck(lib, lib.gmg_icm_device_model(indep, C.byref(self.indep)))
        self.reads = vp()
        lib.gmg_reads_upload.argtypes = [vp, vp, C.c_uint64, vp]
        ck(lib, lib.gmg_reads_upload(packed.ctypes.data, off.ctypes.data, n, C.byref(self.reads)))
        self.prm = capi.MgParams(75, 1, 2**31 - 1, 3, 3, 0, -6.0)
        for i, c in enumerate(("atg", "gtg", "ttg")):
            self.prm.start_codon[i].value = c.encode()
        for i, c in enumerate(("taa", "tag", "tga")):
            self.prm.stop_codon[i].value = c.encode()
        lib.gmg_mg_score_reads.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        lib.gmg_mg_result_free.argtypes = [vp]
        self.ms = []

    def call(self, fetch=False):
        res = vp()
        ck(self.lib, self.lib.gmg_mg_score_reads(self.gene, self.indep, self.reads, C.byref(self.prm), None, C.byref(res), None))
        crc = None
        if fetch:
            no, ns = C.c_uint64(), C.c_uint64()
            self.lib.gmg_mg_result_info.argtypes = [vp, vp, vp]
            ck(self.lib, self.lib.gmg_mg_result_info(res, C.byref(no), C.byref(ns)))
            orfs, starts, first = np.zeros(no.value * 56, np.uint8), np.zeros(ns.value * 24, np.uint8), np.zeros(n + 1, np.uint64)
            self.lib.gmg_mg_result_fetch.argtypes = [vp, vp, vp, vp]
            ck(self.lib, self.lib.gmg_mg_result_fetch(res, orfs.ctypes.data, starts.ctypes.data, first.ctypes.data))
            crc = (no.value, ns.value, "%08x" % zlib.crc32(starts.tobytes(), zlib.crc32(orfs.tobytes())))
        self.lib.gmg_mg_result_free(res)
        return crc


builds = [Build(p) for p in args]
for b in builds:
    for _ in range(2):
        b.call()
    print(b.name, b.call(fetch=True), flush=True)
for rep in range(11):
    for b in builds:
        t0 = time.perf_counter()
        b.call()
        b.call()
        b.ms.append((time.perf_counter() - t0) / 2 * 1e3)
for b in builds:
    print("%-60s median %.3f ms   min %.3f ms" % (b.name, float(np.median(b.ms)), min(b.ms)))
