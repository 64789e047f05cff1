#!/usr/bin/env python3
"""A/B of gmg_frame_score6 between builds of the library in ONE process on ONE GPU (boxes differ by several per cent):
    python tools/f6_ab.py libA.so libB.so ...     -> median / min ms per call of each, calls interleaved
a library may be given as  path:key=value[,key=value]  (gmg_set_option after gmg_init; use copies of one file for A/B of a switch)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gmg_pkg  # noqa: E402

synth = _gmg_pkg.load().synth
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
n, L = 1_000_000, 500
packed, off = synth.packed_reads(n, L, 20260101)
out = torch.empty(6 * n * L, dtype=torch.float64, device="cuda")
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
        for kv in filter(None, opts.split(",")):
            key, val = kv.split("=")
            lib.gmg_set_option.argtypes = [C.c_char_p, C.c_longlong]
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
        lib.gmg_frame_score6.argtypes = [vp, vp, vp, vp, vp]
        self.ms = []

    def call(self):
        ck(self.lib, self.lib.gmg_frame_score6(self.gene, self.indep, self.reads, out.data_ptr(), None))

    def digest(self):
        torch.cuda.synchronize()
        v = out.view(torch.int64)
        return int(v[::1013].sum().item()) ^ int(v[-4096:].sum().item())


builds = [Build(p) for p in sys.argv[1:]]
for b in builds:
    for _ in range(4):
        b.call()
    print(b.name, "digest %016x" % (b.digest() & (2 ** 64 - 1)))
for rep in range(15):
    for b in builds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.call()
        b.call()
        e1.record()
        e1.synchronize()
        b.ms.append(e0.elapsed_time(e1) / 2)
for b in builds:
    print("%-20s median %.4f ms  min %.4f  max %.4f   frac(median) %.4f" % (b.name, np.median(b.ms), min(b.ms), max(b.ms),
                                                                          48.25 * n * L / (np.median(b.ms) * 1e-3) / 8e12))
