#!/usr/bin/env python3
"""Where a round of the six-frame kernel spends its cycles: runs a GMG_F6_STAMPS build (GMG_LIB_PATH) on 1M x 500 bp and
prints, per phase, the mean over waves of the accumulated s_memtime deltas.  Diagnostic only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gmg_pkg  # noqa: E402

import torch  # noqa: E402

gmg = _gmg_pkg.load()
torch.cuda.set_device(0)                                # (torch's HIP runtime first, as in bench.py)
torch.zeros(1, device="cuda")
gmg.init(0)

n, L = 1_000_000, 500
packed, off = gmg.synth.packed_reads(n, L, 20260101)
reads = gmg.Reads(packed, off)
gene = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
indep = gmg.Icm.indep(0.5)
out = torch.empty(6 * n * L, dtype=torch.float64, device="cuda")
for _ in range(5):
    gmg.frame_score6(gene, indep, reads, d_out=out.data_ptr())
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
gmg.frame_score6(gene, indep, reads, d_out=out.data_ptr())
b.record()
torch.cuda.synchronize()
lib = gmg.capi.lib()
nw = 255 * 16
buf = np.zeros(nw * 8, np.uint64)
lib.gmg_debug_f6_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.gmg_debug_f6_stamps(buf.ctypes.data, nw * 8) == 0
st = buf.reshape(nw, 8).astype(np.float64)
names = ["phase 1", "wait barrier A", "swap work", "wait barrier B", "phase 2"]
tot = st[:, :5].sum(1)
print("call %.3f ms; cycles per wave (mean / min / max over %d waves), share of the loop:" % (a.elapsed_time(b), nw))
for i, nm in enumerate(names):
    print("  %-16s %12.0f %12.0f %12.0f   %5.1f %%" % (nm, st[:, i].mean(), st[:, i].min(), st[:, i].max(), 100 * st[:, i].mean() / tot.mean()))
print("  %-16s %12.0f  -> %.2f GHz if the loop is the whole call" % ("loop total", tot.mean(), tot.mean() / (a.elapsed_time(b) * 1e6)))
