#!/usr/bin/env python3
"""How the headline call's time develops on a fresh box: gmg_frame_score6 on 1M x 500 bp in blocks of 100 calls, one line per block
(seconds since the first GPU call, median ms of the block), for [seconds] (default 90); optionally idle [idle] seconds in the middle.
Diagnostic: some runs meet the GPU in a slow state (0.60 of the roof instead of 0.66 - 0.70)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
gmg.build.build_lib()
total_s = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
idle_s = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t_start = time.perf_counter()
gmg.init(0)
n, L = 1_000_000, 500
gene, indep = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")), gmg.Icm.indep(0.5)
reads = gmg.Reads(*gmg.synth.packed_reads(n, L, 20260101))
out = torch.empty(6 * n * L, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream()
print("set-up done at %.1f s" % (time.perf_counter() - t_start), flush=True)
idled = False
while time.perf_counter() - t_start < total_s:
    ms = []
    for _ in range(100):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        gmg.frame_score6(gene, indep, reads, d_out=out.data_ptr(), stream=stream.cuda_stream)
        b.record(stream)
        b.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    print("%6.1f s  median %.3f ms  (min %.3f, max %.3f)  frac %.3f" % (time.perf_counter() - t_start, ms[50], ms[0], ms[-1], 48.25 * n * L / (ms[50] * 1e-3) / 8e12), flush=True)
    if idle_s and not idled and time.perf_counter() - t_start > total_s / 2:
        print("idle %.0f s" % idle_s, flush=True)
        time.sleep(idle_s)
        idled = True
