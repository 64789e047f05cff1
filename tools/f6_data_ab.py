#!/usr/bin/env python3
"""gmg_frame_score6 on synthetic (iid uniform) against genome-sampled reads (NC_000915.fna, both strands) in ONE process on ONE
GPU, calls interleaved: boxes and minutes differ by several per cent, two separate bench.py runs do not resolve a 1 % effect.
    python tools/f6_data_ab.py [calls per input, default 60]      -> one JSON line"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
gmg.build.build_lib()
gmg.init(0)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n, L = 1_000_000, 500
gene, indep = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")), gmg.Icm.indep(0.5)
batches = {"synthetic": gmg.Reads(*gmg.synth.packed_reads(n, L, 20260101)),
           "genome": gmg.Reads(*gmg.synth.genome_reads(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.fna"), n, L, 20260101))}
out = torch.empty(6 * n * L, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream()
ms = {k: [] for k in batches}
for it in range(calls + 5):
    for name, reads in batches.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        gmg.frame_score6(gene, indep, reads, d_out=out.data_ptr(), stream=stream.cuda_stream)
        b.record(stream)
        b.synchronize()
        if it >= 5:
            ms[name].append(a.elapsed_time(b))
res = {k: {"median_ms": round(sorted(v)[len(v) // 2], 4), "min_ms": round(min(v), 4)} for k, v in ms.items()}
res["genome_over_synthetic"] = round(res["genome"]["median_ms"] / res["synthetic"]["median_ms"], 4)
res["frac_of_8TBps"] = {k: round(48.25 * n * L / (res[k]["median_ms"] * 1e-3) / 8e12, 4) for k in batches}
print(json.dumps(res))
