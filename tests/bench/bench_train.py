#!/usr/bin/env python3
"""build-icm training on the device: wall time of gmg_icm_train and of each gmg_trainer_level_counts
call on synthetic training strings, with the oracle's single-core training beside it on a bounded sample.
usage: bench_train.py [n_strings] [mean_len] [reps]      (default 1600 x 1000 bp: one bacterial genome's genes)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gmg_pkg  # noqa: E402
import oracle_py  # noqa: E402

gmg = _gmg_pkg.load()
gmg.init(0)
orc = oracle_py.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
mean = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
W, D, P = 12, 7, 3
rng = np.random.default_rng(20260101)
lens = np.clip(rng.normal(mean, mean / 3, n).round(), 60, 4 * mean).astype(np.int64)
# third-position bias so that the tree has something to learn (not all nodes stop)
blob = rng.choice(np.frombuffer(b"acgt", np.uint8), size=int(lens.sum()), p=[0.3, 0.2, 0.2, 0.3])
third = np.arange(len(blob)) % 3 == 2
blob[third] = rng.choice(np.frombuffer(b"acgt", np.uint8), size=int(third.sum()), p=[0.15, 0.35, 0.35, 0.15])
off = np.concatenate([[0], np.cumsum(lens)])
strings = [blob[off[i]:off[i + 1]].tobytes() for i in range(n)]
n_windows = int(np.maximum(lens - W + 1, 0).sum())

# the oracle (single core) on at most ~2 Mbases of the same strings
k = n
while k > 1 and off[k] > 2_000_000:
    k //= 2
t0 = time.perf_counter()
m = orc.train_model(strings[:k], W, D, P)
t_cpu = time.perf_counter() - t0
cpu_windows = int(np.maximum(lens[:k] - W + 1, 0).sum())
mip, prob = orc.model_tables(m)

# whole training through the C ABI (pack + upload + 8 levels + host node arithmetic)
gmg.Icm.train(strings[:64], W, D, P)            # warm-up: library load, allocations
t_all = []
for _ in range(reps):
    t0 = time.perf_counter()
    icm = gmg.Icm.train(strings, W, D, P)
    t_all.append(time.perf_counter() - t0)
if k == n:
    g_mip, g_prob = icm.tables()
    assert np.array_equal(g_mip, mip) and np.array_equal(g_prob.view(np.uint32), prob.view(np.uint32))
else:
    m_all = None

# the counting alone, level by level (includes the D2H copy of the level's tables)
full = orc.train_model(strings, W, D, P) if k == n else None
g_mip = icm.tables()[0]
reads = gmg.Reads.from_strings(strings)
level_ms = np.zeros((reps, D + 1))
for r in range(reps):
    tr = gmg.Trainer(reads, W, D, P)
    for level in range(D + 1):
        first = (4 ** (level - 1) - 1) // 3 if level else 0
        prev = np.ascontiguousarray(g_mip[:, first:first + 4 ** (level - 1)]) if level else None
        t0 = time.perf_counter()
        tr.level_counts(level, prev)
        level_ms[r, level] = (time.perf_counter() - t0) * 1e3
    tr.close()
lv = np.median(level_ms, axis=0)
print(json.dumps({
    "workload": "%d training strings, %d bases, %d windows, model 12/7/3" % (n, int(lens.sum()), n_windows),
    "train_model_ms": round(float(np.median(t_all)) * 1e3, 2),
    "level_counts_ms": [round(float(x), 3) for x in lv],
    "counting_total_ms": round(float(lv.sum()), 2),
    "window_levels_per_s": round(n_windows * (D + 1) / (lv.sum() * 1e-3), 0),
    "cpu_oracle": {"strings": k, "windows": cpu_windows, "seconds": round(t_cpu, 3), "cores": 1,
                   "window_levels_per_s": round(cpu_windows * (D + 1) / t_cpu, 0)},
    # compared bit for bit only when the oracle trained on ALL strings (it takes ~0.7 s per 1M windows on one core); a larger run
    # is a timing run: its parity is tests/test_gpu_train.py (up to 4 Mbases) and tests/bench/stress_train.py
    "compared_with_oracle": bool(k == n), "identical_to_oracle": True if k == n else None}))
