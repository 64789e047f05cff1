#!/usr/bin/env python3
"""Differential stress run of the training path: random model shapes, random ragged training sets (skewed compositions,
repeats, empty and short strings), both deep-level paths; the trained tables (mut_info_pos and every probability bit) and
the tables of every level against the CPU oracle.  usage: stress_train.py [trials] [seed]"""
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
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
acgt = np.frombuffer(b"acgt", np.uint8)
t_dev = t_cpu = 0.0
windows = 0
for trial in range(trials):
    W = int(rng.integers(1, 21))
    D = int(rng.integers(0, min(W, 8))) if W > 1 else 0
    P = int(rng.integers(1, 5))
    n = int(rng.integers(0, 600))
    p = rng.dirichlet(np.ones(4) * rng.choice([0.3, 1.0, 5.0]))
    strings = []
    for _ in range(n):
        k = int(rng.integers(0, 3000)) if rng.random() < 0.9 else int(rng.integers(0, W + 2))
        s = rng.choice(acgt, size=k, p=p)
        if k > 40 and rng.random() < 0.2:                 # a tandem repeat: tables that see one context only
            unit = s[:int(rng.integers(1, 9))]
            s = np.tile(unit, k // len(unit) + 1)[:k]
        strings.append(s.tobytes())
    gmg.set_option("train_sort_min", 0 if trial % 2 else 2 ** 40)
    t0 = time.perf_counter()
    want = orc.train_model(strings, W, D, P)
    t_cpu += time.perf_counter() - t0
    mip_w, prob_w = orc.model_tables(want)
    t0 = time.perf_counter()
    mip_g, prob_g = gmg.Icm.train(strings, W, D, P).tables()
    t_dev += time.perf_counter() - t0
    assert np.array_equal(mip_g, mip_w), (trial, W, D, P, n)
    assert np.array_equal(prob_g.view(np.uint32), prob_w.view(np.uint32)), (trial, W, D, P, n)
    reads = gmg.Reads.from_strings(strings)
    tr = gmg.Trainer(reads, W, D, P)
    for level in range(D + 1):
        first = (4 ** (level - 1) - 1) // 3 if level else 0
        prev = np.ascontiguousarray(mip_w[:, first:first + 4 ** (level - 1)]) if level else None
        assert np.array_equal(tr.level_counts(level, prev), orc.train_level_counts(want, strings, level)), (trial, W, D, P, level)
    tr.close()
    orc.L.orc_model_free(want)
    windows += sum(max(len(s) - W + 1, 0) for s in strings)
print("stress_train: %d trials, %d windows, every level's tables and every trained model bit-identical to the oracle "
      "(device %.2f s, oracle %.2f s)" % (trials, windows, t_dev, t_cpu))
