#!/usr/bin/env python3
"""Differential stress run of k_mg_tile_starts (glimmer-mg's front half, default mode: running sums as a parallel scan fused with
the start lists; the ORF scan's write pass over queued events) against the round-1 kernels (mg_fused = 0, mg_orfs_events = 0: the
reference's order of additions, every position visited): random batch shapes, codon
sets, Min_Gene_Len, truncation, Ignore_Score_Len, thresholds, one null model or one per read, both table forms, every tile size.
Every byte of the ORF records and start lists must agree.  usage: stress_mg_fused.py [configurations] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
gmg.init(0)
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
gene = gmg.Icm.open(os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm"))
gcs = np.linspace(0.25, 0.75, 23)
nulls = [gmg.Icm.indep(float(g)) for g in gcs]
null_set = gmg.NullSet(nulls)
START_SETS = [("atg", "gtg", "ttg"), ("atg",), ("atg", "rtg", "ttg", "ctg"), ("nnn",), ("atg", "gtg")]
STOP_SETS = [("taa", "tag", "tga"), ("taa", "tag"), ("tga",), ("taa", "tag", "tga", "tta")]
t_start = time.perf_counter()
tot_orfs = tot_starts = 0
for cfg in range(n_cfg):
    kind = rng.integers(0, 5)
    if kind == 0:                                       # uniform
        L = int(rng.choice([3, 40, 100, 250, 500, 566, 567, 568, 900, 1134, 1135, 2000, 2268, 2269, 3000]))
        n = max(30, min(20000, 2_000_000 // max(L, 1)))
        packed, off = gmg.synth.packed_reads(n, L, int(rng.integers(1, 1 << 30)))
        reads = gmg.Reads(packed, off)
    else:
        mean = float(rng.choice([60, 150, 400, 700, 1500]))
        lens = np.clip(rng.normal(mean, mean * rng.choice([0.05, 0.3, 0.8]), int(rng.integers(200, 6000))).round(), 0, 5000).astype(np.uint64)
        if kind == 2:
            lens[rng.integers(0, len(lens), len(lens) // 10)] = 0
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        packed, _ = gmg.synth.packed_reads(1, max(int(off[-1]), 1), int(rng.integers(1, 1 << 30)))
        if kind == 3:                                   # low-complexity stretches: long stop-free regions, dense start / stop codons
            w = packed.view(np.uint32)
            for _ in range(int(rng.integers(1, 40))):
                a = int(rng.integers(0, max(len(w) - 64, 1)))
                w[a:a + int(rng.integers(1, 64))] = int(rng.choice([0x00000000, 0xE4E4E4E4, 0x93939393, 0x3C3C3C3C, 0xCE4CE4CE]))
        reads = gmg.Reads(packed, off) if int(off[-1]) else gmg.Reads.from_strings(["" for _ in lens])
    kw = dict(min_gene_len=int(rng.choice([4, 30, 75, 90, 198, 199, 400])), allow_truncated=bool(rng.integers(0, 2)),
              ignore_score_len=int(rng.choice([2 ** 31 - 1, 10, 150, 400])), start_threshold=float(rng.choice([-6.0, -1e300, 0.0, 3.5])),
              start_codons=START_SETS[int(rng.integers(0, len(START_SETS)))], stop_codons=STOP_SETS[int(rng.integers(0, len(STOP_SETS)))])
    per_read = bool(rng.integers(0, 3) == 0)
    if per_read:
        nul = null_set
        kw["read_null"] = rng.integers(0, len(gcs), reads.n_reads).astype(np.uint32)
        if rng.integers(0, 2):
            kw["read_ignore_score_len"] = rng.choice([2 ** 31 - 1, 60, 200], reads.n_reads).astype(np.int32)
    else:
        nul = gmg.Icm.indep(float(rng.uniform(0.3, 0.7)), kw["stop_codons"]) if len(kw["stop_codons"][0]) == 3 else nulls[5]
    with gmg.option("mg_fused", 0), gmg.option("mg_orfs_events", 0), gmg.option("mg_gene32", int(rng.integers(0, 3))):
        want = gmg.mg_score_reads(gene, nul, reads, **kw)
    for tile in (0, 1, 2, 4):
        for g32 in (1, 0, 2):
            with gmg.option("mg_tile", tile), gmg.option("mg_gene32", g32):
                got = gmg.mg_score_reads(gene, nul, reads, **kw)
            ok = np.array_equal(got[2], want[2]) and got[0].tobytes() == want[0].tobytes() and got[1].tobytes() == want[1].tobytes()
            if not ok:
                print("MISMATCH cfg", cfg, "kind", kind, "tile", tile, "gene32", g32, kw, flush=True)
                sys.exit(1)
    tot_orfs += len(want[0])
    tot_starts += len(want[1])
    if cfg % 10 == 9:
        print("%d configurations, %d ORFs, %d starts identical (%.0f s)" % (cfg + 1, tot_orfs, tot_starts, time.perf_counter() - t_start), flush=True)
print("all %d configurations identical: %d ORFs, %d starts x 12 (tile size, table form) variants" % (n_cfg, tot_orfs, tot_starts))
