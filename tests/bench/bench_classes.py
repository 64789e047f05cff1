#!/usr/bin/env python3
"""glimmer-mg's classification mode on the device (the ICM-grouped loop, glimmer-mg.cc:361-451): 1 M x 500 bp
synthetic reads over G ICM groups (default 64) and N null models (default 100 GC values), reads resident in HBM.
Timed: (a) ONE gmg_mg_score_reads over all reads with one ICM and a null model per read (the single-ICM time);
(b) the grouped job: the reads gathered in group order (gmg_reads_select) and scored group by group with each group's own ICM
handle -- one gmg_mg_score_reads per group (BENCH_PER_GROUP_CALLS=1), and ONE gmg_mg_score_groups.
BENCH_ERR=indel|sub: the same with glimmer-mg -i / -s on ragged ~400-bp reads (BASELINE configs[4]: -c together with -i).
Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _gmg_pkg  # noqa: E402

gmg = _gmg_pkg.load()
api, capi = gmg.api, gmg.capi
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_groups = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_nulls = int(sys.argv[3]) if len(sys.argv) > 3 else 100
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = 500
gmg.init(0)
lib = capi.lib()
DATA = os.path.join(ROOT, "tests", "golden", "data")
files = ["NC_000915.icm", "seqs.cluster-0.run1.filt.gicm", "seqs.cluster-2.run1.filt.gicm", "seqs.cluster-4.run1.filt.gicm",
         "seqs.cluster-5.run1.filt.gicm"]
# one handle (one table in HBM) per group: BENCH_SAME_MODEL=1 (default) 64 copies of ONE file, so that the grouped job does the
# same work as the single-ICM call and must give the same records; 0: five different files in turn
# 0: five different files in turn; distinct: n_groups DIFFERENT 3-periodic tables (the five files + models trained on
# disjoint slices of NC_000915.fna, tests/models64.py): 64 MB of tables against 4 MB of L2 per XCD
same = os.environ.get("BENCH_SAME_MODEL", "1") == "1"
# relabel: n_groups different tables with the VALUES of the five files (each file with the bases renamed by a permutation): the
# cache effect alone -- tables trained on 26-kb slices hold probabilities of zero, which take the exact paths
distinct = os.environ.get("BENCH_SAME_MODEL", "1") in ("distinct", "relabel")
if distinct:
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import models64
    _tmp = tempfile.mkdtemp()
    if os.environ["BENCH_SAME_MODEL"] == "relabel":
        models = [m for m, _ in models64.relabeled_models(gmg, _tmp, models64.GENE_FILES, n_groups)]
    else:
        models = [m for m, _ in models64.gene_models(gmg, _tmp, n_groups)]
else:
    models = [gmg.Icm.open(os.path.join(DATA, files[0 if same else g % len(files)])) for g in range(n_groups)]
err = os.environ.get("BENCH_ERR", "")
if err:                                                 # clipped N(400, 60^2) lengths, as bench_mg.py's ragged reads
    lens = np.clip(np.random.default_rng(12).normal(400, 60, n_reads).round(), 100, 700).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    packed, _ = gmg.synth.packed_reads(1, int(off[-1]), 7)
else:
    packed, off = gmg.synth.packed_reads(n_reads, L, 7)
reads = gmg.Reads(packed, off)
rng = np.random.default_rng(3)
group = rng.integers(0, n_groups, n_reads)
order = np.argsort(group, kind="stable").astype(np.uint64)
begin = np.searchsorted(group[order.astype(np.int64)], np.arange(n_groups + 1)).astype(np.uint64)
nulls = gmg.NullSet.build(np.linspace(0.3, 0.7, n_nulls))
read_null = rng.integers(0, n_nulls, n_reads).astype(np.uint32)
read_isl = np.full(n_reads, 2**31 - 1, np.int32)


rn_sorted = np.ascontiguousarray(read_null[order.astype(np.int64)])
for arr in (order, read_null, read_isl, rn_sorted):    # page-locked: the copies of the per-read arrays run at PCIe speed
    api._ck(lib.gmg_host_register(arr.ctypes.data, arr.nbytes))


def params(rn, isl):
    prm = capi.MgParams(75, 1, 2**31 - 1, 3, 3, {"": 1, "indel": 1 | 2, "sub": 1 | 4}[err], -6.0)      # GMG_MG_ACCEPTED_ONLY, as the driver asks
    prm.min_indel_orf_len, prm.indel_quality_threshold, prm.indel_max, prm.indel_suffix_score_threshold = 15, 18, 2, -12.0
    for i, c in enumerate(("atg", "gtg", "ttg")):
        prm.start_codon[i].value = c.encode()
    for i, c in enumerate(("taa", "tag", "tga")):
        prm.stop_codon[i].value = c.encode()
    prm.nulls, prm.read_null, prm.read_ignore_score_len = nulls.h, rn.ctypes.data, isl.ctypes.data
    return prm


def score(model, batch, prm):
    res = C.c_void_p()
    api._ck(lib.gmg_mg_score_reads(model.device(), nulls.icms[0].device(), batch.h, C.byref(prm), None, C.byref(res), None))
    n_orfs, n_starts = C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_mg_result_info(res, C.byref(n_orfs), C.byref(n_starts)))
    lib.gmg_mg_result_free(res)
    return n_orfs.value, n_starts.value


def single():
    return score(models[0], reads, params(read_null, read_isl))


def grouped():
    tot = [0, 0]
    rn = read_null[order.astype(np.int64)]
    for g in range(n_groups):
        b, e = int(begin[g]), int(begin[g + 1])
        if b == e:
            continue
        batch = reads.select(order[b:e])
        a = score(models[g], batch, params(np.ascontiguousarray(rn[b:e]), np.ascontiguousarray(read_isl[b:e])))
        tot[0] += a[0]
        tot[1] += a[1]
        batch.close()
    return tuple(tot)


def grouped_one_call():
    # gmg_mg_score_groups: the reads gathered in visiting order once, every group a consecutive range under its own model
    batch = reads.select(order)
    prm = params(rn_sorted, read_isl)
    arr = (capi.MgGroup * n_groups)(*[capi.MgGroup(models[g].device(), int(begin[g]), int(begin[g + 1])) for g in range(n_groups)])
    res = C.c_void_p()
    api._ck(lib.gmg_mg_score_groups(arr, n_groups, nulls.icms[0].device(), batch.h, C.byref(prm), C.byref(res), None))
    n_orfs, n_starts = C.c_uint64(), C.c_uint64()
    api._ck(lib.gmg_mg_result_info(res, C.byref(n_orfs), C.byref(n_starts)))
    lib.gmg_mg_result_free(res)
    batch.close()
    return n_orfs.value, n_starts.value


def timed(fn):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], [round(t, 2) for t in ts], r


t_single, all_single, r1 = timed(single)
t_grouped, all_grouped, r2 = timed(grouped) if os.environ.get("BENCH_PER_GROUP_CALLS", "1") == "1" else (float("nan"), [], (0, 0))
t_one, all_one, r3 = timed(grouped_one_call)
print(json.dumps({"reads": n_reads, "read_len": L if not err else "~400 (ragged)", "error_branch": err, "groups": n_groups, "null_models": n_nulls, "single_icm_ms": round(t_single, 3),
                  "score_groups_ms": round(t_one, 3), "ratio": round(t_one / t_single, 3),
                  "one_call_per_group_ms": round(t_grouped, 3), "single_all": all_single, "score_groups_all": all_one,
                  "one_call_per_group_all": all_grouped, "accepted_orfs_single": r1[0], "accepted_orfs_groups": r3[0],
                  "accepted_orfs_per_group_calls": r2[0], "same_model_file": same, "distinct_models": n_groups if distinct else (1 if same else len(files)), "models": os.environ.get("BENCH_SAME_MODEL", "1")}))
