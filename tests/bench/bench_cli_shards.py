#!/usr/bin/env python3
"""Where does sharding the CLI pay?    integration/_build/glimmer-mg_gpu on ONE synthetic FASTA
file of n reads x 500 bp with --shards 1, 2, 4, 8 (all on --gpus G GPUs, default 1: the 16 host cores of a 1-GPU box then run
the reference's single-threaded back half -- events, DP, trace-back, fprintf -- of their shard side by side).  Every run's
<tag>.predict must be the bytes of the one-shard run (the reference CLI itself would need 45 s per 200 k reads: it is timed by
bench_cli.py, not here).  Also reported: the wall time of a run that does NOTHING but start, bind a GPU and exit
(glimmer-mg_gpu on a one-read file): what every shard pays before its first read.
    bench_cli_shards.py [n_reads] [shards ...]        prints one JSON line"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "integration", "_build", "glimmer-mg_gpu")
ICM = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
shard_counts = [int(x) for x in sys.argv[2:]] or [1, 2, 4]       # (a GPU box allows six processes on its card)
n_gpus = int(os.environ.get("BENCH_GPUS", "1"))
flags = os.environ.get("BENCH_CLI_FLAGS", "").split()
L = 500

with tempfile.TemporaryDirectory(dir=os.environ.get("BENCH_TMP", None)) as tmp:
    fa = os.path.join(tmp, "reads.fa")
    rng = np.random.default_rng(17)
    lut = np.frombuffer(b"acgt", np.uint8)
    t0 = time.perf_counter()
    with open(fa, "wb") as f:                                   # 200 k reads per piece: header, bases, newline as one byte matrix
        for r0 in range(0, n_reads, 200_000):
            m = min(200_000, n_reads - r0)
            rows = np.empty((m, 13 + L + 1), np.uint8)
            rows[:, :13] = np.frombuffer(b"".join(b">read%07d\n" % (r0 + i) for i in range(m)), np.uint8).reshape(m, 13)
            rows[:, 13:13 + L] = lut[rng.integers(0, 4, size=(m, L), dtype=np.uint8)]
            rows[:, -1] = 10
            f.write(rows.tobytes())
    t_file = time.perf_counter() - t0
    empty = os.path.join(tmp, "one.fa")                          # one read: start-up, model upload, nothing else to speak of
    open(empty, "wb").write(b">read\n" + b"acgtgctagg" * 50 + b"\n")

    def run(opts, tag, fasta=fa):
        t0 = time.perf_counter()
        res = subprocess.run([EXE, *opts, *flags, "-m", ICM, fasta, os.path.join(tmp, tag)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        if res.returncode != 0:
            raise SystemExit(res.stderr.decode()[-2000:])
        h = hashlib.md5()
        with open(os.path.join(tmp, tag + ".predict"), "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        os.unlink(os.path.join(tmp, tag + ".predict"))
        return dt, h.hexdigest()

    startup = min(run([], "e%d" % i, empty)[0] for i in range(3))
    rows, md5 = [], None
    for s in shard_counts:
        best = None
        for rep in range(2):
            dt, h = run(["--shards", str(s), "--gpus", str(n_gpus)], "s%d_%d" % (s, rep))
            md5 = md5 or h
            if h != md5:
                raise SystemExit("--shards %d wrote other bytes than --shards %d" % (s, shard_counts[0]))
            best = dt if best is None or dt < best else best
        rows.append({"shards": s, "seconds": round(best, 3), "mbases_per_s": round(n_reads * L / best / 1e6, 1)})
base = rows[0]["seconds"]
for r in rows:
    r["speedup_vs_first"] = round(base / r["seconds"], 2)
print(json.dumps({"reads": n_reads, "bases": n_reads * L, "gpus": n_gpus, "host_cores": len(os.sched_getaffinity(0)), "glimmer_mg_flags": " ".join(flags),
                  "predict_md5": md5, "all_runs_identical": True, "startup_only_s": round(startup, 3), "runs": rows,
                  "fasta_file_written_in_s": round(t_file, 1),
                  "note": "process start to exit; every shard is a forked process that pays the start-up (HIP runtime + model upload) once"}))
