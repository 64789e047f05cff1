#!/usr/bin/env python3
"""End-to-end check and timing on the GPU box (SURVEY 8d "(ii) CLI end to end"): the all-reference glimmer-mg binary
(oracle/_ref/glimmer-mg, built in the build container from the reference's own sources) against
integration/_build/glimmer-mg_gpu -- the same reference main loop, events, DP and trace-back, with FASTA parsing, Score_All_Frames,
Find_Orfs and Score_Orfs_Errors replaced by gmg_fasta_ingest + ONE gmg_mg_score_reads call -- on one synthetic FASTA file.
The two .predict files must be byte-identical.  Prints one JSON line.
BENCH_CLI_FLAGS="-i" (or "-s") adds glimmer-mg options to both runs: the error branch end to end (glimmer3 is skipped)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref")
DEV = os.path.join(ROOT, "integration", "_build")
dev_opts = os.environ.get("BENCH_CLI_DEV_OPTS", "").split()          # e.g. "--shards 2" (glimmer-mg_gpu's own options)
ICM = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
L = 500
flags = os.environ.get("BENCH_CLI_FLAGS", "").split()

rng = np.random.default_rng(17)
# reads cut from a random "genome" with planted long ORFs would be more gene-like; uniform random bases are what
# BASELINE's synthetic configs use
bases = np.frombuffer(b"acgt", np.uint8)[rng.integers(0, 4, size=(n_reads, L), dtype=np.uint8)]
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, "reads.fa")
    with open(fa, "wb") as f:
        for i in range(n_reads):
            f.write(b">read%07d\n" % i)
            f.write(bases[i].tobytes())
            f.write(b"\n")

    def run(cmd, tag):
        t0 = time.perf_counter()
        res = subprocess.run(cmd + [fa, os.path.join(tmp, tag)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        if res.returncode != 0:
            raise SystemExit(res.stderr.decode()[-2000:])
        return dt, hashlib.md5(open(os.path.join(tmp, tag + ".predict"), "rb").read()).hexdigest()

    if flags or os.environ.get("BENCH_CLI_SKIP_G3"):
        t3_ref = t3_dev = 0.0
        md5_3ref = md5_3dev = None
    else:
        t3_ref, md5_3ref = run([os.path.join(REF, "glimmer3"), "-m", ICM], "g3ref")
        t3_dev, md5_3dev = min(run([os.path.join(DEV, "glimmer3_gpu"), "-m", ICM], "g3dev%d" % i) for i in range(2))
    t_ref, md5_ref = run([os.path.join(REF, "glimmer-mg"), *flags, "-m", ICM], "ref")
    t_dev, md5_dev = run([os.path.join(DEV, "glimmer-mg_gpu"), *dev_opts, *flags, "-m", ICM], "dev")
    t_dev2, md5_dev2 = run([os.path.join(DEV, "glimmer-mg_gpu"), *dev_opts, *flags, "-m", ICM], "dev2")
    genes = sum(1 for line in open(os.path.join(tmp, "ref.predict")) if line.startswith("orf"))
print(json.dumps({"reads": n_reads, "bases": n_reads * L, "glimmer_mg_flags": " ".join(flags), "glimmer_mg_gpu_options": " ".join(dev_opts), "genes_predicted": genes,
                  "predict_identical": md5_ref == md5_dev == md5_dev2, "md5": md5_ref,
                  "reference_cli_s": round(t_ref, 3), "reference_cli_mbases_per_s": round(n_reads * L / t_ref / 1e6, 3),
                  "device_front_half_cli_s": round(min(t_dev, t_dev2), 3),
                  "device_front_half_cli_mbases_per_s": round(n_reads * L / min(t_dev, t_dev2) / 1e6, 3),
                  "glimmer3_predict_identical": md5_3ref == md5_3dev, "glimmer3_reference_cli_s": round(t3_ref, 3),
                  "glimmer3_device_front_half_cli_s": round(t3_dev, 3),
                  "note": "process start to exit, one host thread each; the device run includes HIP start-up (~1 s), and "
                          "its events / DP / trace-back are the reference's own host code"}))
