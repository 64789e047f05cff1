#!/usr/bin/env python3
"""glimmer-mg's classification mode end to end on the GPU box (BASELINE configs[4], "full glimmer-mg.py path": glimmer-mg.py runs
`glimmer-mg -c <class file>`, scripts/glimmer-mg.py:85-87,103-105): the REAL reference main loop with ICM_dir set
(oracle/_ref/ref_mg_classes with GMG_REF_QUIET=1: the reference's own main(), nothing dumped) against
integration/_build/glimmer-mg_gpu -c on one synthetic FASTA file of ~400-bp reads, every read with one to three classes drawn
from the sample-run's class file (the synthetic .genomeData tree of tests/golden/make_genome_data.py: 240 ICM files, five
tables, a GC value per class, three stop-codon sets).  The two .predict files must be byte-identical.
bench_cli_classes.py [n_reads] ; BENCH_CLI_FLAGS="-i" adds glimmer-mg options to both runs; BENCH_CLI_DEV_OPTS="--shards 2";
BENCH_CHUNK=N: chunks of N reads in both (the reference's Chunk_Sequences, 500000 by default).
Prints one JSON line."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")
sys.path.insert(0, GOLD)
import make_genome_data  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_mg_classes")
DEV = os.path.join(ROOT, "integration", "_build", "glimmer-mg_gpu")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
flags = os.environ.get("BENCH_CLI_FLAGS", "").split()
dev_opts = os.environ.get("BENCH_CLI_DEV_OPTS", "").split()
chunk = os.environ.get("BENCH_CHUNK")
if chunk:
    dev_opts += ["--chunk-reads", chunk]

rng = np.random.default_rng(23)
lens = np.clip(rng.normal(400, 60, n_reads).round(), 100, 700).astype(np.int64)
classes = sorted({c for line in open(os.path.join(DATA, "seqs.class.txt")) for c in line.split()[1:]})
with tempfile.TemporaryDirectory(dir=os.environ.get("BENCH_TMP")) as tmp:
    make_genome_data.build(os.path.join(tmp, ".genomeData"), [os.path.join(DATA, "seqs.class.txt")])
    fa, cls = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "reads.class.txt")
    letters = np.frombuffer(b"acgt", np.uint8)
    with open(fa, "wb") as f, open(cls, "w") as c:
        pick = rng.integers(0, len(classes), size=(n_reads, 3))
        n_cls = rng.integers(1, 4, n_reads)
        for i in range(n_reads):
            f.write(b">read%07d\n" % i)
            f.write(letters[rng.integers(0, 4, size=int(lens[i]), dtype=np.uint8)].tobytes())
            f.write(b"\n")
            c.write("read%07d\t%s\n" % (i, " ".join(classes[k] for k in pick[i, :n_cls[i]])))

    def run(cmd, tag, env=None):
        t0 = time.perf_counter()
        res = subprocess.run(cmd + [fa, os.path.join(tmp, tag)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=tmp,
                             env=dict(os.environ, **(env or {})))
        dt = time.perf_counter() - t0
        if res.returncode != 0:
            raise SystemExit(res.stderr.decode()[-2000:])
        return dt, hashlib.md5(open(os.path.join(tmp, tag + ".predict"), "rb").read()).hexdigest(), res.stderr.decode()

    # ICM_dir as a RELATIVE name in both: the reference visits the ICM groups in the order of a hash of the file NAME
    t_ref, md5_ref, _ = run([REF, *flags, "-c", cls], "ref", dict({"GMG_REF_ICM_DIR": ".genomeData", "GMG_REF_QUIET": "1"}, **({"GMG_REF_CHUNK": chunk} if chunk else {})))
    runs = [run([DEV, "--icm-dir", ".genomeData", *dev_opts, *flags, "-c", cls], "dev%d" % i, {"GMG_CLI_TIMING": "1"}) for i in range(2)]
    genes = sum(1 for line in open(os.path.join(tmp, "ref.predict")) if line.startswith("orf"))
t_dev, _, err_dev = min(runs)
where = [line.split(": ", 1)[1] for line in err_dev.splitlines() if line.startswith("glimmer-mg_gpu timing")]
bases = int(lens.sum())
print(json.dumps({"reads": n_reads, "bases": bases, "classes": len(classes), "glimmer_mg_flags": " ".join(["-c"] + flags),
                  "glimmer_mg_gpu_options": " ".join(dev_opts), "genes_predicted": genes,
                  "predict_identical": all(m == md5_ref for _, m, _ in runs), "md5": md5_ref,
                  "reference_cli_s": round(t_ref, 3), "reference_cli_mbases_per_s": round(bases / t_ref / 1e6, 3),
                  "device_front_half_cli_s": round(t_dev, 3), "device_front_half_cli_mbases_per_s": round(bases / t_dev / 1e6, 3),
                  "speedup": round(t_ref / t_dev, 1), "device_cli_where": where,
                  "note": "process start to exit, one host thread each; the device run includes HIP start-up, reading the class file "
                          "and the per-class feature files; events / DP / trace-back are the reference's own host code in both"}))
