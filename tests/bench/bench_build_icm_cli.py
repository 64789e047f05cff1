#!/usr/bin/env python3
"""build-icm from the command line: the reference's own binary (oracle/_ref/build-icm, all-reference build) beside the same
build-icm.cc compiled against our icm.hh (integration/_build/build-icm_dropin) on one synthetic training file; the model files
must be byte-identical.  Process start and HIP initialisation are inside the drop-in's time.
usage: bench_build_icm_cli.py [n_strings] [mean_len] [reps]"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "build-icm")
DROP = os.path.join(ROOT, "integration", "_build", "build-icm_dropin")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
mean = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(20260101)
lens = np.clip(rng.normal(mean, mean / 3, n).round(), 60, 4 * mean).astype(np.int64)
blob = rng.choice(np.frombuffer(b"acgt", np.uint8), size=int(lens.sum()), p=[0.3, 0.2, 0.2, 0.3])
third = np.arange(len(blob)) % 3 == 2
blob[third] = rng.choice(np.frombuffer(b"acgt", np.uint8), size=int(third.sum()), p=[0.15, 0.35, 0.35, 0.15])
off = np.concatenate([[0], np.cumsum(lens)])
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "train.fa")
with open(fa, "wb") as fp:
    for i in range(n):
        fp.write(b">gene%d\n" % i)
        s = blob[off[i]:off[i + 1]].tobytes()
        for k in range(0, len(s), 70):
            fp.write(s[k:k + 70] + b"\n")


def run(binary, out):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        with open(fa, "rb") as fp:
            subprocess.run([binary, "-r", out], stdin=fp, check=True)
        best = min(best, time.perf_counter() - t0)
    return best


t_ref = run(REF, os.path.join(tmp, "ref.icm"))
t_drop = run(DROP, os.path.join(tmp, "drop.icm"))
same = open(os.path.join(tmp, "ref.icm"), "rb").read() == open(os.path.join(tmp, "drop.icm"), "rb").read()
print(json.dumps({"workload": "build-icm -r on %d strings, %d bases" % (n, int(lens.sum())),
                  "reference_build_icm_s": round(t_ref, 3), "dropin_build_icm_s": round(t_drop, 3),
                  "byte_identical": same}))
assert same
