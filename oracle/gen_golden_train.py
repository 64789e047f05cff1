#!/usr/bin/env python3
"""Generate tests/golden/train/* from the REAL reference's build-icm (oracle/_ref/build-icm, built by oracle/Makefile
from /root/reference; src/ICM/build-icm.cc + icm.cc).  Runs only in the build container; the tests use the committed
fixtures.  Test infrastructure only.

Training sets copied as DATA fixtures (the reference's own sample-run files):
  tests/golden/data/seqs.cluster-{3,4}.run1.filt.gene.fasta   (sample-run/glimmer-mg/results/)
For every case of CASES: the .icm the reference's build-icm writes -- whole when small, else its sha256 -- in
  tests/golden/train/<name>.icm | <name>.sha256 ; the cases themselves in tests/golden/train/cases.json.
The big case (sample-run/glimmer3/results/NC_000915.train, 1.3 MB) is checked here against the oracle and recorded
by sha256 only when the file is present; it does not travel.
"""
import hashlib
import json
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("GMG_REFERENCE", "/root/reference")
BUILD_ICM = os.path.join(HERE, "_ref", "build-icm")
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")
OUT = os.path.join(GOLD, "train")

# name, training set, build-icm options, (model_len, model_depth, periodicity), reversed strings, keep the whole file
CASES = [
    ("c4_r", "seqs.cluster-4.run1.filt.gene.fasta", ["-r"], (12, 7, 3), True, True),
    ("c3_r", "seqs.cluster-3.run1.filt.gene.fasta", ["-r"], (12, 7, 3), True, False),
    ("c3_fwd", "seqs.cluster-3.run1.filt.gene.fasta", [], (12, 7, 3), False, False),
    ("c3_p1_d5_w9", "seqs.cluster-3.run1.filt.gene.fasta", ["-p", "1", "-d", "5", "-w", "9"], (9, 5, 1), False, True),
    ("c3_p2_d3_w6_r", "seqs.cluster-3.run1.filt.gene.fasta", ["-r", "-p", "2", "-d", "3", "-w", "6"], (6, 3, 2), True, True),
    ("c4_p4_d2_w3", "seqs.cluster-4.run1.filt.gene.fasta", ["-p", "4", "-d", "2", "-w", "3"], (3, 2, 4), False, True),
    ("c4_d1_w2", "seqs.cluster-4.run1.filt.gene.fasta", ["-d", "1", "-w", "2"], (2, 1, 3), False, True),
    ("c3_w16_d8_r", "seqs.cluster-3.run1.filt.gene.fasta", ["-r", "-w", "16", "-d", "8"], (16, 8, 3), True, False),
    # our own synthetic file (written below): in-frame stops for -F to skip, upper case, ambiguity codes
    ("syn_F_r", "train_synth.fa", ["-F", "-r"], (12, 7, 3), True, False),
    ("syn_d4", "train_synth.fa", ["-d", "4"], (12, 4, 3), False, True),
    ("c4_text_d3_r", "seqs.cluster-4.run1.filt.gene.fasta", ["-r", "-t", "-d", "3"], (12, 3, 3), True, True),
]


def write_synthetic(path):
    """240 random strings of 30 - 900 characters: a third-position bias, every fourth string free of in-frame stop codons
    (so that -F keeps some), some upper case, a few ambiguity codes (build-icm counts them as Subscript (Filter (ch)))"""
    import numpy as np
    rng = np.random.default_rng(20260103)
    acgt = np.frombuffer(b"acgt", np.uint8)
    with open(path, "wb") as fp:
        for i in range(240):
            n = int(rng.integers(30, 900))
            s = rng.choice(acgt, size=n, p=[0.3, 0.2, 0.2, 0.3])
            third = np.arange(n) % 3 == 2
            s[third] = rng.choice(acgt, size=int(third.sum()), p=[0.1, 0.4, 0.4, 0.1])
            s = bytearray(s.tobytes())
            if i % 4 == 0:
                for j in range(0, n - 2, 3):
                    if bytes(s[j:j + 3]) in (b"taa", b"tag", b"tga"):
                        s[j] = ord("c")
            for k in rng.integers(0, n, size=int(rng.integers(0, 4))):
                s[int(k)] = int(rng.choice(np.frombuffer(b"nryswmkbdhv", np.uint8)))
            if i % 5 == 0:
                s = bytearray(bytes(s).upper())
            fp.write(b">syn%d\n" % i)
            for k in range(0, n, 60):
                fp.write(bytes(s[k:k + 60]) + b"\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in ("seqs.cluster-3.run1.filt.gene.fasta", "seqs.cluster-4.run1.filt.gene.fasta"):
        shutil.copyfile(os.path.join(REF, "sample-run", "glimmer-mg", "results", f), os.path.join(DATA, f))
        os.chmod(os.path.join(DATA, f), 0o644)
    write_synthetic(os.path.join(DATA, "train_synth.fa"))
    cases = []
    for name, train, opts, shape, rev, whole in CASES:
        out = os.path.join(OUT, name + ".icm")
        with open(os.path.join(DATA, train), "rb") as fp:
            subprocess.run([BUILD_ICM, *opts, out], stdin=fp, check=True)
        data = open(out, "rb").read()
        sha = hashlib.sha256(data).hexdigest()
        if not whole:
            os.remove(out)
        cases.append({"name": name, "train": train, "opts": opts, "model_len": shape[0], "model_depth": shape[1],
                      "periodicity": shape[2], "reversed": rev, "whole": whole, "text": "-t" in opts,
                      "bytes": len(data), "sha256": sha})
        print(name, len(data), sha[:16])
    big = os.path.join(REF, "sample-run", "glimmer3", "results", "NC_000915.train")
    if os.path.exists(big):
        with open(big, "rb") as fp:
            subprocess.run([BUILD_ICM, "-r", "/tmp/gmg_nc_train.icm"], stdin=fp, check=True)
        data = open("/tmp/gmg_nc_train.icm", "rb").read()
        cases.append({"name": "nc_r", "train": None, "opts": ["-r"], "model_len": 12, "model_depth": 7,
                      "periodicity": 3, "reversed": True, "whole": False, "text": False, "bytes": len(data),
                      "sha256": hashlib.sha256(data).hexdigest(),
                      "note": "sample-run/glimmer3/results/NC_000915.train; checked where /root/reference exists"})
    with open(os.path.join(OUT, "cases.json"), "w") as fp:
        json.dump(cases, fp, indent=1)


if __name__ == "__main__":
    main()
