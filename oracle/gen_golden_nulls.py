#!/usr/bin/env python3
"""Golden vectors for per-read null models (glimmer-mg's classification mode rebuilds Indep_Model for every read from the
GC of its classes: Update_Meta_Null_ICM, src/Glimmer/glimmer-mg.cc:2050-2068): Score_All_Frames of the first 40 reads of
seqs.fa from the REAL reference (oracle/_ref/ref_dump frames), read i against the null model of GC value GCS[i % 8].
    python oracle/gen_golden_nulls.py          -> tests/golden/frames_multigc.npz
Build container only (needs oracle/_ref, i.e. /root/reference)."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RB = os.path.join(HERE, "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")
GCS = [0.25, 0.31, 0.38, 0.44, 0.5, 0.57, 0.63, 0.71]
N, L = 40, 500


def run(*args):
    return subprocess.run([os.path.join(RB, "ref_dump"), *[str(a) for a in args]], check=True, stdout=subprocess.PIPE).stdout


def main():
    if not os.path.exists(os.path.join(RB, "ref_dump")):
        sys.exit("build oracle/_ref first:  make -C oracle ref")
    fa, nc = os.path.join(DATA, "seqs.fa"), os.path.join(DATA, "NC_000915.icm")
    per_gc = [np.frombuffer(run("frames", nc, fa, 0, N, gc), "<f8").reshape(N, 6, L) for gc in GCS]
    frames = np.stack([per_gc[i % len(GCS)][i] for i in range(N)])
    np.savez_compressed(os.path.join(GOLD, "frames_multigc.npz"), frames=frames, gcs=np.array(GCS),
                        read_null=np.arange(N, dtype=np.uint32) % len(GCS))
    print("frames_multigc.npz:", frames.shape)


if __name__ == "__main__":
    main()
