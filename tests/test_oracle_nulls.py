"""The oracle with one null model per read against the real reference (tests/golden/frames_multigc.npz, written by
oracle/gen_golden_nulls.py from oracle/_ref/ref_dump): glimmer-mg's classification mode rebuilds Indep_Model for every
read (Update_Meta_Null_ICM, glimmer-mg.cc:2050-2068)."""
import os

import numpy as np

from conftest import DATA, GOLD


def test_score_all_frames_with_the_reads_own_null_model(oracle, seqs_fa):
    g = np.load(os.path.join(GOLD, "frames_multigc.npz"))
    gene = oracle.read(os.path.join(DATA, "NC_000915.icm"))
    nulls = [oracle.indep(float(gc)) for gc in g["gcs"]]
    for i, want in enumerate(g["frames"]):
        got = oracle.score_all_frames(gene, nulls[int(g["read_null"][i])], oracle.filter_lower(seqs_fa[1][i]))
        assert np.array_equal(got, want), i
    # the null model matters: the same read under two GC values differs
    a = oracle.score_all_frames(gene, nulls[0], oracle.filter_lower(seqs_fa[1][0]))
    assert not np.array_equal(a, g["frames"][0]) or int(g["read_null"][0]) == 0
