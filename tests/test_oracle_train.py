"""The oracle's restatement of build-icm's training (orc_train_model, orc_train_level_counts; oracle/gmg_oracle.c)
against the .icm files the REAL reference's build-icm wrote from the same training sets (tests/golden/train/, made by
oracle/gen_golden_train.py): every byte of the model file must agree.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLD

TRAIN = os.path.join(GOLD, "train")
CASES = json.load(open(os.path.join(TRAIN, "cases.json")))
REF_BIG = "/root/reference/sample-run/glimmer3/results/NC_000915.train"


def training_strings(case, gmg):
    """build-icm's Read_Training_Data (src/ICM/build-icm.cc:320-340): lower case, then -r reverses every string"""
    path = os.path.join(DATA, case["train"]) if case["train"] else REF_BIG
    _, seqs = gmg.read_fasta(path)
    seqs = [s.lower().encode() for s in seqs]
    if "-F" in case["opts"]:         # Skip_In_Frame_Stop_Strings (build-icm.cc:80-109): default stop codons, before -r
        seqs = [s for s in seqs if not any(s[j:j + 3] in (b"taa", b"tag", b"tga") for j in range(0, len(s) - 2, 3))]
    return [s[::-1] for s in seqs] if case["reversed"] else seqs


def binary_cases():
    return [c for c in CASES if not c["text"] and (c["train"] or os.path.exists(REF_BIG))]


@pytest.mark.parametrize("case", binary_cases(), ids=lambda c: c["name"])
def test_oracle_training_writes_the_reference_icm(case, oracle, gmg, tmp_path):
    m = oracle.train_model(training_strings(case, gmg), case["model_len"], case["model_depth"], case["periodicity"])
    out = str(tmp_path / "o.icm")
    assert oracle.L.orc_model_write(m, out.encode()) == 0
    data = open(out, "rb").read()
    assert len(data) == case["bytes"]
    assert hashlib.sha256(data).hexdigest() == case["sha256"]
    if case["whole"]:
        assert data == open(os.path.join(TRAIN, case["name"] + ".icm"), "rb").read()
    oracle.L.orc_model_free(m)


def test_level_counts_bookkeeping(oracle, gmg):
    """Every complete window lands in exactly one root table per context position; a deeper level holds the windows
    whose ancestors all chose a position, split over the children by the base at that position."""
    case = next(c for c in CASES if c["name"] == "c4_r")
    strings = training_strings(case, gmg)
    W, D, P = 12, 7, 3
    m = oracle.train_model(strings, W, D, P)
    mip, _ = oracle.model_tables(m)
    n_windows = sum(max(len(s) - W + 1, 0) for s in strings)
    prev = None
    for level in range(D + 1):
        ct = oracle.train_level_counts(m, strings, level)          # [P, 4^level, W-1, 16]
        per_pos = ct.sum(axis=3)                                    # windows per (frame, node, position)
        assert (per_pos == per_pos[:, :, :1]).all()                 # the same total for every context position
        if level == 0:
            assert per_pos[:, :, 0].sum() == n_windows
        else:
            first_prev = (4 ** (level - 1) - 1) // 3
            for f in range(P):
                for k in range(4 ** (level - 1)):
                    kids = per_pos[f, 4 * k:4 * k + 4, 0]
                    p = mip[f, first_prev + k]
                    if p < 0:
                        assert kids.sum() == 0
                    else:                                           # children split the parent's table of position p
                        want = prev[f, k, p].reshape(4, 4).sum(axis=1)
                        assert (kids == want).all()
        prev = ct
    oracle.L.orc_model_free(m)


def test_mutual_info_of_known_tables(oracle):
    import ctypes as C
    indep = np.array([1, 2, 3, 4] * 4, np.int32)                    # rows proportional: no information
    assert abs(oracle.L.orc_mutual_info(indep.ctypes.data_as(C.POINTER(C.c_int32)), int(indep.sum()))) < 1e-15
    diag = np.zeros(16, np.int32)
    diag[[0, 5, 10, 15]] = 25                                       # context determines the base: ln 4
    assert oracle.L.orc_mutual_info(diag.ctypes.data_as(C.POINTER(C.c_int32)), 100) == pytest.approx(np.log(4), abs=1e-15)
    assert oracle.L.orc_mutual_info(diag.ctypes.data_as(C.POINTER(C.c_int32)), 0) == 0.0
