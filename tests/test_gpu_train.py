"""build-icm's training through the C ABI: gmg_trainer_* (the pair counts of every tree level, on the
device), gmg_icm_train (ICM_Training_t::Train_Model on top of them) and the reference's own build-icm.cc recompiled
against our icm.hh (integration/_build/build-icm_dropin, when the build container made it) against
  * the .icm files the REAL reference's build-icm wrote (tests/golden/train/): byte-identical model files,
  * the oracle's counts, level by level, on the golden training sets and on ragged random strings (empty strings,
    strings shorter than the window, every model shape the reference's CLI accepts)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import built_binary, DATA, GOLD, ROOT
from test_oracle_train import CASES, TRAIN, training_strings

pytestmark = pytest.mark.gpu

DROPIN = os.path.join(ROOT, "integration", "_build", "build-icm_dropin")
SHIPPED = [c for c in CASES if c["train"]]


def level_mip(mip, level):
    first = (4 ** level - 1) // 3
    return np.ascontiguousarray(mip[:, first:first + 4 ** level])


def check_levels(gpu, oracle, strings, W, D, P):
    """device counts == oracle counts at every level of the tree the oracle trains on these strings"""
    m = oracle.train_model(strings, W, D, P)
    mip, _ = oracle.model_tables(m)
    reads = gpu.Reads.from_strings(strings)
    tr = gpu.Trainer(reads, W, D, P)
    for level in range(D + 1):
        got = tr.level_counts(level, level_mip(mip, level - 1) if level else None)
        want = oracle.train_level_counts(m, strings, level)
        assert got.shape == want.shape
        assert np.array_equal(got, want), (W, D, P, level)
    tr.close()
    oracle.L.orc_model_free(m)
    return mip


@pytest.mark.parametrize("case", [c for c in SHIPPED if not c["text"]], ids=lambda c: c["name"])
def test_counts_of_every_level_equal_the_oracle(case, gpu, oracle, deep_level_path):
    check_levels(gpu, oracle, training_strings(case, gpu), case["model_len"], case["model_depth"], case["periodicity"])


@pytest.mark.parametrize("case", [c for c in SHIPPED if not c["text"]], ids=lambda c: c["name"])
def test_trained_model_file_is_the_reference_file(case, gpu, tmp_path):
    icm = gpu.Icm.train(training_strings(case, gpu), case["model_len"], case["model_depth"], case["periodicity"])
    out = str(tmp_path / "t.icm")
    icm.write(out)
    data = open(out, "rb").read()
    assert len(data) == case["bytes"]
    assert hashlib.sha256(data).hexdigest() == case["sha256"]
    if case["whole"]:
        assert data == open(os.path.join(TRAIN, case["name"] + ".icm"), "rb").read()
    # the trained model scores: its device mirror uploads and a read gets a finite Score_String
    if case["model_len"] >= 2:
        reads = gpu.Reads.from_strings(["acgtacgtgacgatcgatcgatcgatgcatgcatgcatcgatcgatgcatgcatcgtagc"])
        s = gpu.score_reads_strings([icm], reads)
        assert np.isfinite(s).all() and (s < 0).all()


@pytest.mark.parametrize("case", SHIPPED, ids=lambda c: c["name"])
def test_reference_build_icm_cli_on_our_icm_hh(case, tmp_path):
    """src/ICM/build-icm.cc, unchanged, compiled against glimmer-mg_amd/host/icm.hh and linked with libgmg.so: same
    options, same bytes out (binary and -t text form, which also prints the mutual information of every node)"""
    out = str(tmp_path / "d.icm")
    built_binary("integration", "_build", "build-icm_dropin")
    with open(os.path.join(DATA, case["train"]), "rb") as fp:
        subprocess.run([DROPIN, *case["opts"], out], stdin=fp, check=True, timeout=300)
    data = open(out, "rb").read()
    assert hashlib.sha256(data).hexdigest() == case["sha256"]


@pytest.fixture(params=["atomics", "sorted"])
def deep_level_path(request):
    """both ways of counting the levels that do not fit LDS: direct device-wide atomics (small training sets) and the
    sort by table + LDS counting (big ones); the option train_sort_min is the size, in bases, where the library switches"""
    gmg = request.getfixturevalue("gpu")
    with gmg.option("train_sort_min", 0 if request.param == "sorted" else 2 ** 40):
        yield request.param


def random_strings(rng, n, max_len, extra=()):
    lens = [int(x) for x in rng.integers(0, max_len, size=n)] + list(extra)
    return [bytes(rng.choice(np.frombuffer(b"acgt", np.uint8), size=k).tobytes()) for k in lens]


@pytest.mark.parametrize("shape", [(12, 7, 3), (12, 4, 1), (5, 3, 2), (2, 1, 3), (20, 5, 3), (32, 3, 5), (3, 2, 7)],
                         ids=lambda s: "w%d_d%d_p%d" % s)
def test_ragged_random_strings_every_shape(shape, gpu, oracle, deep_level_path):
    W, D, P = shape
    rng = np.random.default_rng(100 + W)
    strings = random_strings(rng, 300, 2500, extra=(0, 0, 1, W - 1, W, W + 1, 1023, 1024, 1025, 4096))
    strings[0] = b""                       # empty strings at both ends of the batch share offsets with neighbours
    strings.append(b"")
    check_levels(gpu, oracle, strings, W, D, P)


def test_skewed_strings_stop_the_tree_early(gpu, oracle, deep_level_path):
    """few, very regular strings: most nodes see no windows, the tree stops (mut_info_pos -1 / -2) and the deeper
    levels must count nothing below a stop"""
    strings = [b"acg" * 400, b"a" * 900, b"acgtt" * 300, b"gattaca" * 50]
    mip = check_levels(gpu, oracle, strings, 12, 7, 3)
    assert (mip < 0).any()
    icm = gpu.Icm.train(strings, 12, 7, 3)
    got, _ = icm.tables()
    assert np.array_equal(got, mip)


def test_model_len_1_and_depth_0(gpu, oracle):
    """Count_Single_Chars (src/ICM/icm.cc:1874-1896): no context at all"""
    rng = np.random.default_rng(9)
    strings = random_strings(rng, 50, 400)
    for W, P in ((1, 3), (4, 2), (12, 3)):
        want = oracle.train_model(strings, W, 0, P)
        mip_w, prob_w = oracle.model_tables(want)
        icm = gpu.Icm.train(strings, W, 0, P)
        mip_g, prob_g = icm.tables()
        assert np.array_equal(mip_g, mip_w)
        assert np.array_equal(prob_g.view(np.uint32), prob_w.view(np.uint32))


def test_trained_tables_bit_identical_to_the_oracle_on_random_strings(gpu, oracle):
    rng = np.random.default_rng(77)
    strings = random_strings(rng, 400, 1500)
    for W, D, P in ((12, 7, 3), (8, 6, 1)):
        want = oracle.train_model(strings, W, D, P)
        mip_w, prob_w = oracle.model_tables(want)
        mip_g, prob_g = gpu.Icm.train(strings, W, D, P).tables()
        assert np.array_equal(mip_g, mip_w)
        assert np.array_equal(prob_g.view(np.uint32), prob_w.view(np.uint32))


def test_ambiguity_codes_and_upper_case_count_as_subscript_maps_them(gpu, oracle):
    """Subscript (Filter (ch)) (src/ICM/icm.cc:2008-2027, src/Common/gene.cc:1139-1175): r -> g, y -> c, ..., anything
    else -> c; the training counts must see the same codes"""
    rng = np.random.default_rng(12)
    alphabet = np.frombuffer(b"acgtacgtacgtnryswmkbdhvxACGTN-", np.uint8)
    strings = [bytes(rng.choice(alphabet, size=int(k)).tobytes()) for k in rng.integers(0, 1200, size=120)]
    check_levels(gpu, oracle, strings, 12, 5, 3)
    want = oracle.train_model(strings, 12, 5, 3)
    mip_w, prob_w = oracle.model_tables(want)
    mip_g, prob_g = gpu.Icm.train(strings, 12, 5, 3).tables()
    assert np.array_equal(mip_g, mip_w) and np.array_equal(prob_g.view(np.uint32), prob_w.view(np.uint32))


def test_no_strings_and_argument_errors(gpu, oracle):
    want = oracle.train_model([], 12, 2, 3)
    mip_w, prob_w = oracle.model_tables(want)
    mip_g, prob_g = gpu.Icm.train([], 12, 2, 3).tables()
    assert np.array_equal(mip_g, mip_w) and np.array_equal(prob_g.view(np.uint32), prob_w.view(np.uint32))
    reads = gpu.Reads.from_strings([b"acgtacgtacgtacgtacgt"])
    tr = gpu.Trainer(reads, 12, 2, 3)
    with pytest.raises(gpu.GmgError):            # levels go in order
        tr.level_counts(1, np.zeros((3, 1), np.int16))
    tr.level_counts(0)
    with pytest.raises(gpu.GmgError):            # a context position outside the window
        tr.level_counts(1, np.full((3, 1), 11, np.int16))
    with pytest.raises(gpu.GmgError):
        tr.level_counts(1, None)
    tr.level_counts(1, np.full((3, 1), 10, np.int16))
    with pytest.raises(gpu.GmgError):            # model shapes the device side refuses
        gpu.Trainer(reads, 33, 2, 3)
    with pytest.raises(gpu.GmgError):
        gpu.Trainer(reads, 12, 2, 0)
    with pytest.raises(gpu.GmgError):            # more counters on the last level than 32 bits index
        gpu.Trainer(reads, 32, 12, 3)


def test_full_size_training_set_properties(gpu, oracle, deep_level_path):
    """a Phymm-scale genome's worth of genes (4,000 strings, ~4 Mbases): counts are deterministic, every level keeps
    the window bookkeeping (see test_oracle_train.test_level_counts_bookkeeping), the model equals the oracle's"""
    rng = np.random.default_rng(5)
    strings = random_strings(rng, 4000, 2000)
    W, D, P = 12, 7, 3
    want = oracle.train_model(strings, W, D, P)
    mip, prob = oracle.model_tables(want)
    reads = gpu.Reads.from_strings(strings)
    n_windows = sum(max(len(s) - W + 1, 0) for s in strings)
    for rep in range(2):
        tr = gpu.Trainer(reads, W, D, P)
        prev = None
        for level in range(D + 1):
            ct = tr.level_counts(level, level_mip(mip, level - 1) if level else None)
            per_pos = ct.sum(axis=3)
            assert (per_pos == per_pos[:, :, :1]).all()
            if level == 0:
                assert per_pos[:, :, 0].sum() == n_windows
            else:
                p = level_mip(mip, level - 1)                                  # [P, 4^(level-1)]
                kids = per_pos[:, :, 0].reshape(P, -1, 4)
                for f in range(P):
                    live = p[f] >= 0
                    assert kids[f][~live].sum() == 0
                    idx = np.nonzero(live)[0]
                    want_kids = prev[f, idx, p[f, idx]].reshape(-1, 4, 4).sum(axis=2)
                    assert np.array_equal(kids[f][idx], want_kids)
            prev = ct
        tr.close()
    mip_g, prob_g = gpu.Icm.train(strings, W, D, P).tables()
    assert np.array_equal(mip_g, mip) and np.array_equal(prob_g.view(np.uint32), prob.view(np.uint32))
