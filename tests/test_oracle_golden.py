"""Pins the CPU oracle (oracle/gmg_oracle.c) to the reference:
 - the reference's own committed outputs (icm-N.scores.tmp, 6 x 999 Score_String values), and
 - golden vectors dumped from the real reference objects (oracle/gen_golden.py).
Exact equality everywhere: every value is an fp32 table entry or a sequential double sum of them.
CPU only (-m "not gpu")."""
import hashlib
import os

import numpy as np
import pytest

from conftest import DATA, GOLD


@pytest.fixture(scope="module")
def nc(oracle):
    return oracle.read(os.path.join(DATA, "NC_000915.icm"))


@pytest.fixture(scope="module")
def reads(oracle, seqs_fa):
    return [oracle.filter_lower(s) for s in seqs_fa[1]]


def test_reference_scores_tmp_all_six_models(oracle, seqs_fa, reads):
    """sample-run/glimmer-mg/results/icm-N.scores.tmp: Score_String(read, len, 0) printed %10.4f"""
    hdrs = seqs_fa[0]
    checked = 0
    for k in range(6):
        m = oracle.read(os.path.join(DATA, "cluster-%d.icm" % k))
        assert m.contents.periodicity == 1 and m.contents.model_len == 12
        with open(os.path.join(DATA, "icm-%d.scores.tmp" % k)) as fh:
            lines = fh.read().splitlines()
        assert len(lines) == len(reads) == 999
        for line, hdr, s in zip(lines, hdrs, reads):
            name, val = line.split("\t")
            assert name.strip() == hdr.split()[0]
            assert "%.4f" % oracle.score_string(m, s, 0) == val.strip()
            checked += 1
    assert checked == 5994


def test_frames_nc_first_reads_and_hash_of_all(oracle, nc, reads):
    g = np.load(os.path.join(GOLD, "frames_nc.npz"))
    indep = oracle.indep(float(g["gc"]))
    h = hashlib.sha256()
    for r, s in enumerate(reads):
        out = oracle.score_all_frames(nc, indep, s)
        if r < g["frames"].shape[0]:
            assert np.array_equal(out, g["frames"][r]), "read %d" % r
        h.update(out.tobytes())
    assert h.hexdigest() == str(g["sha256_all"])


def test_frames_small_model_other_stops(oracle, reads):
    g = np.load(os.path.join(GOLD, "frames_gicm.npz"))
    m = oracle.read(os.path.join(DATA, "seqs.cluster-4.run1.filt.gicm"))
    indep = oracle.indep(float(g["gc"]), tuple(str(g["stops"]).split(",")))
    for r in range(g["frames"].shape[0]):
        assert np.array_equal(oracle.score_all_frames(m, indep, reads[r]), g["frames"][r])


def test_score_string_whole_reads(oracle, nc, reads):
    g = np.load(os.path.join(GOLD, "sstring.npz"))
    c4 = oracle.read(os.path.join(DATA, "cluster-4.icm"))
    for r, s in enumerate(reads):
        for f in range(3):
            assert oracle.score_string(nc, s, f) == g["nc"][r, f]
            assert oracle.score_string(c4, s, 0) == g["cluster4"][r, f]


def test_cumulative_score_and_all_frame_on_orf_buffers(oracle, nc, reads):
    g = np.load(os.path.join(GOLD, "segs.npz"))
    indep = oracle.indep(float(g["gc"]))
    off = 0
    for i, (r, lo, ln, strand) in enumerate(g["segs"]):
        orient = 1 if strand > 0 else 2
        buf = oracle.buffer(reads[r], int(lo), int(ln), orient)
        assert np.array_equal(oracle.cumulative_score(nc, buf, 1), g["gene_cum"][off:off + ln])
        assert np.array_equal(oracle.cumulative_score(indep, buf, 1), g["indep_cum"][off:off + ln])
        off += ln
        # frame 3 is the identity permutation: raw order of glimmer3.cc:346-354
        assert np.array_equal(oracle.all_frame_score(nc, buf, int(ln), 3), g["allframe_raw"][i])
    assert off == g["gene_cum"].size


def test_all_frame_permutation_is_a_permutation(oracle, nc, reads):
    buf = oracle.buffer(reads[3], 20, 300, 1)
    raw = oracle.all_frame_score(nc, buf, 300, 3)
    expect = {1: [2, 0, 1, 5, 3, 4], 2: [1, 2, 0, 4, 5, 3], -1: [3, 5, 4, 0, 2, 1],
              -2: [4, 3, 5, 1, 0, 2], -3: [5, 4, 3, 2, 1, 0]}      # glimmer3.cc:1013-1088
    for fr, perm in expect.items():
        assert np.array_equal(oracle.all_frame_score(nc, buf, 300, fr), raw[perm])


def test_full_windows(oracle, nc):
    g = np.load(os.path.join(GOLD, "windows.npz"))
    for i in range(g["windows"].shape[0]):
        w = g["windows"][i].tobytes()
        for f in range(3):
            p, dist = oracle.full_window(nc, w, f)
            assert p == g["prob"][i, f]
            assert np.array_equal(dist, g["dist"][i, f])


def test_partial_windows(oracle, nc, reads):
    g = np.load(os.path.join(GOLD, "partial.npz"))
    c4 = oracle.read(os.path.join(DATA, "cluster-4.icm"))
    for r in range(g["nc"].shape[0]):
        for i in range(11):
            for f in range(3):
                assert oracle.partial_window(nc, i, reads[r], f) == g["nc"][r, f, i]
            assert oracle.partial_window(c4, i, reads[r], 0) == g["cluster4"][r, 0, i]


def test_null_model_tables(oracle):
    g = np.load(os.path.join(GOLD, "indep.npz"))
    keys = sorted(k[:-5] for k in g.files if k.endswith("_prob"))
    assert len(keys) == 10
    for key in keys:
        gc = float(key[2:key.index("_")])
        stops = tuple(key[key.index("_") + 1:].split("-"))
        mip, prob = oracle.tables(oracle.indep(gc, stops))
        assert np.array_equal(prob.view(np.uint32), g[key + "_prob"].view(np.uint32)), key
        # the reference writer drops nodes with mip < -1 and a reader marks them -2; the builder
        # itself leaves calloc zeros there.  Compare where the golden stream had the node.
        present = g[key + "_mip"] != -2
        assert np.array_equal(mip[present], g[key + "_mip"][present])


def test_model_reader_and_writer_round_trip(oracle, nc, tmp_path):
    out = tmp_path / "rt.icm"
    assert oracle.L.orc_model_write(nc, str(out).encode()) == 0
    assert out.read_bytes() == open(os.path.join(DATA, "NC_000915.icm"), "rb").read()
    mip, prob = oracle.tables(nc)
    assert mip.shape == (3, 21845) and (mip == -2).sum() == 3 * 21845 - 62743


def test_cumulative_frame_score_matches_direct_sum(oracle, nc, reads):
    """glimmer-mg.cc:561-604 slices of the 6xL table == running sums in the stated order"""
    import ctypes as C
    indep = oracle.indep(0.39)
    s = reads[5]
    fs = oracle.score_all_frames(nc, indep, s)
    L = len(s)
    for frame, lo, hi in ((1, 10, 400), (-2, 33, 333)):
        out = np.empty(hi - lo)
        oracle.L.orc_cumulative_frame_score(fs.ctypes.data_as(C.POINTER(C.c_double)), L, frame, lo, hi,
                                            out.ctypes.data_as(C.POINTER(C.c_double)))
        cum, f, exp = 0.0, 1, []
        for i in range(hi - lo):
            si = hi - 1 - i if frame > 0 else lo - 1 + i
            cum = cum + fs[f if frame > 0 else 3 + f, si]
            exp.append(cum)
            f = 0 if f == 2 else f + 1
        assert np.array_equal(out, np.array(exp))


def test_character_helpers(oracle):
    for ch, want in zip("acgtACGTrRyYnNxX-", "acgtACGTggcccccc" + "c"):
        assert chr(oracle.L.orc_filter(ord(ch))) == want
    comp = dict(zip("acgtrykmbdhvswn", "tgcayrmkvhdbswn"))
    for a, b in comp.items():
        assert chr(oracle.L.orc_complement(ord(a))) == b
        assert chr(oracle.L.orc_complement(ord(a.upper()))) == b.upper()
    assert [oracle.L.orc_subscript(ord(c)) for c in "acgtRWn"] == [0, 1, 2, 3, 2, 3, 1]
