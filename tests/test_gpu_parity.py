"""Parity tests proper: the HIP path, called through the C ABI (include/gmg.h), against
 - the committed golden vectors of the real reference (tests/golden, oracle/gen_golden.py), and
 - the CPU oracle on the same seeded inputs.
Bar: bit-exact.  Every per-base value is an fp32 table entry widened to double; the gene - null
difference is one exact-order double subtraction; sums are sequential double adds in reference
order (SURVEY.md finding 3) -- so the tolerance is zero, well inside north_star's 1e-6 relative.
Run on the GPU box with  -m gpu ."""
import hashlib
import os

import numpy as np
import pytest

from conftest import DATA, GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nc(gpu):
    return gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))


@pytest.fixture(scope="module")
def o_nc(oracle):
    return oracle.read(os.path.join(DATA, "NC_000915.icm"))


@pytest.fixture(scope="module")
def fa_reads(gpu, seqs_fa):
    return gpu.Reads.from_strings(seqs_fa[1])


def test_library_loaded_is_the_in_tree_hip_build(gpu):
    maps = open("/proc/self/maps").read()
    assert "glimmer-mg_amd/lib/libgmg.so" in maps


# ---------------------------------------------------------------- a13 / a7: six-frame scores

def test_frame6_golden_nc(gpu, nc, fa_reads):
    g = np.load(os.path.join(GOLD, "frames_nc.npz"))
    out = gpu.frame_score6(nc, gpu.Icm.indep(float(g["gc"])), fa_reads)
    L = 500
    per_read = out.reshape(6, 999, L).transpose(1, 0, 2)
    n = g["frames"].shape[0]
    assert np.array_equal(per_read[:n], g["frames"])
    assert hashlib.sha256(np.ascontiguousarray(per_read).tobytes()).hexdigest() == str(g["sha256_all"])


def test_frame6_golden_small_model_other_stops(gpu, fa_reads):
    g = np.load(os.path.join(GOLD, "frames_gicm.npz"))
    m = gpu.Icm.open(os.path.join(DATA, "seqs.cluster-4.run1.filt.gicm"))
    out = gpu.frame_score6(m, gpu.Icm.indep(float(g["gc"]), tuple(str(g["stops"]).split(","))), fa_reads)
    n = g["frames"].shape[0]
    assert np.array_equal(out.reshape(6, 999, 500).transpose(1, 0, 2)[:n], g["frames"])


def test_frame6_ragged_and_tiny_reads_vs_oracle(gpu, nc, oracle, o_nc):
    rng = np.random.default_rng(7)
    lens = [0, 1, 2, 3, 10, 11, 12, 13, 22, 23, 24, 0, 63, 64, 65, 500, 1, 1023, 1024, 1025, 2, 0]
    seqs = ["".join(rng.choice(list("acgt"), n)) for n in lens]
    seqs[15] = seqs[15][:100] + "NNNRYKMnnnnryswkmbdhv" + seqs[15][121:]      # Filter() collapses IUPAC
    reads = gpu.Reads.from_strings(seqs)
    out = gpu.frame_score6(nc, gpu.Icm.indep(0.42), reads)
    o_indep = oracle.indep(0.42)
    for r, s in enumerate(seqs):
        lo, hi = int(reads.offsets[r]), int(reads.offsets[r + 1])
        exp = oracle.score_all_frames(o_nc, o_indep, oracle.filter_lower(s))
        assert np.array_equal(out[:, lo:hi], exp), "read %d len %d" % (r, len(s))


def test_frame6_row_stride_does_not_change_the_table(gpu, nc):
    """gmg_frame_score6_strided: the same six rows whatever their distance (odd strides take the one-double-per-lane
    stores, even ones the paired stores; both cover the batch tail and the partial-window heads)."""
    rng = np.random.default_rng(11)
    lens = [int(x) for x in rng.integers(0, 700, 301)]
    reads = gpu.Reads.from_strings(["".join(rng.choice(list("acgt"), n)) for n in lens])
    total = reads.total_bases
    base = gpu.frame_score6(nc, gpu.Icm.indep(0.5), reads)
    for stride in (total, total + 1, total + 2, total + 7, (total + 15) // 16 * 16 + 16):
        assert np.array_equal(gpu.frame_score6(nc, gpu.Icm.indep(0.5), reads, row_stride=stride), base), stride
    with pytest.raises(gpu.GmgError):
        gpu.frame_score6(nc, gpu.Icm.indep(0.5), reads, row_stride=total - 1)


def test_frame6_empty_batch(gpu, nc):
    reads = gpu.Reads.from_strings([])
    assert gpu.frame_score6(nc, gpu.Icm.indep(0.5), reads).shape == (6, 0)
    reads = gpu.Reads.from_strings(["", ""])
    assert gpu.frame_score6(nc, gpu.Icm.indep(0.5), reads).shape == (6, 0)


def test_frame6_synthetic_reads_vs_oracle(gpu, nc, oracle, o_nc):
    n, L, seed = 2000, 500, 20260101
    packed, off = gpu.synth.packed_reads(n, L, seed)
    reads = gpu.Reads(packed, off)
    out = gpu.frame_score6(nc, gpu.Icm.indep(0.5), reads).reshape(6, n, L).transpose(1, 0, 2)
    ascii_all = gpu.synth.unpack_ascii(packed, 0, n * L)
    exp = oracle.score_reads_6frame(o_nc, oracle.indep(0.5), ascii_all, n, L)
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("n,L", [(4000, 500), (3000, 293), (1500, 1000), (700, 2049), (300, 4100), (5000, 400)])
def test_frame6_uniform_batches_vs_oracle(gpu, nc, oracle, o_nc, n, L):
    """uniform batches of several read lengths (read boundaries at every phase of the 2,048-base chunks): sampled reads equal the
    oracle's table, and the call is deterministic"""
    packed, off = gpu.synth.packed_reads(n, L, 77 + L)
    reads = gpu.Reads(packed, off)
    indep = gpu.Icm.indep(0.47)
    a = gpu.frame_score6(nc, indep, reads)
    assert a.tobytes() == gpu.frame_score6(nc, indep, reads).tobytes()
    o_indep = oracle.indep(0.47)
    for r in (0, 1, 7, n // 2, n - 2, n - 1):
        s = gpu.synth.unpack_ascii(packed, r * L, L)
        assert np.array_equal(a[:, r * L:(r + 1) * L], oracle.score_all_frames(o_nc, o_indep, s)), r


def test_options_api(gpu):
    assert gpu.get_option("strings_fused") == 1
    assert (gpu.get_option("mg_fused"), gpu.get_option("mg_err_skip"), gpu.get_option("mg_orfs_events")) == (1, 1, 2)
    with gpu.option("mg_tile", 512):
        assert gpu.get_option("mg_tile") == 512
    assert gpu.get_option("mg_tile") == 0
    with pytest.raises(gpu.GmgError):
        gpu.set_option("no_such_switch", 1)
    with pytest.raises(gpu.GmgError):
        gpu.set_option("diag", 1)                       # the ablation kernels are not in the product build
    with pytest.raises(gpu.GmgError):
        gpu.set_option("mg_max_entries", 2 ** 31)


def test_frame6_generic_shapes_use_same_answers(gpu, oracle):
    """period-1 Phymm-style models are refused (Frame_Score would assert); a (3,2,3) model as the
    'gene' model exercises the non-default shape path."""
    c4 = gpu.Icm.open(os.path.join(DATA, "cluster-4.icm"))
    reads = gpu.Reads.from_strings(["acgtacgtacgtacgtacgt"])
    with pytest.raises(gpu.GmgError):
        gpu.frame_score6(c4, gpu.Icm.indep(0.5), reads)
    rng = np.random.default_rng(3)
    seqs = ["".join(rng.choice(list("acgt"), n)) for n in (40, 7, 300)]
    reads = gpu.Reads.from_strings(seqs)
    out = gpu.frame_score6(gpu.Icm.indep(0.3), gpu.Icm.indep(0.6, ("taa", "tag")), reads)
    a, b = oracle.indep(0.3), oracle.indep(0.6, ("taa", "tag"))
    for r, s in enumerate(seqs):
        lo, hi = int(reads.offsets[r]), int(reads.offsets[r + 1])
        assert np.array_equal(out[:, lo:hi], oracle.score_all_frames(a, b, s))


# ---------------------------------------------------------------- a5 / a6 / a11 / a12: segments

def golden_segments(gpu, fa_reads):
    g = np.load(os.path.join(GOLD, "segs.npz"))
    rows = [(r, lo, ln, gpu.REVERSED if st > 0 else gpu.COMPLEMENTED) for r, lo, ln, st in g["segs"]]
    return g, gpu.Segments(fa_reads, rows)


def test_cumulative_score_golden_orf_buffers(gpu, nc, fa_reads):
    g, segs = golden_segments(gpu, fa_reads)
    assert np.array_equal(gpu.segment_cumscore(nc, fa_reads, segs, 1), g["gene_cum"])
    indep = gpu.Icm.indep(float(g["gc"]))
    assert np.array_equal(gpu.segment_cumscore(indep, fa_reads, segs, 1), g["indep_cum"])


def test_all_frame_score_golden_and_permutation(gpu, nc, fa_reads, oracle, o_nc, seqs_fa):
    g, segs = golden_segments(gpu, fa_reads)
    n = segs.n
    lens = segs.rows[:, 2]
    raw = gpu.all_frame_score(nc, fa_reads, segs, lens, np.full(n, 3, np.int32))
    assert np.array_equal(raw, g["allframe_raw"])
    frames = np.array([(1, 2, 3, -1, -2, -3)[i % 6] for i in range(n)], np.int32)
    prefix = np.maximum(lens.astype(np.int64) - (np.arange(n) % 5), 0).astype(np.uint32)
    got = gpu.all_frame_score(nc, fa_reads, segs, prefix, frames)
    for i in range(n):
        r, lo, ln, orient = (int(x) for x in segs.rows[i])
        buf = oracle.buffer(oracle.filter_lower(seqs_fa[1][r]), lo, ln, orient)
        assert np.array_equal(got[i], oracle.all_frame_score(o_nc, buf, int(prefix[i]), int(frames[i])))


def test_score_string_golden_whole_reads(gpu, nc, fa_reads):
    g = np.load(os.path.join(GOLD, "sstring.npz"))
    rows = [(r, 0, 500, gpu.FORWARD) for r in range(999)]
    segs = gpu.Segments(fa_reads, rows)
    c4 = gpu.Icm.open(os.path.join(DATA, "cluster-4.icm"))
    for f in range(3):
        assert np.array_equal(gpu.score_string(nc, fa_reads, segs, f), g["nc"][:, f])
        assert np.array_equal(gpu.score_string(c4, fa_reads, segs, f), g["cluster4"][:, f])


def test_reference_scores_tmp_through_hip(gpu, fa_reads, seqs_fa):
    """the reference's own committed outputs: icm-N.scores.tmp, Score_String(read, 500, 0)"""
    segs = gpu.Segments(fa_reads, [(r, 0, 500, gpu.FORWARD) for r in range(999)])
    for k in range(6):
        m = gpu.Icm.open(os.path.join(DATA, "cluster-%d.icm" % k))
        got = gpu.score_string(m, fa_reads, segs, 0)
        lines = open(os.path.join(DATA, "icm-%d.scores.tmp" % k)).read().splitlines()
        assert ["%.4f" % v for v in got] == [ln.split("\t")[1].strip() for ln in lines]


def test_segment_kernels_all_orientations_vs_oracle(gpu, nc, oracle, o_nc):
    rng = np.random.default_rng(11)
    seqs = ["".join(rng.choice(list("acgt"), n)) for n in (700, 50, 12, 5)]
    reads = gpu.Reads.from_strings(seqs)
    rows = []
    for r, s in enumerate(seqs):
        for orient in range(4):
            for _ in range(4):
                ln = int(rng.integers(0, len(s) + 1))
                lo = int(rng.integers(0, len(s) - ln + 1))
                rows.append((r, lo, ln, orient))
    segs = gpu.Segments(reads, rows)
    for f in range(3):
        cum = segs.split(gpu.segment_cumscore(nc, reads, segs, f))
        per = segs.split(gpu.segment_frame_score(nc, reads, segs, f))
        tot = gpu.score_string(nc, reads, segs, f)
        part = gpu.segment_partial_prob(nc, reads, segs, f)
        for i, (r, lo, ln, orient) in enumerate(rows):
            buf = oracle.buffer(seqs[r], lo, ln, orient)
            assert np.array_equal(cum[i], oracle.cumulative_score(o_nc, buf, f))
            assert np.array_equal(per[i], oracle.frame_score(o_nc, buf, f))
            assert tot[i] == oracle.score_string(o_nc, buf, f)
            if ln:
                assert part[i] == oracle.partial_window(o_nc, ln - 1, buf, f)


def test_segment_range_and_frame_errors(gpu, nc):
    reads = gpu.Reads.from_strings(["acgtacgtacgt"])
    with pytest.raises(gpu.GmgError):
        gpu.Segments(reads, [(0, 5, 8, gpu.FORWARD)])
    with pytest.raises(gpu.GmgError):
        gpu.Segments(reads, [(1, 0, 1, gpu.FORWARD)])
    segs = gpu.Segments(reads, [(0, 0, 12, gpu.FORWARD)])
    with pytest.raises(gpu.GmgError):          # assert(0 <= frame && frame < periodicity), icm.cc:877
        gpu.score_string(nc, reads, segs, 3)


# ---------------------------------------------------------------- a2 / a3 / a8: windows

def test_full_window_prob_and_distrib_golden(gpu, nc):
    g = np.load(os.path.join(GOLD, "windows.npz"))
    codes = np.searchsorted(np.frombuffer(b"acgt", np.uint8), g["windows"]).astype(np.uint8)
    for f in range(3):
        dist, prob = gpu.window_distrib(nc, codes, np.full(len(codes), f, np.int32))
        assert np.array_equal(prob, g["prob"][:, f])
        assert np.array_equal(dist.view(np.uint32), g["dist"][:, f].view(np.uint32))


def test_partial_window_prob_golden(gpu, nc, fa_reads):
    g = np.load(os.path.join(GOLD, "partial.npz"))
    n = g["nc"].shape[0]
    rows = [(r, 0, i + 1, gpu.FORWARD) for r in range(n) for i in range(11)]
    segs = gpu.Segments(fa_reads, rows)
    for f in range(3):
        assert np.array_equal(gpu.segment_partial_prob(nc, fa_reads, segs, f).reshape(n, 11), g["nc"][:, f])
    c4 = gpu.Icm.open(os.path.join(DATA, "cluster-4.icm"))
    assert np.array_equal(gpu.segment_partial_prob(c4, fa_reads, segs, 0).reshape(n, 11), g["cluster4"][:, 0])


# ---------------------------------------------------------------- size-independent properties at full size

def test_frame6_full_size_properties(gpu, nc, oracle, o_nc):
    """BASELINE configs[1] shape (1M x 500 bp is 24 GB of output; here 100k x 500 bp = 2.4 GB, same
    code path).  Properties: (1) determinism: two launches give identical bits; (2) locality: a read's
    scores do not depend on its neighbours -- any read re-scored alone matches its slice; (3) sampled
    reads equal the oracle; (4) strand symmetry: scoring the reverse complement of a read swaps and
    reverses rows 0-2 <-> 3-5."""
    n, L, seed = 100_000, 500, 99
    packed, off = gpu.synth.packed_reads(n, L, seed)
    reads = gpu.Reads(packed, off)
    indep = gpu.Icm.indep(0.5)
    a = gpu.frame_score6(nc, indep, reads)
    b = gpu.frame_score6(nc, indep, reads)
    assert np.array_equal(a, b)
    del b
    o_indep = oracle.indep(0.5)
    rng = np.random.default_rng(5)
    for r in [0, 1, n - 1] + list(rng.integers(0, n, 20)):
        s = gpu.synth.unpack_ascii(packed, r * L, L)
        assert np.array_equal(a[:, r * L:(r + 1) * L], oracle.score_all_frames(o_nc, o_indep, s))
    r = 12345
    s = gpu.synth.unpack_ascii(packed, r * L, L).decode()
    rc = s[::-1].translate(str.maketrans("acgt", "tgca"))
    one = gpu.frame_score6(nc, indep, gpu.Reads.from_strings([s, rc]))
    assert np.array_equal(one[:, :L], a[:, r * L:(r + 1) * L])
    assert np.array_equal(one[0:3, L:][:, ::-1], one[3:6, :L])
    assert np.array_equal(one[3:6, L:][:, ::-1], one[0:3, :L])


def test_frame6_true_size_every_row_beyond_4_gib(gpu, nc, oracle, o_nc):
    """BASELINE configs[1] at its TRUE size: 1M x 500 bp, the 24 GB fp64 table on the device (rows of 4.0e9 bytes: row 1 crosses byte
    offset 2^32 inside read 73,741, row 4 crosses element index 2^31 inside read 294,967; rows 2 - 5 lie beyond 2^32 bytes
    altogether).  Reads from the head, the middle, the tail and on both sides of those crossings, all six rows, against the oracle."""
    import ctypes as C
    n, L, seed = 1_000_000, 500, 20260101
    packed, off = gpu.synth.packed_reads(n, L, seed)
    reads = gpu.Reads(packed, off)
    total = n * L
    indep, o_indep = gpu.Icm.indep(0.5), oracle.indep(0.5)
    buf = gpu.api._DeviceBuffer(6 * total * 8)
    gpu.frame_score6(nc, indep, reads, d_out=buf.ptr.value)
    lib = gpu.capi.lib()
    gpu.api._ck(lib.gmg_synchronize(None))
    cross_bytes = (2**32 - total * 8) // 8 // L           # read of row 1 that holds byte offset 2^32
    cross_elems = (2**31 - 4 * total) // L                # read of row 4 that holds element index 2^31
    assert (cross_bytes, cross_elems) == (73_741, 294_967)
    rng = np.random.default_rng(11)
    picks = [0, 1, 2, n // 2 - 1, n // 2, n - 2, n - 1, cross_bytes - 1, cross_bytes, cross_bytes + 1, cross_elems - 1, cross_elems, cross_elems + 1]
    picks += [int(x) for x in rng.integers(0, n, 12)]
    row = np.empty(L, np.float64)
    for r in picks:
        want = oracle.score_all_frames(o_nc, o_indep, gpu.synth.unpack_ascii(packed, r * L, L))
        for f in range(6):
            gpu.api._ck(lib.gmg_memcpy_d2h(row.ctypes.data_as(C.c_void_p), C.c_void_p(buf.ptr.value + (f * total + r * L) * 8), L * 8, None))
            gpu.api._ck(lib.gmg_synchronize(None))
            assert np.array_equal(row, want[f]), (r, f)
    buf.free()


# ---------------------------------------------------------------- a12: Score_Orfs inner loop

ORF_PATHS = {"events": 0, "exact": 1, "fused": 2, "events-walk64": 0, "events-dense": 0, "events-prefetch": 0, "events-sparse": 0}    # option orfs_exact_path (gmg_orfs.hip); events-walk64: the
# running sums by k_orf_walk_sums (option orfs_walk8 = 0), events-dense: by k_orf_walk_sums8 with every base written (= 2), events-sparse: only where
# k_orf_events can ask (= 1), events-prefetch (= 3), instead of the compact form (= 4, the default: those values back to back per unit).  orfs_q_poison: the array is NaNs before the sums are written -- a read of an unwritten entry shows.
ORF_WALK = {"events-walk64": 0, "events-dense": 2, "events-prefetch": 3, "events-sparse": 1}


@pytest.mark.parametrize("name,kw", [("orfs_default", {}), ("orfs_X", {"allow_truncated": True}),
                                     ("orfs_g90_first", {"min_gene_len": 90, "use_first_start": True})])
@pytest.mark.parametrize("path", sorted(ORF_PATHS))
def test_score_orfs_golden_start_lists(gpu, nc, fa_reads, name, kw, path, request_finalizers):
    """ORFs from the reference's Find_Orfs on seqs.fa; start lists, gene score and gene length must equal
    what the reference's Score_Orfs handed to Add_Events_* (tests/golden/orfs_*.npz), bit for bit --
    through the events path (running sums per strand and class, an ORF visits its start codons only), the fused path
    (gene-only six-frame pass + k_orf_fused: one lane walks the ORF) and the exact any-model path."""
    gpu.set_option("orfs_exact_path", ORF_PATHS[path])
    gpu.set_option("orfs_walk8", ORF_WALK.get(path, 4))
    gpu.set_option("orfs_q_poison", 1)
    request_finalizers.append(lambda: (gpu.set_option("orfs_exact_path", 0), gpu.set_option("orfs_walk8", 4), gpu.set_option("orfs_q_poison", 0)))
    g = np.load(os.path.join(GOLD, name + ".npz"))
    gc = float(np.load(os.path.join(GOLD, "frames_nc.npz"))["gc"])
    res, starts = gpu.score_orfs(nc, gpu.Icm.indep(gc), fa_reads, g["orfs"], **kw)
    accepted = np.zeros(len(res), bool)
    accepted[g["gene_orf"]] = True
    assert np.array_equal(res["is_tentative_gene"] != 0, accepted)
    sel = res[g["gene_orf"]]
    assert np.array_equal(sel["gene_score"], g["gene_score"])
    assert np.array_equal(sel["best_j"] + 1, g["gene_len"])
    assert np.array_equal(sel["n_starts"], g["gene_nstarts"])
    for r, b, cnt in zip(sel, g["gene_start_begin"], g["gene_nstarts"]):
        st = starts[r["start_begin"]:r["start_begin"] + r["n_starts"]]
        assert np.array_equal(st["score"], g["start_score"][b:b + cnt])
        got = np.stack([st["j"], st["pos"], st["which"], st["truncated"], st["first"]], 1)
        assert np.array_equal(got, g["start_int"][b:b + cnt])


@pytest.mark.parametrize("path", sorted(ORF_PATHS))
def test_score_orfs_random_orfs_on_ragged_reads_vs_oracle(gpu, nc, oracle, o_nc, path, request_finalizers):
    """random in-range ORFs (both strands, lengths 3..read length, also not a multiple of 3) on reads of ragged lengths:
    every field of every start and of the per-ORF result must equal the oracle's Score_Orfs restatement"""
    gpu.set_option("orfs_exact_path", ORF_PATHS[path])
    gpu.set_option("orfs_walk8", ORF_WALK.get(path, 4))
    gpu.set_option("orfs_q_poison", 1)
    request_finalizers.append(lambda: (gpu.set_option("orfs_exact_path", 0), gpu.set_option("orfs_walk8", 4), gpu.set_option("orfs_q_poison", 0)))
    rng = np.random.default_rng(77)
    lengths = [int(x) for x in rng.integers(40, 900, size=120)] + [12, 13, 30, 500, 2100]
    seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]
    reads = gpu.Reads.from_strings(seqs)
    rows = []
    for r, n in enumerate(lengths):
        for _ in range(6):
            ln = int(rng.integers(3, n + 1))
            if rng.random() < 0.8:
                ln -= ln % 3
            lo = int(rng.integers(0, n - ln + 1))
            if rng.random() < 0.5:
                rows.append((r, 1 + lo % 3, lo + ln + 1, ln))        # forward: hi = stop - 1, lo = hi - len
            else:
                rows.append((r, -1 - lo % 3, lo - 2, ln))            # reverse: lo = stop + 2
    rows = np.array(rows)
    kw = dict(min_gene_len=30, allow_truncated=True, ignore_score_len=200)
    res, starts = gpu.score_orfs(nc, gpu.Icm.indep(0.4), reads, rows, **kw)
    o_indep = oracle.indep(0.4)
    prm = oracle.orf_params(**kw)
    n_genes = 0
    for (r, frame, stop, ln), got in zip(rows, res):
        n, out, want = oracle.score_orf(o_nc, o_indep, seqs[r], int(frame), int(stop), int(ln), prm)
        assert (got["first_j"], got["best_j"], got["best_pos"], got["orf_is_truncated"]) == \
               (out.first_j, out.best_j, out.best_pos, out.orf_is_truncated)
        assert got["best_score"] == out.best_score
        if n < 0:
            assert got["n_starts"] == 0 and not got["is_tentative_gene"]
            continue
        assert got["n_starts"] == n and bool(got["is_tentative_gene"]) == bool(out.is_tentative_gene)
        assert got["gene_score"] == out.gene_score or (np.isnan(got["gene_score"]) and np.isnan(out.gene_score))
        st = starts[got["start_begin"]:got["start_begin"] + n]
        assert [(s["j"], s["pos"], s["which"], s["truncated"], s["first"], s["score"]) for s in st] == \
               [(w.j, w.pos, w.which, w.truncated, w.first, w.score) for w in want]
        n_genes += int(out.is_tentative_gene)
    assert n_genes > 20


def test_score_orfs_rejects_wrapping_orfs(gpu, nc, fa_reads):
    with pytest.raises(gpu.GmgError):
        gpu.score_orfs(nc, gpu.Icm.indep(0.5), fa_reads, np.array([[0, 1, 30, 90]]))     # lo < 0: circular wrap
    with pytest.raises(gpu.GmgError):
        gpu.score_orfs(nc, gpu.Icm.indep(0.5), fa_reads, np.array([[0, -1, 450, 90]]))   # hi > L


def test_score_orfs_full_size_properties(gpu, nc, oracle, o_nc, request_finalizers):
    """BASELINE configs[1] shape on the glimmer3 side: 1 M x 500 bp, every ORF Find_Orfs gives (gmg_find_orfs, ~5 M of them)
    through gmg_orfs_upload + gmg_score_orfs on the events path.  Properties: (1) determinism: two calls, the same bytes;
    (2) bookkeeping: start lists back to back in ORF order; (3) the three device paths agree on a 40,000-read slice; (4) sampled
    ORFs equal the oracle's Score_Orfs; (5) a batch beyond the 32-bit index fields is refused with GMG_ETOOBIG, not wrapped."""
    n, L = 1_000_000, 500
    packed, off = gpu.synth.packed_reads(n, L, 11)
    reads = gpu.Reads(packed, off)
    found, first = gpu.find_orfs(reads, min_gene_len=90)
    assert len(found) > 3 * n
    rows = np.stack([found["read"].astype(np.int64), found["frame"], found["stop_position"], found["orf_len"]], 1)
    indep = gpu.Icm.indep(0.5)
    kw = dict(min_gene_len=90)
    res, starts = gpu.score_orfs(nc, indep, reads, rows, **kw)
    res2, starts2 = gpu.score_orfs(nc, indep, reads, rows, **kw)
    total = int(res["start_begin"][-1]) + int(res["n_starts"][-1])
    assert res.tobytes() == res2.tobytes() and starts[:total].tobytes() == starts2[:total].tobytes()
    del res2, starts2
    assert int(res["n_starts"].sum(dtype=np.uint64)) == total > n
    assert np.array_equal(res["start_begin"], np.concatenate([[0], np.cumsum(res["n_starts"], dtype=np.uint64)[:-1]]).astype(np.uint32))
    # (3) a slice through all three paths
    m = 40_000
    sl_reads = gpu.Reads(*gpu.synth.packed_reads(m, L, 11))
    sl_rows = rows[:int(first[m])]
    got = {}
    for name, opt in ORF_PATHS.items():
        gpu.set_option("orfs_exact_path", opt)
        gpu.set_option("orfs_walk8", ORF_WALK.get(name, 4))
        gpu.set_option("orfs_q_poison", 1)
        request_finalizers.append(lambda: (gpu.set_option("orfs_exact_path", 0), gpu.set_option("orfs_walk8", 4), gpu.set_option("orfs_q_poison", 0)))
        r_, s_ = gpu.score_orfs(nc, indep, sl_reads, sl_rows, **kw)
        got[name] = (r_.tobytes(), s_[:int(r_["start_begin"][-1]) + int(r_["n_starts"][-1])].tobytes())
    gpu.set_option("orfs_exact_path", 0)
    gpu.set_option("orfs_walk8", 4)
    gpu.set_option("orfs_q_poison", 0)
    assert got["events"] == got["fused"] == got["exact"] == got["events-walk64"] == got["events-dense"] == got["events-prefetch"] == got["events-sparse"]
    assert got["events"][0] == res[:len(sl_rows)].tobytes()                 # ... and the slice of the big batch
    # (4) the oracle on sampled ORFs
    o_indep = oracle.indep(0.5)
    prm = oracle.orf_params(**kw)
    rng = np.random.default_rng(5)
    for i in [0, 1, len(rows) - 1] + [int(x) for x in rng.integers(0, len(rows), 300)]:
        r, frame, stop, ln = (int(x) for x in rows[i])
        seq = gpu.synth.unpack_ascii(packed, r * L, L)
        cnt, out, want = oracle.score_orf(o_nc, o_indep, seq, frame, stop, ln, prm)
        g_ = res[i]
        assert (g_["first_j"], g_["best_j"], g_["best_pos"], g_["best_score"]) == (out.first_j, out.best_j, out.best_pos, out.best_score)
        if cnt < 0:
            assert g_["n_starts"] == 0
            continue
        st = starts[g_["start_begin"]:g_["start_begin"] + g_["n_starts"]]
        assert [(s["j"], s["pos"], s["which"], s["truncated"], s["first"], s["score"]) for s in st] == \
               [(w.j, w.pos, w.which, w.truncated, w.first, w.score) for w in want]
    # (5) the guard: with the limit lowered the same upload is refused, never wrapped
    gpu.set_option("mg_max_entries", 1_000_000)
    request_finalizers.append(lambda: gpu.set_option("mg_max_entries", 0x7ffffffe))
    with pytest.raises(gpu.GmgError) as e:
        gpu.score_orfs(nc, indep, reads, rows, **kw)
    assert e.value.code == -7                                                 # GMG_ETOOBIG
