"""Drop-in check on the GPU: the reference's own glimmer3 / glimmer-mg sources, compiled unchanged
against OUR ICM_t (glimmer-mg_amd/host/icm.hh) and linked with libgmg.so (integration/Makefile,
built in the build container; the binaries travel in integration/_build/), must write .predict files
byte-identical to the goldens written by the all-reference build.  Every score in these runs comes from
the HIP layer (one launch per ICM_t call)."""
import os
import subprocess

import pytest

from conftest import DATA, GOLD, ROOT, built_binary

pytestmark = pytest.mark.gpu

CASES = [
    ("glimmer3_dropin", [], "glimmer3.default.predict"),
    ("glimmer3_dropin", ["-X"], "glimmer3.X.predict"),
    ("glimmer-mg_dropin", [], "glimmer-mg.default.predict"),
]


@pytest.mark.parametrize("binary,flags,golden", CASES)
def test_reference_cli_on_our_icm_is_byte_identical(gpu, tmp_path, binary, flags, golden):
    exe = built_binary("integration", "_build", binary)
    tag = str(tmp_path / "out")
    cmd = [exe, *flags, "-m", os.path.join(DATA, "NC_000915.icm"), os.path.join(DATA, "seqs.fa"), tag]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    got = open(tag + ".predict", "rb").read()
    want = open(os.path.join(GOLD, "predict", golden), "rb").read()
    assert got == want
    maps_ok = b"libgmg.so" in subprocess.run(["ldd", exe], stdout=subprocess.PIPE).stdout
    assert maps_ok


@pytest.mark.parametrize("flags,golden", [([], "glimmer3.default.predict"), (["-X"], "glimmer3.X.predict")])
def test_glimmer3_with_batched_score_orfs_is_byte_identical(gpu, tmp_path, flags, golden):
    """integration/glimmer3_gpu: glimmer3's own Add_Events / Process_Events / Trace_Back around ONE gmg_fasta_ingest, ONE
    gmg_find_orfs and ONE gmg_score_orfs call for all 999 reads."""
    exe = built_binary("integration", "_build", "glimmer3_gpu")
    tag = str(tmp_path / "out")
    cmd = [exe, *flags, "-m", os.path.join(DATA, "NC_000915.icm"), os.path.join(DATA, "seqs.fa"), tag]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert open(tag + ".predict", "rb").read() == open(os.path.join(GOLD, "predict", golden), "rb").read()


@pytest.mark.parametrize("ref,dev", [("glimmer3", "glimmer3_gpu"), ("glimmer-mg", "glimmer-mg_gpu")])
def test_long_and_tiny_sequences_through_the_device_front_halves(gpu, tmp_path, ref, dev):
    """one 30 kb sequence, one of 10 bases, one of 5 kb, an empty record: the all-reference CLI and the device front half
    (ingest + Find_Orfs + scoring on the GPU, the reference's events / DP / trace-back on the host) must write the same bytes.
    (The reference's glimmer-mg aborts on an empty record -- assertion in Complement_Transfer, glimmer_base.cc:422 -- so
    that case is left to glimmer3.)"""
    import numpy as np
    exe_ref, exe_dev = built_binary("oracle", "_ref", ref), built_binary("integration", "_build", dev)
    rng = np.random.default_rng(8)
    fa = tmp_path / "mixed.fa"
    with open(fa, "w") as f:
        for name, n in (("long", 30000), ("tiny", 10), ("empty", 0), ("mid", 5000)):
            if n == 0 and ref == "glimmer-mg":
                continue
            s = "".join("acgt"[c] for c in rng.integers(0, 4, size=n))
            f.write(">%s some description\n" % name)
            for i in range(0, n, 60):
                f.write(s[i:i + 60] + "\n")
    icm = os.path.join(DATA, "NC_000915.icm")
    out = []
    for exe, extra, tag in ((exe_ref, [], "a"), (exe_dev, [], "b")):
        res = subprocess.run([exe, *extra, "-m", icm, str(fa), str(tmp_path / tag)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        out.append(open(str(tmp_path / tag) + ".predict", "rb").read())
    assert out[0] == out[1] and out[0].count(b"orf") >= 1


def _genome():
    return "".join(line.strip() for line in open(os.path.join(DATA, "NC_000915.fna")) if not line.startswith(">"))


def _write_fasta(path, records):
    with open(path, "w") as f:
        for name, s in records:
            f.write(">%s\n" % name)
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")


@pytest.mark.parametrize("mode", ["separate_input", "orflist", "orflist_wrap", "ignore_regions", "ignore_regions_truncated", "mg_circular"])
def test_modes_outside_the_default_loop(gpu, tmp_path, mode):
    """glimmer3 -M (every input sequence is one gene, Score_Separate_Input) and -L (ORFs from a coordinate file, Score_Orflist) are
    batched by glimmer3_gpu: one gmg_score_string call per model for all entries.  -i (ignore regions): Find_Orfs with the regions on
    the device too (gmg_find_orfs, gmg_mg_params.n_ignore_regions), all ORFs scored by ONE gmg_score_orfs call.  Those three must work WITHOUT the drop-in
    binary (the driver is run from a directory that holds nothing else).  glimmer-mg -r (circular genome) and coordinate lists
    with an entry that is no plain segment of the sequence (wrap-around, out of range) are handed to the *_dropin binary beside the
    driver (the reference's own main() on the device-backed ICM_t, started as a child process).  The bytes must be the all-reference
    binary's in every case."""
    import shutil
    import numpy as np
    rng = np.random.default_rng(21)
    icm = os.path.join(DATA, "NC_000915.icm")
    fa = str(tmp_path / "in.fa")
    mg = mode == "mg_circular"
    ref = built_binary("oracle", "_ref", "glimmer-mg" if mg else "glimmer3")
    dev = built_binary("integration", "_build", "glimmer-mg_gpu" if mg else "glimmer3_gpu")
    extra = []
    if mode == "separate_input":
        g = _genome()
        _write_fasta(fa, [("gene%d" % i, g[int(b):int(b) + int(n)]) for i, (b, n) in enumerate(zip(rng.integers(0, 1_600_000, 40), rng.integers(90, 900, 40) // 3 * 3))])
        extra = ["-M"]
    elif mode in ("orflist", "orflist_wrap"):
        _write_fasta(fa, [("chromosome", _genome()[400_000:430_000])])
        # coordinates: the genes the reference predicts on that sequence, as an ORF list (tag, start, stop, direction)
        res = subprocess.run([ref, "-m", icm, fa, str(tmp_path / "pre")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        coords = str(tmp_path / "orfs.coords")
        n_lines = 0
        with open(coords, "w") as f:
            for line in open(str(tmp_path / "pre") + ".predict"):
                if line.startswith("orf"):
                    tag, a, b, frame = line.split()[:4]
                    f.write("%s %s %s %d\n" % (tag, a, b, 1 if int(frame) > 0 else -1))
                    n_lines += 1
            if mode == "orflist_wrap":                  # a forward gene across the end of the sequence: not a plain segment
                f.write("wrap 29701 300 1\n")
        assert n_lines >= 3
        extra = ["-L", coords]
    elif mode == "ignore_regions":
        _write_fasta(fa, [("chromosome", _genome()[400_000:430_000])])
        ign = str(tmp_path / "ignore.txt")
        with open(ign, "w") as f:
            f.write("# lo hi\n2000 3500\n9000 8000\n15000 15100\n")
        extra = ["-i", ign]
    elif mode == "ignore_regions_truncated":            # several sequences (the same regions apply to each), -X, overlapping regions, one at the very start
        g = _genome()
        _write_fasta(fa, [("c1", g[100_000:112_000]), ("c2 second", g[500_000:503_001]), ("c3", g[900_000:900_900])])
        ign = str(tmp_path / "ignore.txt")
        with open(ign, "w") as f:
            f.write("1 40\n700 1300\n1200 1500\n2990 3005\n11000 13000\n")
        extra = ["-X", "-i", ign]
    else:
        g = _genome()
        _write_fasta(fa, [("plasmid", g[700_000:712_000]), ("plasmid2", g[900_000:904_000])])
        extra = ["-r"]
    out = []
    env = dict(os.environ)
    if mode in ("separate_input", "orflist", "ignore_regions", "ignore_regions_truncated"):
        alone = tmp_path / "alone"
        alone.mkdir()
        shutil.copy(dev, str(alone / "glimmer3_gpu"))
        dev = str(alone / "glimmer3_gpu")
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "glimmer-mg_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    for exe, tag in ((ref, "a"), (dev, "b")):
        res = subprocess.run([exe, *extra, "-m", icm, fa, str(tmp_path / tag)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        if mode == "orflist_wrap":
            # the reference itself cannot do this (Sequence_Len is not set in the -L branch: Reverse_Transfer's assertion, glimmer_base.cc:2517);
            # the driver must end the same way -- through the drop-in, not with a made-up score
            assert res.returncode != 0 and b"Reverse_Transfer" in res.stderr
            continue
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        out.append(open(str(tmp_path / tag) + ".predict", "rb").read())
    if mode != "orflist_wrap":
        assert out[0] == out[1] and out[0].count(b"\n") >= 4


def test_separate_input_batched_at_scale(gpu, tmp_path):
    """glimmer3 -M on 20,000 sequences of 300 - 900 bases (each one ORF): glimmer3_gpu = ONE ingest + two gmg_score_string calls;
    byte-identical to the reference binary (measured 20x and more end to end: the reference spends its time in eight Score_String /
    Cumulative_Score passes per sequence)"""
    import time
    import numpy as np
    rng = np.random.default_rng(5)
    g = _genome()
    fa = str(tmp_path / "genes.fa")
    starts, lens = rng.integers(0, 1_600_000, 20_000), rng.integers(300, 900, 20_000) // 3 * 3
    _write_fasta(fa, [("g%d extra words" % i if i % 3 else "", g[int(b):int(b) + int(n)]) for i, (b, n) in enumerate(zip(starts, lens))])
    icm = os.path.join(DATA, "NC_000915.icm")
    ref, dev = built_binary("oracle", "_ref", "glimmer3"), built_binary("integration", "_build", "glimmer3_gpu")
    out, secs = [], []
    for exe, tag in ((ref, "a"), (dev, "b")):
        t0 = time.perf_counter()
        res = subprocess.run([exe, "-M", "-m", icm, fa, str(tmp_path / tag)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        secs.append(time.perf_counter() - t0)
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        out.append(open(str(tmp_path / tag) + ".predict", "rb").read())
    assert out[0] == out[1] and out[0].count(b"\n") == 20_000
    # (wall clock of one un-warmed run each, on a box shared with other jobs: reported, not asserted -- tests/bench/bench_cli.py times it)
    print("glimmer3 -M, 20,000 sequences: reference %.2f s, glimmer3_gpu %.2f s (%.1fx)" % (secs[0], secs[1], secs[0] / secs[1]))


G3_OPTION_SETS = [
    ["-A", "atg,gtg"], ["-C", "38.5"], ["-f", "x"], ["-g", "60", "-o", "20"], ["-q", "150"], ["-t", "40"], ["-z", "4"], ["-Z", "taa,tga"],
    ["-b", os.path.join(DATA, "seqs.cluster-2.run1.filt.motif")],
    ["-X", "-g", "90", "-A", "atg", "-t", "20"], ["-u", "1.5", "-P", "0.6,0.3,0.1"], ["-n"], ["-l"],
]       # (the reference's getopt string gives -f an argument it never reads and -F none although it reads one: -F cannot be used)


@pytest.mark.parametrize("opts", G3_OPTION_SETS, ids=["_".join(o[:2]).replace(",", "").replace("/", "")[:20] for o in G3_OPTION_SETS])
def test_glimmer3_gpu_options_against_the_reference_run_here(gpu, tmp_path, opts):
    """every glimmer3 option that changes the ORFs, the scoring or the weights of the DP (start / stop codon sets, GC, first-start rule,
    lengths, overlap, Ignore_Score_Len, threshold, RBS matrix, feature file, prior): oracle/_ref/glimmer3 run in the test against
    glimmer3_gpu on the 999 sample reads"""
    ref, dev = built_binary("oracle", "_ref", "glimmer3"), built_binary("integration", "_build", "glimmer3_gpu")
    out = []
    for exe, tag in ((ref, "a"), (dev, "b")):
        res = subprocess.run([exe, *opts, "-m", os.path.join(DATA, "NC_000915.icm"), os.path.join(DATA, "seqs.fa"), str(tmp_path / tag)],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert res.returncode == 0, (opts, res.stderr.decode()[-2000:])
        out.append(open(str(tmp_path / tag) + ".predict", "rb").read())
    assert out[0] == out[1] and out[0].count(b">") == 999


def test_glimmer3_gpu_on_the_nasty_fasta_file(gpu, tmp_path):
    """junk before the first record, empty records, '>' inside lines, IUPAC letters, digits, CR LF, tabs: the headers and the filtered
    sequences glimmer3 sees (Fasta_Read + tolower (Filter ())) come from gmg_fasta_ingest in glimmer3_gpu -- same .predict bytes"""
    ref, dev = built_binary("oracle", "_ref", "glimmer3"), built_binary("integration", "_build", "glimmer3_gpu")
    out = []
    for exe, tag in ((ref, "a"), (dev, "b")):
        # (default Min_Gene_Len: with -g 30 the reference itself never returns on this file)
        res = subprocess.run([exe, "-m", os.path.join(DATA, "NC_000915.icm"), os.path.join(DATA, "nasty.fa"), str(tmp_path / tag)],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        out.append(open(str(tmp_path / tag) + ".predict", "rb").read())
    assert out[0] == out[1] and out[0].count(b">") >= 10


# ---- the whole path at genome scale --------------------------------------------------------------------------------------
# One 1.67 Mbp sequence = one "read" that spans > 800 tiles of the six-frame pass, ORFs of several kb, ~100 k ORFs in ONE
# record list: where a tile / offset bug of gmg_find_orfs / gmg_score_orfs (events path: Q[p_j] - Q[p_12] over long ORFs)
# would hide.  tests/golden/predict/NC_000915.run1.predict is the REFERENCE'S OWN committed answer file
# (sample-run/glimmer3/results, made by scripts/g3-iterated.py:58); the other three goldens are oracle/_ref runs of
# oracle/gen_golden.py, and the reference binary is run again beside the device path in the test.

GENOME = os.path.join(DATA, "NC_000915.fna")
RUN1 = ["-u", "-12", "-m", os.path.join(DATA, "NC_000915.icm")]
STEP6 = ["-b", os.path.join(DATA, "NC_000915.run1.motif"), "-m", os.path.join(DATA, "NC_000915.run1.gicm")]


def _predict(exe, opts, tag, timeout=900):
    res = subprocess.run([exe, *opts, GENOME, tag], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    return open(tag + ".predict", "rb").read()


@pytest.mark.parametrize("binary", ["glimmer3_gpu", "glimmer3_dropin"])
def test_genome_run1_is_the_answer_file_the_reference_holds(gpu, tmp_path, binary):
    """glimmer3 -u -12 -m NC_000915.icm NC_000915.fna (g3-iterated.py step 4): 1,549 genes, byte for byte"""
    want = open(os.path.join(GOLD, "predict", "NC_000915.run1.predict"), "rb").read()
    got = _predict(built_binary("integration", "_build", binary), RUN1, str(tmp_path / "run1"))
    assert want.count(b"\norf") == 1549
    assert got == want


@pytest.mark.parametrize("name,ref,dev,opts", [
    ("NC_000915.step6", "glimmer3", "glimmer3_gpu", STEP6),                     # g3-iterated.py:74 without -f (-f ignores its file)
    ("NC_000915.glimmer-mg", "glimmer-mg", "glimmer-mg_gpu", RUN1[2:]),
    ("NC_000915.glimmer3.X_l", "glimmer3", "glimmer3_gpu", ["-X", "-l"] + RUN1[2:]),
], ids=["step6", "glimmer-mg", "glimmer3_X_l"])
def test_genome_other_shapes_against_the_reference_run_here(gpu, tmp_path, name, ref, dev, opts):
    want = open(os.path.join(GOLD, "predict", name + ".predict"), "rb").read()
    again = _predict(built_binary("oracle", "_ref", ref), opts, str(tmp_path / "ref"))
    assert again == want and want.count(b"\norf") >= 2000
    assert _predict(built_binary("integration", "_build", dev), opts, str(tmp_path / "dev")) == want
