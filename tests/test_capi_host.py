"""CPU-side checks of the product library (no compute launches): libgmg.so loads, exports every
symbol include/*.h declares, packs bases like the reference's Filter/Subscript, and the host-side
ICM_t (model reader/writer, null-model builder) matches the oracle and the goldens bit for bit.
Also: scoring entry points fail loudly (GMG_ENODEV) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import DATA, GOLD, ROOT


def declared_functions():
    names = []
    for hdr in ("gmg.h", "gmg_icm.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(gmg_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(gmg):
    lib = C.CDLL(gmg.build.LIB)
    decl = declared_functions()
    assert len(decl) >= 35
    for name in decl:
        assert hasattr(lib, name), "libgmg.so lacks %s" % name
    assert set(decl) == set(gmg.capi.PROTOTYPES), "ctypes prototypes out of sync with include/*.h"
    assert b"gfx950" in gmg.capi.lib().gmg_version()


def test_base_code_matches_filter_subscript(gmg, oracle):
    lib = gmg.capi.lib()
    for ch in range(1, 128):
        assert lib.gmg_base_code(ch) == oracle.L.orc_subscript(ch), chr(ch)


def test_pack_bases_layout(gmg):
    packed, off = gmg.api.pack_strings(["acgtacgtacgtacgtTG", "", "nnRy"])
    assert list(off) == [0, 18, 18, 22]
    codes = [(int(packed[g >> 4]) >> (2 * (g & 15))) & 3 for g in range(22)]
    assert codes == [0, 1, 2, 3] * 4 + [3, 2] + [1, 1, 2, 1]


def test_pack_bases_word_at_a_time_equals_base_at_a_time(gmg):
    """gmg_pack_bases takes whole words where it can; any start offset, any length, every byte value"""
    rng = np.random.default_rng(3)
    lib = gmg.capi.lib()
    for first, n in [(0, 0), (0, 1), (5, 11), (15, 1), (15, 2), (16, 16), (3, 100), (31, 257), (7, 4096)]:
        raw = rng.integers(1, 256, size=n, dtype=np.uint8).tobytes()
        words = np.zeros(int(lib.gmg_packed_words(first + n)), np.uint32)
        assert lib.gmg_pack_bases(raw, n, first, words.ctypes.data) == 0
        for i, ch in enumerate(raw):
            g = first + i
            assert (int(words[g >> 4]) >> (2 * (g & 15))) & 3 == lib.gmg_base_code(ch)
        assert all(((int(words[g >> 4]) >> (2 * (g & 15))) & 3) == 0 for g in range(first))


def test_synthetic_stream_matches_splitmix_reference_values(gmg):
    # SplitMix64 with seed 0: first outputs are well known
    packed, off = gmg.synth.packed_reads(2, 32, 0)
    z = packed[:4].view(np.uint64)
    assert int(z[0]) == 0xE220A8397B1DCDAF and int(z[1]) == 0x6E789E6AA1B965F4
    s = gmg.synth.unpack_ascii(packed, 0, 64)
    assert len(s) == 64 and set(s) <= set(b"acgt")
    assert list(off) == [0, 32, 64]


def test_host_icm_reader_matches_oracle(gmg, oracle):
    for name in ("NC_000915.icm", "cluster-4.icm", "seqs.cluster-4.run1.filt.gicm"):
        path = os.path.join(DATA, name)
        icm = gmg.Icm.open(path)
        om = oracle.read(path)
        c = om.contents
        assert icm.params == (c.model_len, c.model_depth, c.periodicity, c.num_nodes)
        mip, prob = icm.tables()
        omip, oprob = oracle.tables(om)
        assert np.array_equal(mip, omip)
        assert np.array_equal(prob.view(np.uint32), oprob.view(np.uint32))


def test_host_icm_writer_round_trip(gmg, tmp_path):
    src = os.path.join(DATA, "NC_000915.icm")
    out = tmp_path / "copy.icm"
    gmg.Icm.open(src).write(out)
    assert out.read_bytes() == open(src, "rb").read()


def test_host_null_model_matches_golden(gmg):
    g = np.load(os.path.join(GOLD, "indep.npz"))
    keys = sorted(k[:-5] for k in g.files if k.endswith("_prob"))
    for key in keys:
        gc = float(key[2:key.index("_")])
        stops = tuple(key[key.index("_") + 1:].split("-"))
        mip, prob = gmg.Icm.indep(gc, stops).tables()
        assert np.array_equal(prob.view(np.uint32), g[key + "_prob"].view(np.uint32)), key
        present = g[key + "_mip"] != -2
        assert np.array_equal(mip[present], g[key + "_mip"][present])


def test_host_icm_error_paths(gmg, tmp_path):
    with pytest.raises(gmg.GmgError, match="Could not open"):
        gmg.Icm.open(tmp_path / "missing.icm")
    bad = tmp_path / "short.icm"
    bad.write_bytes(b"x" * 100)
    with pytest.raises(gmg.GmgError, match="ERROR reading ICM header"):
        gmg.Icm.open(bad)
    blob = bytearray(open(os.path.join(DATA, "cluster-4.icm"), "rb").read())
    blob[150:154] = (199).to_bytes(4, "little")
    bad.write_bytes(bytes(blob))
    with pytest.raises(gmg.GmgError, match="Bad ICM version = 199  should be 200"):
        gmg.Icm.open(bad)
    with pytest.raises(gmg.GmgError, match="Incompatible"):
        m = gmg.Icm.new(12, 7, 3)
        arr = (C.c_char_p * 1)(b"taa")
        gmg.api._ck(gmg.capi.lib().gmg_icm_build_indep(m.h, 0.5, arr, 1))


def test_no_gpu_means_loud_failure_not_fallback(gmg):
    """On a box without a GPU every device entry point must refuse (there is no CPU path)."""
    if gmg.capi.lib().gmg_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(gmg.GmgError) as e:
        gmg.init(0)
    assert e.value.code == -2
    with pytest.raises(gmg.GmgError):
        gmg.Reads.from_strings(["acgt"])
    with pytest.raises(gmg.GmgError):
        gmg.Icm.open(os.path.join(DATA, "cluster-4.icm")).device()
    with pytest.raises(gmg.GmgError) as e:                      # training counts on the device or not at all
        gmg.Icm.train([b"acgtacgtacgtacgtacgtacgt"], 12, 2, 3)
    assert "no CPU fallback" in str(e.value)


def test_fasta_split_cuts_only_where_a_record_starts(gmg):
    """gmg_fasta_split is host code: every cut is a '>' directly behind a newline, the pieces tile the file"""
    import ctypes as C
    import numpy as np
    lib = gmg.capi.lib()
    data = b"junk\n" + b"".join(b">r%d has > inside\nACGT>mid%d\nAC\nGT\n\n" % (i, i) for i in range(200))
    cuts = np.zeros(64, np.uint64)
    n = lib.gmg_fasta_split(data, len(data), 700, cuts.ctypes.data_as(C.c_void_p), 63)
    assert 5 < n <= 63 and cuts[0] == 0 and cuts[n] == len(data)
    for c in cuts[1:n]:
        assert data[int(c) - 1:int(c) + 1] == b"\n>"
    assert np.all(np.diff(cuts[:n + 1].astype(np.int64)) > 0)
    assert lib.gmg_fasta_split(data, len(data), 10**9, cuts.ctypes.data_as(C.c_void_p), 63) == 1 and cuts[1] == len(data)


def test_build_reverse_codon_wo_stops_tables_without_a_device():
    """ICM_t::Build_Reverse_Codon_WO_Stops (src/ICM/icm.cc:219-350; host/icm.cc) is host arithmetic: the drop-in build of
    ref_dump writes the tables the reference's class wrote into tests/golden/revcodon.npz (the 72 bytes behind them are
    Score_String values, which need the device: tests/test_gpu_icm_class.py)"""
    import subprocess
    from conftest import built_binary
    exe = built_binary("integration", "_build", "ref_dump_dropin")
    g = np.load(os.path.join(GOLD, "revcodon.npz"))
    for key, seed, stops in (("s1_taa_tag_tga", 1, "taa,tag,tga"), ("s20260105_taa_tag", 20260105, "taa,tag"), ("s7_tga", 7, "tga")):
        res = subprocess.run([exe, "revcodon", str(seed), stops, "model"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert res.returncode == 0, res.stderr.decode()[-500:]
        want = g[key].tobytes()
        assert len(res.stdout) == len(want) - 72 and res.stdout == want[:-72]
