"""Thin object wrappers over the C ABI (include/gmg.h, include/gmg_icm.h) for tests and bench.py.

Names follow the reference: an Icm is an ICM_t (src/ICM/icm.hh:116-180), reads are the sequences
glimmer3 / glimmer-mg score, a segment is an ORF-style scoring buffer cut from a read.
Nothing here computes a score: every function forwards to libgmg.so and raises GmgError with the
library's message when a call fails (including "no GPU").
"""
import ctypes as C
import numpy as np

from . import capi

FORWARD, REVERSED, COMPLEMENTED, REVCOMP = 0, 1, 2, 3
DEFAULT_STOPS = ("taa", "tag", "tga")     # src/Glimmer/glimmer_base.cc Set_Start_And_Stop_Codons defaults


class GmgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gmg status %d: %s" % (code, msg))
        self.code = code


def _ck(rc):
    if rc != 0:
        raise GmgError(rc, capi.lib().gmg_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def init(device=0):
    """gmg_init: bind this process to one GPU.  Raises GmgError (GMG_ENODEV) without a gfx950 device."""
    _ck(capi.lib().gmg_init(int(device)))


def set_option(key, value):
    """gmg_set_option: tuning / test switches (include/gmg.h)"""
    _ck(capi.lib().gmg_set_option(key.encode(), int(value)))


def get_option(key):
    v = C.c_longlong()
    _ck(capi.lib().gmg_get_option(key.encode(), C.byref(v)))
    return v.value


class option:
    """with gmg.option("mg_err_flat", 1): ...   -- sets a switch for the block and restores it"""

    def __init__(self, key, value):
        self.key, self.value = key, value

    def __enter__(self):
        self.old = get_option(self.key)
        set_option(self.key, self.value)

    def __exit__(self, *exc):
        set_option(self.key, self.old)


def read_fasta(path):
    """Host-side ingest with the reference's semantics (src/Common/fasta.cc:236-286 Fasta_Read):
    header = text after '>' (leading blanks skipped) up to the newline; sequence = every
    non-whitespace character up to the next '>'.  Returns (headers, sequences) as str lists."""
    hdrs, seqs = [], []
    with open(path, "rb") as fh:
        data = fh.read()
    pos = data.find(b">")
    while pos >= 0:
        pos += 1
        while pos < len(data) and data[pos:pos + 1] == b" ":
            pos += 1
        eol = data.find(b"\n", pos)
        if eol < 0:
            eol = len(data)
        nxt = data.find(b">", eol)
        body = data[eol:nxt if nxt >= 0 else len(data)]
        hdrs.append(data[pos:eol].decode("latin-1"))
        seqs.append(b"".join(body.split()).decode("latin-1"))
        pos = nxt
    return hdrs, seqs


class _DeviceBuffer:
    """device scratch from gmg_device_malloc, copied back with gmg_memcpy_d2h"""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        _ck(capi.lib().gmg_device_malloc(C.byref(self.ptr), self.nbytes))

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(max(arr.nbytes, 1))
        if arr.nbytes:
            _ck(capi.lib().gmg_memcpy_h2d(buf.ptr, _ptr(arr), arr.nbytes, None))
        return buf

    def to_host(self, dtype, count):
        out = np.empty(count, dtype)
        if out.nbytes:
            _ck(capi.lib().gmg_memcpy_d2h(_ptr(out), self.ptr, out.nbytes, None))
        return out

    def free(self):
        if self.ptr:
            capi.lib().gmg_device_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Icm:
    """Host ICM_t behind gmg_icm (model I/O + null-model builder run on the host, as in the reference)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def open(cls, path):
        h = C.c_void_p()
        _ck(capi.lib().gmg_icm_open(str(path).encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def new(cls, model_len=12, model_depth=7, periodicity=3):
        h = C.c_void_p()
        _ck(capi.lib().gmg_icm_new(model_len, model_depth, periodicity, C.byref(h)))
        return cls(h)

    @classmethod
    def indep(cls, gc_frac, stops=DEFAULT_STOPS):
        """ICM_t Indep_Model(3,2,3); Indep_Model.Build_Indep_WO_Stops(gc, stops)  (glimmer3.cc:64,214)"""
        m = cls.new(3, 2, 3)
        arr = (C.c_char_p * len(stops))(*[s.encode() for s in stops])
        _ck(capi.lib().gmg_icm_build_indep(m.h, float(gc_frac), arr, len(stops)))
        return m

    @classmethod
    def train(cls, strings, model_len=12, model_depth=7, periodicity=3):
        """ICM_Training_t model(w, d, p); model.Train_Model(strings)  (src/ICM/build-icm.cc:67,120).  strings: lower-case
        str/bytes in the orientation Train_Model gets them.  The counting runs on the device (gmg_trainer_*)."""
        raw = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in strings]
        arr = (C.c_char_p * max(len(raw), 1))(*raw)
        h = C.c_void_p()
        _ck(capi.lib().gmg_icm_train(arr, len(raw), model_len, model_depth, periodicity, C.byref(h)))
        return cls(h)

    @property
    def params(self):
        w, d, p, n = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _ck(capi.lib().gmg_icm_params(self.h, C.byref(w), C.byref(d), C.byref(p), C.byref(n)))
        return w.value, d.value, p.value, n.value

    def tables(self):
        """(mip int16 [P,N], prob float32 [P,N,4])"""
        _, _, p, n = self.params
        mip = np.empty((p, n), np.int16)
        prob = np.empty((p, n, 4), np.float32)
        _ck(capi.lib().gmg_icm_tables(self.h, _ptr(mip), _ptr(prob)))
        return mip, prob

    def write(self, path):
        _ck(capi.lib().gmg_icm_write(self.h, str(path).encode()))

    def device(self):
        """gmg_model handle (uploaded once, owned by this Icm)"""
        out = C.c_void_p()
        _ck(capi.lib().gmg_icm_device_model(self.h, C.byref(out)))
        return out

    def close(self):
        if self.h:
            capi.lib().gmg_icm_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NullSet:
    """gmg_null_set: (3,2,3) null models side by side on the device (glimmer-mg -c: Indep_Model per read)"""

    def __init__(self, icms):
        self.icms = list(icms)                           # keep the host models (and their device mirrors) alive
        arr = (C.c_void_p * len(self.icms))(*[m.device() for m in self.icms])
        self.h = C.c_void_p()
        _ck(capi.lib().gmg_null_set_upload(arr, len(self.icms), C.byref(self.h)))

    @classmethod
    def build(cls, gcs, stops=DEFAULT_STOPS):
        """gmg_null_set_build: Build_Indep_WO_Stops (gc, stops) for every GC value, one upload"""
        self = cls.__new__(cls)
        gcs = np.ascontiguousarray(gcs, np.float64)
        buf = ((C.c_char * 4) * 8)()
        for i, c in enumerate(stops):
            buf[i].value = c.encode()
        self.icms = [Icm.indep(float(gcs[0]), stops)]    # (mg_score_reads wants a model object for the unused null argument)
        self.h = C.c_void_p()
        _ck(capi.lib().gmg_null_set_build(_ptr(gcs), len(gcs), buf, len(stops), C.byref(self.h)))
        return self

    def close(self):
        if self.h:
            capi.lib().gmg_null_set_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_strings(seqs):
    """Filter + lower + 2-bit pack a list of str/bytes sequences -> (packed uint32, offsets uint64)."""
    lens = np.array([len(s) for s in seqs], np.uint64)
    off = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    packed = np.zeros(int(capi.lib().gmg_packed_words(total)), np.uint32)
    blob = b"".join(s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs)
    _ck(capi.lib().gmg_pack_bases(blob, total, 0, _ptr(packed)))
    return packed, off


class Reads:
    """A batch of reads resident in HBM (gmg_reads)."""

    def __init__(self, packed, offsets):
        self.offsets = np.ascontiguousarray(offsets, np.uint64)
        packed = np.ascontiguousarray(packed, np.uint32)
        self.n_reads = len(self.offsets) - 1
        self.total_bases = int(self.offsets[-1]) if len(self.offsets) else 0
        self.h = C.c_void_p()
        _ck(capi.lib().gmg_reads_upload(_ptr(packed), _ptr(self.offsets), self.n_reads, C.byref(self.h)))

    @classmethod
    def from_strings(cls, seqs):
        return cls(*pack_strings(seqs))

    @classmethod
    def from_fasta_bytes(cls, data):
        """gmg_fasta_ingest: the file's bytes are parsed on the device (Fasta_Read + Filter + tolower + 2-bit packing).
        -> (Reads, headers [bytes], gc_count)"""
        data = bytes(data)
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        index = C.c_void_p()
        _ck(capi.lib().gmg_fasta_ingest(data, len(data), C.byref(self.h), C.byref(index)))
        try:
            n, total, gc = C.c_uint64(), C.c_uint64(), C.c_uint64()
            _ck(capi.lib().gmg_fasta_info(index, C.byref(n), C.byref(total), C.byref(gc)))
            hb, he = np.zeros(max(n.value, 1), np.uint64), np.zeros(max(n.value, 1), np.uint64)
            _ck(capi.lib().gmg_fasta_headers(index, _ptr(hb), _ptr(he)))
        finally:
            capi.lib().gmg_fasta_free(index)
        self.n_reads, self.total_bases = int(n.value), int(total.value)
        self.offsets = None                              # live on the device only
        headers = [data[int(b):int(e)] for b, e in zip(hb[:n.value], he[:n.value])]
        return self, headers, int(gc.value)

    def download(self):
        """-> (packed uint32 words, offsets uint64): the batch as it sits in HBM (tests)"""
        n, total = C.c_uint64(), C.c_uint64()
        _ck(capi.lib().gmg_reads_info(self.h, C.byref(n), C.byref(total)))
        words = int(capi.lib().gmg_packed_words(total.value))
        packed, off = np.zeros(max(words, 1), np.uint32), np.zeros(n.value + 1, np.uint64)
        _ck(capi.lib().gmg_reads_download(self.h, _ptr(packed), _ptr(off)))
        return packed[:words], off

    def select(self, idx):
        """gmg_reads_select: a new device batch made of the reads idx (any order, repeats allowed)"""
        idx = np.ascontiguousarray(idx, np.uint64)
        sub = Reads.__new__(Reads)
        sub.h = C.c_void_p()
        _ck(capi.lib().gmg_reads_select(self.h, _ptr(idx), len(idx), C.byref(sub.h)))
        n, total = C.c_uint64(), C.c_uint64()
        _ck(capi.lib().gmg_reads_info(sub.h, C.byref(n), C.byref(total)))
        sub.n_reads, sub.total_bases, sub.offsets = int(n.value), int(total.value), None
        return sub

    def close(self):
        if self.h:
            capi.lib().gmg_reads_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Trainer:
    """gmg_trainer: the pair counts of build-icm's training, one tree level per call (levels in order)."""

    def __init__(self, strings, model_len=12, model_depth=7, periodicity=3):
        self.strings = strings            # the Reads batch must outlive the trainer
        self.shape = (model_len, model_depth, periodicity)
        self.h = C.c_void_p()
        _ck(capi.lib().gmg_trainer_create(strings.h, model_len, model_depth, periodicity, C.byref(self.h)))

    def level_counts(self, level, mip_prev=None):
        """-> int32 [periodicity, 4^level, max(model_len - 1, 1), 16]; mip_prev: int16 [periodicity, 4^(level-1)]"""
        w, _, p = self.shape
        out = np.zeros((p, 4 ** level, max(w - 1, 1), 16), np.int32)
        if mip_prev is not None:
            mip_prev = np.ascontiguousarray(mip_prev, np.int16)
        _ck(capi.lib().gmg_trainer_level_counts(self.h, level, _ptr(mip_prev) if mip_prev is not None else None,
                                                _ptr(out)))
        return out

    def close(self):
        if self.h:
            capi.lib().gmg_trainer_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Segments:
    """ORF-style scoring buffers (gmg_segments): rows of (read, lo, len, orient)."""

    def __init__(self, reads, rows):
        rows = np.ascontiguousarray(np.asarray(rows, np.uint32).reshape(-1, 4))
        self.n = rows.shape[0]
        self.rows = rows
        self.offsets = np.zeros(self.n + 1, np.uint64)
        total = C.c_uint64()
        self.h = C.c_void_p()
        _ck(capi.lib().gmg_segments_upload(reads.h, _ptr(rows), self.n, _ptr(self.offsets), C.byref(total),
                                           C.byref(self.h)))
        self.total_len = int(total.value)

    def split(self, flat):
        return [flat[int(self.offsets[i]):int(self.offsets[i + 1])] for i in range(self.n)]

    def close(self):
        if self.h:
            capi.lib().gmg_segments_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_score6(gene, null, reads, d_out=None, stream=None, row_stride=None, read_null=None):
    """Score_All_Frames (glimmer-mg.cc:1468-1510) for every read.  With d_out (a device pointer to
    6*row_stride doubles) the call is asynchronous on `stream` and returns None; otherwise the
    result comes back as a float64 array [6, total_bases].  row_stride (default total_bases) is the
    distance between the six rows in doubles (gmg_frame_score6_strided).
    null may be a NullSet with read_null[r] = the null model of read r (gmg_frame_score6_nulls)."""
    stride = reads.total_bases if row_stride is None else int(row_stride)

    def call(ptr):
        if isinstance(null, NullSet):
            rn = np.ascontiguousarray(read_null, np.uint32)
            assert len(rn) == reads.n_reads
            _ck(capi.lib().gmg_frame_score6_nulls(gene.device(), null.h, _ptr(rn), reads.h, ptr, stride, stream))
        else:
            _ck(capi.lib().gmg_frame_score6_strided(gene.device(), null.device(), reads.h, ptr, stride, stream))

    if d_out is not None:
        call(C.c_void_p(d_out))
        return None
    buf = _DeviceBuffer(6 * max(stride, 1) * 8)
    call(buf.ptr)
    _ck(capi.lib().gmg_synchronize(stream))
    out = buf.to_host(np.float64, 6 * stride).reshape(6, stride)[:, :reads.total_bases].copy()
    buf.free()
    return out


def _seg_call(fn, model, reads, segs, frame, count):
    buf = _DeviceBuffer(max(count, 1) * 8)
    _ck(fn(model.device(), reads.h, segs.h, int(frame), buf.ptr, None))
    _ck(capi.lib().gmg_synchronize(None))
    out = buf.to_host(np.float64, count)
    buf.free()
    return out


def segment_frame_score(model, reads, segs, frame):
    """ICM_t::Frame_Score (icm.cc:485-509) per segment, flat float64[total_len]"""
    return _seg_call(capi.lib().gmg_segment_frame_score, model, reads, segs, frame, segs.total_len)


def segment_cumscore(model, reads, segs, frame0):
    """ICM_t::Cumulative_Score (icm.cc:354-405) per segment, flat float64[total_len]"""
    return _seg_call(capi.lib().gmg_segment_cumscore, model, reads, segs, frame0, segs.total_len)


def score_string(model, reads, segs, frame0):
    """ICM_t::Score_String (icm.cc:864-903) per segment, float64[n]"""
    return _seg_call(capi.lib().gmg_score_string, model, reads, segs, frame0, segs.n)


def segment_partial_prob(model, reads, segs, frame):
    """ICM_t::Partial_Window_Prob (icm.cc:807-842) of each segment's last base, float64[n]"""
    return _seg_call(capi.lib().gmg_segment_partial_prob, model, reads, segs, frame, segs.n)


def all_frame_score(gene, reads, segs, prefix_len, frames):
    """All_Frame_Score (glimmer3.cc:328-359) per segment -> float64[n, 6]"""
    pre = _DeviceBuffer.from_host(np.asarray(prefix_len, np.uint32))
    frs = _DeviceBuffer.from_host(np.asarray(frames, np.int32))
    buf = _DeviceBuffer(max(segs.n, 1) * 48)
    _ck(capi.lib().gmg_all_frame_score(gene.device(), reads.h, segs.h, pre.ptr, frs.ptr, buf.ptr, None))
    _ck(capi.lib().gmg_synchronize(None))
    out = buf.to_host(np.float64, 6 * segs.n).reshape(segs.n, 6)
    for b in (pre, frs, buf):
        b.free()
    return out


ORF_DTYPE = np.dtype([("read", "<u4"), ("frame", "<i4"), ("stop_position", "<i4"), ("orf_len", "<i4")])
START_DTYPE = np.dtype([("score", "<f8"), ("j", "<i4"), ("pos", "<i4"), ("which", "<i4"), ("truncated", "<i2"),
                        ("first", "<i2")])
ORF_RESULT_DTYPE = np.dtype([("gene_score", "<f8"), ("best_score", "<f8"), ("start_begin", "<u4"), ("n_starts", "<u4"),
                             ("first_j", "<i4"), ("best_j", "<i4"), ("best_pos", "<i4"), ("is_tentative_gene", "<i2"),
                             ("orf_is_truncated", "<i2")])


def score_orfs(gene, null, reads, orfs, min_gene_len=75, allow_truncated=False, use_first_start=False,
               ignore_score_len=2**31 - 1, start_threshold=-6.0, start_codons=("atg", "gtg", "ttg")):
    """The scoring part of Score_Orfs (glimmer3.cc:1275-1552) for a batch of ORFs.
    orfs: rows (read, frame, stop_position, orf_len) as Find_Orfs produced them.
    -> (results[ORF_RESULT_DTYPE], starts[START_DTYPE]); ORF i's starts are
       starts[res.start_begin : res.start_begin + res.n_starts]."""
    rows = np.asarray(orfs)
    o = np.zeros(len(rows), ORF_DTYPE)
    if len(rows):
        o["read"], o["frame"], o["stop_position"], o["orf_len"] = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
    assert ORF_DTYPE.itemsize == 16 and START_DTYPE.itemsize == 24 and ORF_RESULT_DTYPE.itemsize == 40
    prm = capi.OrfParams(min_gene_len, int(allow_truncated), int(use_first_start), ignore_score_len,
                         start_threshold, len(start_codons))
    for i, c in enumerate(start_codons):
        prm.start_codon[i].value = c.encode()
    batch, max_starts = C.c_void_p(), C.c_uint64()
    _ck(capi.lib().gmg_orfs_upload(reads.h, _ptr(o), len(o), C.byref(max_starts), C.byref(batch)))
    res = np.zeros(len(o), ORF_RESULT_DTYPE)
    try:
        n_st = C.c_uint64()
        _ck(capi.lib().gmg_score_orfs_begin(gene.device(), null.device(), reads.h, batch, C.byref(prm), C.byref(n_st), None))
        starts = np.zeros(max(int(n_st.value), 1), START_DTYPE)
        _ck(capi.lib().gmg_score_orfs_fetch(batch, _ptr(res), _ptr(starts), None))
    finally:
        capi.lib().gmg_orf_batch_free(batch)
    return res, starts


MG_ORF_DTYPE = np.dtype([("read", "<u4"), ("frame", "<i4"), ("stop_position", "<i4"), ("orf_len", "<i4"),
                         ("gene_len", "<i4"), ("lo", "<i4"), ("hi", "<i4"), ("first_j", "<i4"),
                         ("start_begin", "<u4"), ("n_starts", "<u4"), ("accepted", "<i2"),
                         ("orf_is_truncated", "<i2"), ("reserved", "<i4"), ("best_score", "<f8")])


def find_orfs(reads, min_gene_len=75, allow_truncated=False, start_codons=("atg", "gtg", "ttg"),
              stop_codons=("taa", "tag", "tga"), circular=False, ignore_regions=()):
    """gmg_find_orfs: Find_Orfs for every read -> (orfs[MG_ORF_DTYPE], read_orf_off[uint64 n_reads+1]).
    circular: Genome_Is_Circular; ignore_regions: [(lo, hi)] as Get_Ignore_Regions leaves them (0-based lo, hi one past the end)"""
    prm = capi.MgParams(min_gene_len, int(allow_truncated), 2**31 - 1, len(start_codons), len(stop_codons), 0, 0.0)
    for i, c in enumerate(start_codons):
        prm.start_codon[i].value = c.encode()
    for i, c in enumerate(stop_codons):
        prm.stop_codon[i].value = c.encode()
    prm.circular = int(bool(circular))
    lo = np.ascontiguousarray([r[0] for r in ignore_regions], np.int32)
    hi = np.ascontiguousarray([r[1] for r in ignore_regions], np.int32)
    prm.n_ignore_regions = len(lo)
    if len(lo):
        prm.ignore_lo, prm.ignore_hi = lo.ctypes.data, hi.ctypes.data
    res = C.c_void_p()
    _ck(capi.lib().gmg_find_orfs(reads.h, C.byref(prm), C.byref(res), None))
    try:
        n_orfs, n_starts = C.c_uint64(), C.c_uint64()
        _ck(capi.lib().gmg_mg_result_info(res, C.byref(n_orfs), C.byref(n_starts)))
        orfs = np.zeros(max(n_orfs.value, 1), MG_ORF_DTYPE)
        off = np.zeros(reads.n_reads + 1, np.uint64)
        _ck(capi.lib().gmg_mg_result_fetch(res, _ptr(orfs), None, _ptr(off)))
    finally:
        capi.lib().gmg_mg_result_free(res)
    return orfs[:n_orfs.value], off


START_ERRORS_DTYPE = np.dtype([("pos", "<i4", (2,)), ("type", "i1", (2,)), ("n", "i1"), ("reserved", "i1")])
MG_ACCEPTED_ONLY, MG_ALLOW_INDELS, MG_ALLOW_SUBS = 1, 2, 4


class Classes:
    """gmg_classes: glimmer-mg -c bookkeeping (Parse_Classes, Read_Meta_ICMs / _GC / _Stops; host only)"""

    def __init__(self, class_text, icm_dir):
        if isinstance(class_text, str):
            class_text = class_text.encode()
        self.h = C.c_void_p()
        _ck(capi.lib().gmg_classes_load(class_text, len(class_text), icm_dir.encode(), C.byref(self.h)))
        n, k, c, miss = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        _ck(capi.lib().gmg_classes_info(self.h, C.byref(n), C.byref(k), C.byref(c), C.byref(miss)))
        self.n_reads, self.n_icms, self.n_classes, self.n_missing_gc = n.value, k.value, c.value, miss.value

    def icm_file(self, k):
        p = capi.lib().gmg_classes_icm_file(self.h, k)
        if p is None:
            _ck(-1)
        return p.decode()

    def plan(self, headers):
        """headers: the header LINES of one chunk -> (order, icm_begin, gc, transl): gmg_classes_plan"""
        hb = [h.encode() if isinstance(h, str) else h for h in headers]
        n = len(hb)
        arr = (C.c_char_p * max(n, 1))(*hb)
        lens = np.array([len(h) for h in hb], np.uint32)
        order = np.zeros(max(n, 1), np.uint64)
        icm_begin = np.zeros(self.n_icms + 1, np.uint64)
        gc = np.zeros(max(n, 1), np.float64)
        transl = np.zeros(max(n, 1), np.int32)
        k = C.c_uint64()
        _ck(capi.lib().gmg_classes_plan(self.h, arr, _ptr(lens), n, _ptr(order), _ptr(icm_begin), _ptr(gc), _ptr(transl), C.byref(k)))
        return order[:k.value], icm_begin, gc[:k.value], transl[:k.value]

    def close(self):
        if self.h:
            capi.lib().gmg_classes_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stop_codons_by_code(code):
    """gmg_stop_codons_by_code: Set_Stop_Codons_By_Code (gene.cc:1560-1624) -> tuple of codons"""
    buf = ((C.c_char * 4) * 8)()
    n = C.c_int()
    _ck(capi.lib().gmg_stop_codons_by_code(int(code), buf, C.byref(n)))
    return tuple(buf[i].value.decode() for i in range(n.value))


def ignore_score_len(gc, stops=DEFAULT_STOPS):
    """gmg_ignore_score_len: Set_Ignore_Score_Len (glimmer_base.cc:2597-2633)"""
    buf = ((C.c_char * 4) * 8)()
    for i, c in enumerate(stops):
        buf[i].value = c.encode()
    out = C.c_int32()
    _ck(capi.lib().gmg_ignore_score_len(float(gc), buf, len(stops), C.byref(out)))
    return out.value


def mg_score_reads(gene, null, reads, min_gene_len=75, allow_truncated=True, ignore_score_len=2**31 - 1,
                   start_threshold=-6.0, start_codons=("atg", "gtg", "ttg"), stop_codons=("taa", "tag", "tga"),
                   frame_scores=None, accepted_only=False, allow_indels=False, allow_subs=False, quality=None,
                   min_indel_orf_len=15, indel_quality_threshold=18, indel_max=2, indel_suffix_score_threshold=-12.0,
                   read_null=None, read_ignore_score_len=None, groups=None):
    """glimmer-mg's front half for a batch of reads (include/gmg.h: gmg_mg_score_reads): Score_All_Frames,
    Find_Orfs, Score_Orf_Starts and the filter of Score_Orfs_Errors.
    -> (orfs[MG_ORF_DTYPE], starts[START_DTYPE], read_orf_off[uint64 n_reads+1]).
    frame_scores: optional _DeviceBuffer of 6*total_bases doubles that receives the Frame_Scores table.
    allow_indels / allow_subs: glimmer-mg's error branch (-i / -s); the starts' Error_t lists come back as a fourth
    array (START_ERRORS_DTYPE, parallel to starts).  quality: uint8 Phred values of all bases (the -q file), or None
    for Set_Quality_454."""
    assert MG_ORF_DTYPE.itemsize == 56 and START_ERRORS_DTYPE.itemsize == 12
    err = bool(allow_indels or allow_subs)
    flags = (MG_ACCEPTED_ONLY if accepted_only else 0) | (MG_ALLOW_INDELS if allow_indels else 0) | (MG_ALLOW_SUBS if allow_subs else 0)
    prm = capi.MgParams(min_gene_len, int(allow_truncated), ignore_score_len, len(start_codons), len(stop_codons),
                        flags, start_threshold)
    prm.min_indel_orf_len, prm.indel_quality_threshold, prm.indel_max = min_indel_orf_len, indel_quality_threshold, indel_max
    prm.indel_suffix_score_threshold = indel_suffix_score_threshold
    if quality is not None:
        quality = np.ascontiguousarray(quality, np.uint8)
        if quality.size != reads.total_bases:
            raise ValueError("quality needs one value per base of the batch")
        prm.quality = quality.ctypes.data
    for i, c in enumerate(start_codons):
        prm.start_codon[i].value = c.encode()
    for i, c in enumerate(stop_codons):
        prm.stop_codon[i].value = c.encode()
    null_model = null
    if isinstance(null, NullSet):                       # classification mode: null model (and Ignore_Score_Len) per read
        read_null = np.ascontiguousarray(read_null, np.uint32)
        assert len(read_null) == reads.n_reads
        prm.nulls, prm.read_null = null.h, read_null.ctypes.data
        if read_ignore_score_len is not None:
            read_ignore_score_len = np.ascontiguousarray(read_ignore_score_len, np.int32)
            assert len(read_ignore_score_len) == reads.n_reads
            prm.read_ignore_score_len = read_ignore_score_len.ctypes.data
        null_model = null.icms[0]
    res = C.c_void_p()
    if groups is not None:                              # gmg_mg_score_groups: [(Icm, read_begin, read_end), ...], `gene` is not used
        arr = (capi.MgGroup * max(len(groups), 1))(*[capi.MgGroup(m.device(), int(b), int(e)) for m, b, e in groups])
        _ck(capi.lib().gmg_mg_score_groups(arr, len(groups), null_model.device(), reads.h, C.byref(prm), C.byref(res), None))
    else:
        _ck(capi.lib().gmg_mg_score_reads(gene.device(), null_model.device(), reads.h, C.byref(prm),
                                          frame_scores.ptr if frame_scores is not None else None, C.byref(res), None))
    try:
        n_orfs, n_starts = C.c_uint64(), C.c_uint64()
        _ck(capi.lib().gmg_mg_result_info(res, C.byref(n_orfs), C.byref(n_starts)))
        orfs = np.zeros(max(n_orfs.value, 1), MG_ORF_DTYPE)
        starts = np.zeros(max(n_starts.value, 1), START_DTYPE)
        off = np.zeros(reads.n_reads + 1, np.uint64)
        _ck(capi.lib().gmg_mg_result_fetch(res, _ptr(orfs), _ptr(starts), _ptr(off)))
        if err:
            errs = np.zeros(max(n_starts.value, 1), START_ERRORS_DTYPE)
            _ck(capi.lib().gmg_mg_result_fetch_errors(res, _ptr(errs)))
    finally:
        capi.lib().gmg_mg_result_free(res)
    if err:
        return orfs[:n_orfs.value], starts[:n_starts.value], off, errs[:n_starts.value]
    return orfs[:n_orfs.value], starts[:n_starts.value], off


def score_reads_strings(models, reads):
    """gmg_score_reads_strings: whole-read Score_String (frame 0) of every read and of its reverse complement under
    every model -> float64 [n_models, n_reads, 2]"""
    n = len(models)
    arr = (C.c_void_p * max(n, 1))(*[m.device() for m in models])
    buf = _DeviceBuffer(max(n * reads.n_reads * 2, 1) * 8)
    _ck(capi.lib().gmg_score_reads_strings(arr, n, reads.h, buf.ptr, None))
    out = buf.to_host(np.float64, n * reads.n_reads * 2).reshape(n, reads.n_reads, 2)
    buf.free()
    return out


def window_distrib(model, windows, frames):
    """Full_Window_Distrib / Full_Window_Prob (icm.cc:512-610).  windows: uint8 codes [n, model_len]
    -> (dist float32 [n,4], prob float64 [n])"""
    windows = np.ascontiguousarray(windows, np.uint8)
    n = windows.shape[0]
    dw = _DeviceBuffer.from_host(windows)
    df = _DeviceBuffer.from_host(np.asarray(frames, np.int32))
    dd = _DeviceBuffer(max(n, 1) * 16)
    dp = _DeviceBuffer(max(n, 1) * 8)
    _ck(capi.lib().gmg_window_distrib(model.device(), dw.ptr, df.ptr, n, dd.ptr, dp.ptr, None))
    _ck(capi.lib().gmg_synchronize(None))
    dist = dd.to_host(np.float32, 4 * n).reshape(n, 4)
    prob = dp.to_host(np.float64, n)
    for b in (dw, df, dd, dp):
        b.free()
    return dist, prob
