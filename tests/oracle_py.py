"""ctypes binding over oracle/libgmg_oracle.so -- the CPU oracle.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libgmg_oracle.so")


class Model(C.Structure):
    _fields_ = [("model_len", C.c_int), ("model_depth", C.c_int), ("periodicity", C.c_int),
                ("num_nodes", C.c_int), ("mip", C.POINTER(C.c_int16)), ("prob", C.POINTER(C.c_float))]


class Start(C.Structure):
    _fields_ = [("score", C.c_double), ("j", C.c_int), ("pos", C.c_int), ("which", C.c_int),
                ("truncated", C.c_int), ("first", C.c_int)]


class OrfParams(C.Structure):
    _fields_ = [("min_gene_len", C.c_int), ("allow_truncated", C.c_int), ("use_first_start", C.c_int),
                ("ignore_score_len", C.c_int), ("start_threshold", C.c_double), ("n_start_codons", C.c_int),
                ("start_codon", C.c_char_p * 8)]


class OrfOut(C.Structure):
    _fields_ = [("gene_score", C.c_double), ("best_score", C.c_double), ("first_j", C.c_int), ("best_j", C.c_int),
                ("best_pos", C.c_int), ("is_tentative_gene", C.c_int), ("orf_is_truncated", C.c_int)]


MP = C.POINTER(Model)
dp = C.POINTER(C.c_double)


class Orf(C.Structure):
    _fields_ = [("frame", C.c_int), ("stop_position", C.c_int), ("gene_len", C.c_int), ("orf_len", C.c_int)]


class MgParams(C.Structure):
    _fields_ = [("min_gene_len", C.c_int), ("allow_truncated", C.c_int), ("ignore_score_len", C.c_int),
                ("start_threshold", C.c_double), ("n_start_codons", C.c_int), ("n_stop_codons", C.c_int),
                ("start_codon", C.c_char_p * 8), ("stop_codon", C.c_char_p * 8)]


class MgOut(C.Structure):
    _fields_ = [("lo", C.c_int), ("hi", C.c_int), ("first_j", C.c_int), ("accepted", C.c_int),
                ("orf_is_truncated", C.c_int), ("best_score", C.c_double)]


class MgErrParams(C.Structure):
    _fields_ = [("allow_indels", C.c_int), ("allow_subs", C.c_int), ("indel_quality_threshold", C.c_int),
                ("indel_max", C.c_int), ("indel_suffix_score_threshold", C.c_double)]


class StartErr(C.Structure):
    _fields_ = [("s", Start), ("n_errors", C.c_int), ("err_pos", C.c_int * 4), ("err_type", C.c_int * 4)]


def build():
    src = [os.path.join(ORACLE_DIR, f) for f in ("gmg_oracle.c", "gmg_oracle.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, stdout=subprocess.DEVNULL)
    return LIB


class Oracle:
    def __init__(self):
        L = C.CDLL(build())
        self.L = L
        L.orc_model_new.restype = MP
        L.orc_model_new.argtypes = [C.c_int] * 3
        L.orc_model_read.restype = MP
        L.orc_model_read.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_model_free.argtypes = [MP]
        L.orc_model_write.argtypes = [MP, C.c_char_p]
        L.orc_build_indep_wo_stops.argtypes = [MP, C.c_double, C.POINTER(C.c_char_p), C.c_int]
        L.orc_full_window_prob.restype = C.c_double
        L.orc_full_window_prob.argtypes = [MP, C.c_char_p, C.c_int]
        L.orc_partial_window_prob.restype = C.c_double
        L.orc_partial_window_prob.argtypes = [MP, C.c_int, C.c_char_p, C.c_int]
        L.orc_full_window_distrib.argtypes = [MP, C.c_char_p, C.c_int, C.POINTER(C.c_float)]
        L.orc_score_string.restype = C.c_double
        L.orc_score_string.argtypes = [MP, C.c_char_p, C.c_int, C.c_int]
        L.orc_cumulative_score.argtypes = [MP, C.c_char_p, C.c_int, dp, C.c_int]
        L.orc_cumulative_score_string.argtypes = [MP, C.c_char_p, C.c_int, C.c_int, dp]
        L.orc_frame_score.argtypes = [MP, C.c_char_p, C.c_int, dp, C.c_int]
        L.orc_score_all_frames.argtypes = [MP, MP, C.c_char_p, C.c_int, dp]
        L.orc_cumulative_frame_score.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.orc_all_frame_score.argtypes = [MP, C.c_char_p, C.c_int, C.c_int, dp]
        L.orc_reverse_transfer.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.orc_complement_transfer.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.orc_score_reads_6frame.restype = C.c_long
        L.orc_score_reads_6frame.argtypes = [MP, MP, C.c_char_p, C.c_int, C.c_int, dp]
        L.orc_score_orf.argtypes = [MP, MP, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrfParams),
                                    C.POINTER(Start), C.c_int, C.POINTER(OrfOut)]
        ip = C.POINTER(C.c_int)
        L.orc_find_orfs.argtypes = [C.c_char_p, C.c_int, C.POINTER(MgParams), C.POINTER(Orf), C.c_int]
        L.orc_save_prev_stops.argtypes = [C.c_char_p, C.c_int, C.POINTER(MgParams), ip, ip]
        L.orc_find_orfs_err.argtypes = [C.c_char_p, C.c_int, C.POINTER(MgParams), C.c_int, C.POINTER(Orf), C.c_int]
        L.orc_set_quality_454.argtypes = [C.c_char_p, C.c_int, ip]
        L.orc_clean_quality_454.argtypes = [C.c_char_p, C.c_int, ip, C.c_int]
        L.orc_mg_score_orf_errors.argtypes = [dp, C.c_char_p, C.c_int, ip, ip, ip, C.c_int, C.c_int, C.POINTER(MgParams),
                                              C.POINTER(MgErrParams), C.POINTER(StartErr), C.c_int, C.POINTER(MgOut)]
        L.orc_mg_score_orf.argtypes = [dp, C.c_char_p, C.c_int, ip, ip, C.c_int, C.c_int, C.POINTER(MgParams),
                                       C.POINTER(Start), C.c_int, C.POINTER(MgOut)]
        lp = C.POINTER(C.c_long)
        L.orc_fasta_next.argtypes = [C.c_char_p, C.c_long, lp, lp, lp, C.c_char_p, lp]
        L.orc_fasta_all.restype = C.c_long
        L.orc_fasta_all.argtypes = [C.c_char_p, C.c_long, C.c_char_p, lp, lp]
        L.orc_train_level_counts.argtypes = [MP, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.orc_mutual_info.restype = C.c_double
        L.orc_mutual_info.argtypes = [C.POINTER(C.c_int32), C.c_int]
        L.orc_train_model.restype = MP
        L.orc_train_model.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        for f in ("orc_filter", "orc_complement", "orc_subscript"):
            getattr(L, f).argtypes = [C.c_int]
        L.orc_classes_load.restype = C.c_void_p
        L.orc_classes_load.argtypes = [C.c_char_p, C.c_long, C.c_char_p]
        L.orc_classes_free.argtypes = [C.c_void_p]
        L.orc_classes_n_icms.argtypes = [C.c_void_p]
        L.orc_classes_icm_file.restype = C.c_char_p
        L.orc_classes_icm_file.argtypes = [C.c_void_p, C.c_int]
        L.orc_classes_plan.restype = C.c_long
        L.orc_classes_plan.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_long, lp, lp, dp, ip]
        L.orc_stop_codons_by_code.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.orc_ignore_score_len.argtypes = [C.c_double, C.POINTER(C.c_char_p), C.c_int]

    # ---- glimmer-mg -c bookkeeping
    def classes_load(self, text, icm_dir):
        if isinstance(text, str):
            text = text.encode()
        return self.L.orc_classes_load(text, len(text), icm_dir.encode())       # None where the reference would crash

    def classes_icm_files(self, c):
        return [self.L.orc_classes_icm_file(c, k).decode() for k in range(self.L.orc_classes_n_icms(c))]

    def classes_plan(self, c, headers):
        n = len(headers)
        arr = (C.c_char_p * max(n, 1))(*[h.encode() for h in headers])
        order = np.zeros(max(n, 1), np.int64)
        icm_begin = np.zeros(self.L.orc_classes_n_icms(c) + 1, np.int64)
        gc = np.zeros(max(n, 1), np.float64)
        transl = np.zeros(max(n, 1), np.int32)
        k = self.L.orc_classes_plan(c, arr, n, order.ctypes.data_as(C.POINTER(C.c_long)), icm_begin.ctypes.data_as(C.POINTER(C.c_long)),
                                    gc.ctypes.data_as(dp), transl.ctypes.data_as(C.POINTER(C.c_int)))
        return order[:k], icm_begin, gc[:k], transl[:k]

    def stop_codons_by_code(self, code):
        out = (C.c_char_p * 8)()
        n = self.L.orc_stop_codons_by_code(int(code), out)
        return tuple(out[i].decode() for i in range(n))

    def ignore_score_len(self, gc, stops=("taa", "tag", "tga")):
        arr = (C.c_char_p * len(stops))(*[s.encode() for s in stops])
        return self.L.orc_ignore_score_len(float(gc), arr, len(stops))

    # ---- models
    def read(self, path):
        err = C.create_string_buffer(256)
        m = self.L.orc_model_read(str(path).encode(), err, 256)
        if not m:
            raise RuntimeError(err.value.decode())
        return m

    def indep(self, gc, stops=("taa", "tag", "tga")):
        m = self.L.orc_model_new(3, 2, 3)
        arr = (C.c_char_p * len(stops))(*[s.encode() for s in stops])
        assert self.L.orc_build_indep_wo_stops(m, float(gc), arr, len(stops)) == 0
        return m

    def tables(self, m):
        c = m.contents
        n = c.periodicity * c.num_nodes
        mip = np.ctypeslib.as_array(c.mip, (n,)).reshape(c.periodicity, c.num_nodes).copy()
        prob = np.ctypeslib.as_array(c.prob, (4 * n,)).reshape(c.periodicity, c.num_nodes, 4).copy()
        return mip, prob

    # ---- scoring
    def score_all_frames(self, gene, indep, seq):
        s = seq.encode() if isinstance(seq, str) else seq
        out = np.empty((6, len(s)), np.float64)
        self.L.orc_score_all_frames(gene, indep, s, len(s), out.ctypes.data_as(dp))
        return out

    def score_string(self, m, s, frame, n=None):
        s = s.encode() if isinstance(s, str) else s
        return self.L.orc_score_string(m, s, len(s) if n is None else n, frame)

    def cumulative_score(self, m, s, frame):
        s = s.encode() if isinstance(s, str) else s
        out = np.empty(len(s), np.float64)
        self.L.orc_cumulative_score(m, s, len(s), out.ctypes.data_as(dp), frame)
        return out

    def frame_score(self, m, s, frame):
        s = s.encode() if isinstance(s, str) else s
        out = np.empty(len(s), np.float64)
        self.L.orc_frame_score(m, s, len(s), out.ctypes.data_as(dp), frame)
        return out

    def full_window(self, m, w, frame):
        w = w.encode() if isinstance(w, str) else w
        dist = (C.c_float * 4)()
        self.L.orc_full_window_distrib(m, w, frame, dist)
        return self.L.orc_full_window_prob(m, w, frame), np.array(dist[:], np.float32)

    def partial_window(self, m, pos, s, frame):
        s = s.encode() if isinstance(s, str) else s
        return self.L.orc_partial_window_prob(m, pos, s, frame)

    def all_frame_score(self, m, s, n, frame):
        s = s.encode() if isinstance(s, str) else s
        out = np.empty(6, np.float64)
        self.L.orc_all_frame_score(m, s, n, frame, out.ctypes.data_as(dp))
        return out

    def buffer(self, seq, lo, ln, orient):
        """the scoring buffer of include/gmg.h's gmg_orient, built with the oracle's transfer routines"""
        s = seq.encode() if isinstance(seq, str) else seq
        buf = C.create_string_buffer(ln + 1)
        if ln == 0:
            return b""
        if orient == 1:       # REVERSED
            self.L.orc_reverse_transfer(buf, s, len(s), lo + ln - 1, ln)
        elif orient == 2:     # COMPLEMENTED
            self.L.orc_complement_transfer(buf, s, len(s), lo, ln)
        elif orient == 0:     # FORWARD
            return s[lo:lo + ln]
        else:                 # REVCOMP
            tmp = C.create_string_buffer(ln + 1)
            self.L.orc_reverse_transfer(tmp, s, len(s), lo + ln - 1, ln)
            self.L.orc_complement_transfer(buf, tmp.raw[:ln], ln, 0, ln)
        return buf.raw[:ln]

    @staticmethod
    def orf_params(min_gene_len=75, allow_truncated=False, use_first_start=False, ignore_score_len=2**31 - 1,
                   start_threshold=-6.0, start_codons=("atg", "gtg", "ttg")):
        """defaults of src/Glimmer/glimmer3.cc:23,61,71,122,148 and glimmer_base.hh DEFAULT_START_CODON"""
        p = OrfParams(min_gene_len, int(allow_truncated), int(use_first_start), ignore_score_len, start_threshold,
                      len(start_codons))
        for i, c in enumerate(start_codons):
            p.start_codon[i] = c.encode()
        return p

    def score_orf(self, gene, indep, seq, frame, stop_position, orf_len, prm):
        """-> (n_starts or -1, OrfOut, [Start])"""
        s = seq.encode() if isinstance(seq, str) else seq
        cap = orf_len // 3 + 4
        starts = (Start * cap)()
        out = OrfOut()
        n = self.L.orc_score_orf(gene, indep, s, len(s), frame, stop_position, orf_len, C.byref(prm), starts, cap,
                                 C.byref(out))
        return n, out, list(starts[:max(n, 0)])

    # ---- glimmer-mg front half
    @staticmethod
    def mg_params(min_gene_len=75, allow_truncated=True, ignore_score_len=2**31 - 1, start_threshold=-6.0,
                  start_codons=("atg", "gtg", "ttg"), stop_codons=("taa", "tag", "tga")):
        p = MgParams(min_gene_len, int(allow_truncated), ignore_score_len, start_threshold, len(start_codons),
                     len(stop_codons))
        for i, c in enumerate(start_codons):
            p.start_codon[i] = c.encode()
        for i, c in enumerate(stop_codons):
            p.stop_codon[i] = c.encode()
        return p

    def find_orfs(self, seq, prm):
        """Find_Orfs -> int32 [n, 4]: frame, stop_position, gene_len, orf_len"""
        s = seq.encode() if isinstance(seq, str) else seq
        cap = 2 * len(s) + 16
        buf = (Orf * cap)()
        n = self.L.orc_find_orfs(s, len(s), C.byref(prm), buf, cap)
        assert n <= cap
        return np.array([(o.frame, o.stop_position, o.gene_len, o.orf_len) for o in buf[:n]], np.int32).reshape(-1, 4)

    def find_orfs_general(self, seq, prm, circular=False, regions=(), min_indel_orf_len=-1):
        """Find_Orfs with ignore regions [(lo, hi)] (as Get_Ignore_Regions leaves them) and on a circular sequence
        -> int32 [n, 4]: frame, stop_position, gene_len, orf_len; None where the reference's assert would fire"""
        s = seq.encode() if isinstance(seq, str) else seq
        cap = 2 * len(s) + 16
        buf = (Orf * cap)()
        lo = (C.c_int * max(len(regions), 1))(*[int(r[0]) for r in regions])
        hi = (C.c_int * max(len(regions), 1))(*[int(r[1]) for r in regions])
        self.L.orc_find_orfs_general.argtypes = [C.c_char_p, C.c_int, C.POINTER(MgParams), C.c_int, C.c_int, C.POINTER(C.c_int),
                                                 C.POINTER(C.c_int), C.c_int, C.POINTER(Orf), C.c_int]
        self.L.orc_find_orfs_general.restype = C.c_int
        n = self.L.orc_find_orfs_general(s, len(s), C.byref(prm), int(min_indel_orf_len), int(circular), lo, hi, len(regions), buf, cap)
        if n < 0:
            return None
        assert n <= cap
        return np.array([(o.frame, o.stop_position, o.gene_len, o.orf_len) for o in buf[:n]], np.int32).reshape(-1, 4)

    def save_prev_stops(self, seq, prm):
        s = seq.encode() if isinstance(seq, str) else seq
        fwd, rev = np.zeros(max(len(s), 1), np.int32), np.zeros(max(len(s), 1), np.int32)
        ip = C.POINTER(C.c_int)
        self.L.orc_save_prev_stops(s, len(s), C.byref(prm), fwd.ctypes.data_as(ip), rev.ctypes.data_as(ip))
        return fwd, rev

    def mg_score_orf(self, frame_scores, seq, fwd_prev, rev_next, frame, stop_position, prm):
        """-> (MgOut, [Start] in push order)"""
        s = seq.encode() if isinstance(seq, str) else seq
        fs = np.ascontiguousarray(frame_scores, np.float64)
        cap = len(s) // 3 + 8
        starts = (Start * cap)()
        out = MgOut()
        ip = C.POINTER(C.c_int)
        n = self.L.orc_mg_score_orf(fs.ctypes.data_as(dp), s, len(s), fwd_prev.ctypes.data_as(ip),
                                    rev_next.ctypes.data_as(ip), frame, stop_position, C.byref(prm), starts, cap,
                                    C.byref(out))
        assert n <= cap
        return out, list(starts[:n])

    def mg_read(self, gene, indep, seq, prm):
        """whole front half for one read: -> (orfs [n,4], [(MgOut, [Start])])"""
        orfs = self.find_orfs(seq, prm)
        if len(orfs) == 0:
            return orfs, []
        fs = self.score_all_frames(gene, indep, seq)
        fwd, rev = self.save_prev_stops(seq, prm)
        return orfs, [self.mg_score_orf(fs, seq, fwd, rev, int(o[0]), int(o[1]), prm) for o in orfs]

    # ---- glimmer-mg's error branch (-i / -s)
    @staticmethod
    def mg_err_params(allow_indels=False, allow_subs=False, indel_quality_threshold=18, indel_max=2,
                      indel_suffix_score_threshold=-12.0):
        """defaults of src/Glimmer/glimmer-mg.cc:134-138"""
        return MgErrParams(int(allow_indels), int(allow_subs), indel_quality_threshold, indel_max, indel_suffix_score_threshold)

    def find_orfs_err(self, seq, prm, min_indel_orf_len=15):
        s = seq.encode() if isinstance(seq, str) else seq
        cap = 2 * len(s) + 16
        buf = (Orf * cap)()
        n = self.L.orc_find_orfs_err(s, len(s), C.byref(prm), min_indel_orf_len, buf, cap)
        assert n <= cap
        return np.array([(o.frame, o.stop_position, o.gene_len, o.orf_len) for o in buf[:n]], np.int32).reshape(-1, 4)

    def quality_454(self, seq, user=None, indel_quality_threshold=18):
        """Quality_Values of a read: Set_Quality_454 (no file) or Clean_Quality_454 of the user's values -> int32 [n]"""
        s = seq.encode() if isinstance(seq, str) else seq
        q = np.zeros(max(len(s), 1), np.int32) if user is None else np.ascontiguousarray(user, np.int32).copy()
        ip = C.POINTER(C.c_int)
        if user is None:
            self.L.orc_set_quality_454(s, len(s), q.ctypes.data_as(ip))
        else:
            assert len(q) == len(s)
            self.L.orc_clean_quality_454(s, len(s), q.ctypes.data_as(ip), indel_quality_threshold)
        return q[:len(s)]

    def mg_score_orf_errors(self, frame_scores, seq, fwd_prev, rev_next, quality, frame, stop_position, prm, ep, cap=1 << 10):
        """-> (MgOut, [StartErr] in push order)"""
        s = seq.encode() if isinstance(seq, str) else seq
        fs = np.ascontiguousarray(frame_scores, np.float64)
        ip = C.POINTER(C.c_int)
        qp = None if quality is None else np.ascontiguousarray(quality, np.int32).ctypes.data_as(ip)
        while True:
            starts = (StartErr * cap)()
            out = MgOut()
            n = self.L.orc_mg_score_orf_errors(fs.ctypes.data_as(dp), s, len(s), fwd_prev.ctypes.data_as(ip),
                                               rev_next.ctypes.data_as(ip), qp, frame, stop_position, C.byref(prm),
                                               C.byref(ep), starts, cap, C.byref(out))
            if n <= cap:
                return out, list(starts[:n])
            cap = n

    def mg_read_errors(self, gene, indep, seq, prm, ep, user_quality=None, min_indel_orf_len=15):
        """front half of one read with the error branch: -> (orfs [n,4], quality, [(MgOut, [StartErr])])"""
        orfs = self.find_orfs_err(seq, prm, min_indel_orf_len)
        quality = self.quality_454(seq, user_quality, ep.indel_quality_threshold) if ep.allow_indels else None
        if len(orfs) == 0:
            return orfs, quality, []
        fs = self.score_all_frames(gene, indep, seq)
        fwd, rev = self.save_prev_stops(seq, prm)
        return orfs, quality, [self.mg_score_orf_errors(fs, seq, fwd, rev, quality, int(o[0]), int(o[1]), prm, ep) for o in orfs]

    def fasta_records(self, data):
        """the loop  while (Fasta_Read (fp, s, hdr))  -> [(hdr bytes, tolower(Filter(s)) bytes)], gc count"""
        data = bytes(data)
        pos, hb, he, sl = C.c_long(0), C.c_long(), C.c_long(), C.c_long()
        buf = C.create_string_buffer(len(data) + 1)
        out, gc = [], 0
        while self.L.orc_fasta_next(data, len(data), C.byref(pos), C.byref(hb), C.byref(he), buf, C.byref(sl)):
            seq = self.filter_lower(buf.raw[:sl.value])
            gc += seq.count(b"g") + seq.count(b"c")
            out.append((data[hb.value:he.value], seq))
        return out, gc

    def fasta_all(self, data):
        """-> (n_records, n_bases, gc count): the reference's ingest loop in C (timing baseline)"""
        nb, gc = C.c_long(), C.c_long()
        n = self.L.orc_fasta_all(data, len(data), None, C.byref(nb), C.byref(gc))
        return n, nb.value, gc.value

    def filter_lower(self, seq):
        """tolower(Filter(c)) per character (glimmer3.cc:270-271)"""
        return bytes(ord(chr(self.L.orc_filter(c)).lower()) for c in (seq.encode() if isinstance(seq, str) else seq))

    # ---- training (build-icm)
    @staticmethod
    def _strings(strings):
        return (C.c_char_p * len(strings))(*strings)

    def train_model(self, strings, W=12, D=7, P=3, want_mut_info=False):
        """strings: list of bytes (lower case).  Returns the model (and the mut_info floats)."""
        n_nodes = (4 ** (D + 1) - 1) // 3
        mi = np.zeros(P * n_nodes, np.float32)
        m = self.L.orc_train_model(self._strings(strings), len(strings), W, D, P,
                                   mi.ctypes.data_as(C.POINTER(C.c_float)))
        return (m, mi) if want_mut_info else m

    def train_level_counts(self, m, strings, level):
        mm = m.contents
        out = np.zeros((mm.periodicity, 4 ** level, max(mm.model_len - 1, 1), 16), np.int32)
        self.L.orc_train_level_counts(m, self._strings(strings), len(strings), level,
                                      out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out

    def model_tables(self, m):
        mm = m.contents
        n = mm.periodicity * mm.num_nodes
        return (np.ctypeslib.as_array(mm.mip, (n,)).copy().reshape(mm.periodicity, mm.num_nodes),
                np.ctypeslib.as_array(mm.prob, (4 * n,)).copy().reshape(mm.periodicity, mm.num_nodes, 4))

    def score_reads_6frame(self, gene, indep, seqs_bytes, n_reads, L):
        out = np.empty((n_reads, 6, L), np.float64)
        self.L.orc_score_reads_6frame(gene, indep, seqs_bytes, n_reads, L, out.ctypes.data_as(dp))
        return out


_inst = None


def load():
    global _inst
    if _inst is None:
        _inst = Oracle()
    return _inst
