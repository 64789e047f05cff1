"""ctypes prototypes for include/gmg.h and include/gmg_icm.h.  Loading fails loudly when the
library is missing: there is no Python or CPU fallback."""
import ctypes as C
import os

from .build import LIB

_lib = None

vp = C.c_void_p
u64 = C.c_uint64
i32 = C.c_int


class Segment(C.Structure):
    _fields_ = [("read", C.c_uint32), ("lo", C.c_uint32), ("len", C.c_uint32), ("orient", C.c_uint32)]


class MgGroup(C.Structure):
    _fields_ = [("gene", C.c_void_p), ("read_begin", C.c_uint64), ("read_end", C.c_uint64)]


class OrfParams(C.Structure):
    _fields_ = [("min_gene_len", C.c_int32), ("allow_truncated", C.c_int32), ("use_first_start", C.c_int32),
                ("ignore_score_len", C.c_int32), ("start_threshold", C.c_double), ("n_start_codons", C.c_int32),
                ("start_codon", (C.c_char * 4) * 8)]


class MgParams(C.Structure):
    _fields_ = [("min_gene_len", C.c_int32), ("allow_truncated", C.c_int32), ("ignore_score_len", C.c_int32),
                ("n_start_codons", C.c_int32), ("n_stop_codons", C.c_int32), ("flags", C.c_int32),
                ("start_threshold", C.c_double), ("start_codon", (C.c_char * 4) * 8),
                ("stop_codon", (C.c_char * 4) * 8),
                # the error branch (GMG_MG_ALLOW_INDELS / GMG_MG_ALLOW_SUBS)
                ("min_indel_orf_len", C.c_int32), ("indel_quality_threshold", C.c_int32), ("indel_max", C.c_int32),
                ("circular", C.c_int32), ("indel_suffix_score_threshold", C.c_double), ("quality", C.c_void_p),
                # classification mode: per-read null model / Ignore_Score_Len
                ("nulls", C.c_void_p), ("read_null", C.c_void_p), ("read_ignore_score_len", C.c_void_p),
                # gmg_find_orfs only: ignore regions (glimmer3 -i)
                ("n_ignore_regions", C.c_int32), ("reserved2", C.c_int32), ("ignore_lo", C.c_void_p), ("ignore_hi", C.c_void_p)]


PROTOTYPES = {
    # include/gmg.h
    "gmg_init": (i32, [i32]),
    "gmg_device_count": (i32, []),
    "gmg_last_error": (C.c_char_p, []),
    "gmg_version": (C.c_char_p, []),
    "gmg_synchronize": (i32, [vp]),
    "gmg_shard_plan": (i32, [vp, u64, i32, vp]),
    "gmg_fasta_shard_ranges": (i32, [C.c_char_p, u64, i32, vp]),
    "gmg_gc_fraction": (C.c_double, [vp, vp, i32, i32]),
    "gmg_set_option": (i32, [C.c_char_p, C.c_longlong]),
    "gmg_get_option": (i32, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "gmg_base_code": (i32, [i32]),
    "gmg_pack_bases": (i32, [C.c_char_p, u64, u64, vp]),
    "gmg_packed_words": (u64, [u64]),
    "gmg_model_upload": (i32, [vp, vp, i32, i32, i32, i32, C.POINTER(vp)]),
    "gmg_model_free": (i32, [vp]),
    "gmg_model_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "gmg_reads_upload": (i32, [vp, vp, u64, C.POINTER(vp)]),
    "gmg_reads_wrap_device": (i32, [vp, vp, u64, u64, C.POINTER(vp)]),
    "gmg_reads_free": (i32, [vp]),
    "gmg_reads_info": (i32, [vp, C.POINTER(u64), C.POINTER(u64)]),
    "gmg_reads_download": (i32, [vp, vp, vp]),
    "gmg_reads_select": (i32, [vp, vp, u64, C.POINTER(vp)]),
    "gmg_single_create": (i32, [C.POINTER(vp)]),
    "gmg_single_stage": (i32, [vp, C.c_char_p, u64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "gmg_single_fetch": (i32, [vp, vp, C.c_size_t]),
    "gmg_single_free": (i32, [vp]),
    "gmg_single_window": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "gmg_segments_upload": (i32, [vp, vp, u64, vp, C.POINTER(u64), C.POINTER(vp)]),
    "gmg_segments_free": (i32, [vp]),
    "gmg_frame_score6": (i32, [vp, vp, vp, vp, vp]),
    "gmg_frame_score6_strided": (i32, [vp, vp, vp, vp, u64, vp]),
    "gmg_null_set_upload": (i32, [vp, i32, C.POINTER(vp)]),
    "gmg_null_set_from_tables": (i32, [vp, vp, i32, C.POINTER(vp)]),
    "gmg_null_set_build": (i32, [vp, i32, vp, i32, C.POINTER(vp)]),
    "gmg_null_set_free": (i32, [vp]),
    "gmg_frame_score6_nulls": (i32, [vp, vp, vp, vp, vp, u64, vp]),
    "gmg_segment_frame_score": (i32, [vp, vp, vp, i32, vp, vp]),
    "gmg_segment_cumscore": (i32, [vp, vp, vp, i32, vp, vp]),
    "gmg_score_string": (i32, [vp, vp, vp, i32, vp, vp]),
    "gmg_segment_partial_prob": (i32, [vp, vp, vp, i32, vp, vp]),
    "gmg_all_frame_score": (i32, [vp, vp, vp, vp, vp, vp, vp]),
    "gmg_window_distrib": (i32, [vp, vp, vp, u64, vp, vp, vp]),
    "gmg_mg_score_reads": (i32, [vp, vp, vp, vp, vp, C.POINTER(vp), vp]),
    "gmg_mg_score_groups": (i32, [vp, i32, vp, vp, vp, C.POINTER(vp), vp]),
    "gmg_find_orfs": (i32, [vp, vp, C.POINTER(vp), vp]),
    "gmg_mg_result_info": (i32, [vp, C.POINTER(u64), C.POINTER(u64)]),
    "gmg_mg_result_fetch": (i32, [vp, vp, vp, vp]),
    "gmg_mg_result_fetch_errors": (i32, [vp, vp]),
    "gmg_mg_result_free": (i32, [vp]),
    "gmg_trim_cache": (i32, []),
    "gmg_score_reads_strings": (i32, [vp, i32, vp, vp, vp]),
    "gmg_host_register": (i32, [vp, C.c_size_t]),
    "gmg_host_unregister": (i32, [vp]),
    "gmg_fasta_ingest": (i32, [C.c_char_p, u64, C.POINTER(vp), C.POINTER(vp)]),
    "gmg_fasta_ingest_on": (i32, [C.c_char_p, u64, C.POINTER(vp), C.POINTER(vp), vp]),
    "gmg_fasta_split": (i32, [C.c_char_p, u64, u64, vp, i32]),
    "gmg_mg_result_fetch_on": (i32, [vp, vp, vp, vp, vp]),
    "gmg_stream_create": (i32, [C.POINTER(vp)]),
    "gmg_stream_destroy": (i32, [vp]),
    "gmg_fasta_info": (i32, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
    "gmg_fasta_headers": (i32, [vp, vp, vp]),
    "gmg_fasta_free": (i32, [vp]),
    "gmg_orfs_upload": (i32, [vp, vp, u64, C.POINTER(u64), C.POINTER(vp)]),
    "gmg_orf_batch_free": (i32, [vp]),
    "gmg_score_orfs": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "gmg_score_orfs_begin": (i32, [vp, vp, vp, vp, vp, C.POINTER(u64), vp]),
    "gmg_score_orfs_fetch": (i32, [vp, vp, vp, vp]),
    "gmg_classes_load": (i32, [C.c_char_p, u64, C.c_char_p, C.POINTER(vp)]),
    "gmg_classes_free": (i32, [vp]),
    "gmg_classes_info": (i32, [vp, C.POINTER(u64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(u64)]),
    "gmg_classes_icm_file": (C.c_char_p, [vp, C.c_uint32]),
    "gmg_classes_plan": (i32, [vp, vp, vp, u64, vp, vp, vp, vp, C.POINTER(u64)]),
    "gmg_stop_codons_by_code": (i32, [i32, vp, C.POINTER(i32)]),
    "gmg_ignore_score_len": (i32, [C.c_double, vp, i32, C.POINTER(C.c_int32)]),
    "gmg_trainer_create": (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
    "gmg_trainer_level_counts": (i32, [vp, i32, vp, vp]),
    "gmg_trainer_free": (i32, [vp]),
    "gmg_device_malloc": (i32, [C.POINTER(vp), C.c_size_t]),
    "gmg_device_free": (i32, [vp]),
    "gmg_memcpy_h2d": (i32, [vp, vp, C.c_size_t, vp]),
    "gmg_memcpy_d2h": (i32, [vp, vp, C.c_size_t, vp]),
    # include/gmg_icm.h
    "gmg_icm_new": (i32, [i32, i32, i32, C.POINTER(vp)]),
    "gmg_icm_train": (i32, [C.POINTER(C.c_char_p), i32, i32, i32, i32, C.POINTER(vp)]),
    "gmg_icm_open": (i32, [C.c_char_p, C.POINTER(vp)]),
    "gmg_icm_build_indep": (i32, [vp, C.c_double, C.POINTER(C.c_char_p), i32]),
    "gmg_icm_write": (i32, [vp, C.c_char_p]),
    "gmg_icm_free": (i32, [vp]),
    "gmg_icm_params": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "gmg_icm_tables": (i32, [vp, vp, vp]),
    "gmg_icm_device_model": (i32, [vp, C.POINTER(vp)]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                               "there is no fallback path" % LIB)
        # GMG_LIB_PATH: a variant build of the same sources (kernel experiments, tools/build_variants.sh); never a fallback
        _lib = C.CDLL(os.environ.get("GMG_LIB_PATH") or LIB)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(_lib, name)      # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
    return _lib
