"""glimmer-mg_amd -- MI355X (gfx950) native IMM scorer for Glimmer-MG.

The product is the C-ABI shared library `lib/libgmg.so` (include/gmg.h, include/gmg_icm.h):
hand-written HIP kernels + the host-side C++ ICM_t.  This Python package is only a ctypes
binding over that ABI for tests and bench.py; it contains no scoring code and no fallback.
"""
from . import build as build            # noqa: F401
from . import capi as capi              # noqa: F401
from . import synth as synth            # noqa: F401
from . import shard as shard            # noqa: F401
from .api import (Icm, Reads, NullSet, Segments, Trainer, init, frame_score6, segment_frame_score, segment_cumscore,  # noqa: F401
                  score_string, segment_partial_prob, all_frame_score, window_distrib, score_orfs, find_orfs, mg_score_reads, score_reads_strings, GmgError,
                  FORWARD, REVERSED, COMPLEMENTED, REVCOMP, read_fasta, set_option, get_option, option)
