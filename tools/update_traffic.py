#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_frame6.sh: HBM-side bytes per gmg_frame_score6 call
(WRITE_SIZE in KiB + 2 x FETCH_SIZE in KiB, the gfx950 correction of the MI355X guide), stamped with the SHA-256 of the kernel
source they were measured with -- bench.py reports `roofline.traffic` only while that source is unchanged.
    tools/update_traffic.py gpurun_out/prof_<tag> [reads x length, default 1000000x500]"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "glimmer-mg_amd", "csrc", "gmg_frame6.hip")


def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\w+)\s+per-launch avg = ([0-9.e+]+)", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


d = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "1000000x500"
t, p = counters(os.path.join(d, "summary.txt")), counters(os.path.join(d, "summary_k_frame6p.txt"))
parts = {"k_frame6t_write": int(t["WRITE_SIZE"] * 1024), "k_frame6t_fetch_x2": int(2 * t["FETCH_SIZE"] * 1024),
         "k_frame6p_write": int(p["WRITE_SIZE"] * 1024), "k_frame6p_fetch_x2": int(2 * p["FETCH_SIZE"] * 1024)}
doc = {"_note": "HBM-side bytes per gmg_frame_score6 call (k_frame6t + k_frame6p) from rocprofv3 PMC passes (tools/profile_frame6.sh -> "
                "tools/update_traffic.py): WRITE_SIZE (KiB; exact for 16-byte streaming stores) + 2 x FETCH_SIZE (gfx950 reports half of "
                "wide coalesced reads), per launch.  bench.py reports it only while gmg_frame6.hip has the digest below.",
       "source": "glimmer-mg_amd/csrc/gmg_frame6.hip", "source_sha256": hashlib.sha256(open(SRC, "rb").read()).hexdigest(),
       "measured_in": os.path.basename(os.path.normpath(d)), key: sum(parts.values()), "parts": parts}
json.dump(doc, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
