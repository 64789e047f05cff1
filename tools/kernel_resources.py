#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel of one source file, from hipcc's own remarks:
    python tools/kernel_resources.py glimmer-mg_amd/csrc/gmg_mg.hip [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "glimmer-mg_amd", "csrc"), "--offload-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
cmd += ["-D" + d for d in os.environ.get("DEFINES", "").split()]
err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip().replace("(MgArgs)", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print("%-60s vgpr %3d agpr %3d  waves/SIMD %d  lds %6d  spill v%d s%d" % (k[:60], v.get("VGPRs", 0), v.get("AGPRs", 0), v.get("Occupancy [waves/SIMD]", 0),
              v.get("LDS Size [bytes/block]", 0), v.get("VGPRs Spill", 0), v.get("SGPRs Spill", 0)))
