#!/usr/bin/env python3
"""Static count of vector instructions per source line of one kernel (which lines does the compiler spend the kernel's instructions on):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -Iinclude -Iglimmer-mg_amd/csrc --offload-device-only -S x.hip -o x.s
    python tools/isa_lines.py x.s <mangled kernel name prefix> <source file> [min count]"""
import collections
import re
import sys

asm, kern, srcf = sys.argv[1:4]
minc = int(sys.argv[4]) if len(sys.argv) > 4 else 8
lines = open(asm).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(kern) and ":" in l][0]
end = start
while "s_endpgm" not in lines[end]:
    end += 1
cur, per, tot = 0, collections.Counter(), collections.Counter()
for l in lines[start:end + 1]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = int(m.group(2))
        continue
    m = re.match(r"([a-z_0-9]+)\s", t + " ")
    if not m or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = m.group(1)
    kind = "v" if op.startswith("v_") else "s" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem" if op.startswith(("global_", "buffer_", "scratch_")) else "?"
    tot[kind] += 1
    if kind == "v":
        per[cur] += 1
src = open(srcf).read().split("\n")
print(dict(tot))
for ln in sorted(per):
    if per[ln] >= minc:
        print("%5d %4d  %s" % (ln, per[ln], src[ln - 1].strip()[:130] if 0 < ln <= len(src) else ""))
