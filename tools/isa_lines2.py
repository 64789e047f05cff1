#!/usr/bin/env python3
"""Static count of vector / scalar / LDS / memory instructions per source line of one kernel, across included files:
    hipcc ... -gline-tables-only --offload-device-only -S x.hip -o x.s ; python tools/isa_lines2.py x.s <mangled name prefix> [min count]"""
import re, collections, sys, os
asm, kern = sys.argv[1:3]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lines = open(asm).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(kern) and ':' in l][0]
end = start
while 's_endpgm' not in lines[end]: end += 1
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
cur = (0, 0); per = collections.defaultdict(collections.Counter); tot = collections.Counter()
for l in lines[start:end + 1]:
    t = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r'([a-z_0-9]+)\s', t + ' ')
    if not m or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    op = m.group(1)
    kind = 'v' if op.startswith('v_') else 's' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'mem'
    tot[kind] += 1; per[cur][kind] += 1
print(dict(tot))
srcs = {}
for (f, ln), c in sorted(per.items()):
    if c['v'] + c['s'] >= minc:
        fn = files.get(f, '?')
        if fn not in srcs:
            try: srcs[fn] = open(fn if fn.startswith('/') else os.path.join('glimmer-mg_amd', fn)).read().split('\n')
            except Exception: srcs[fn] = []
        sl = srcs[fn]
        print("%-16s %5d v%4d s%4d l%3d m%3d  %s" % (fn.split('/')[-1][:16], ln, c['v'], c['s'], c['lds'], c['mem'], sl[ln - 1].strip()[:100] if 0 < ln <= len(sl) else ''))
