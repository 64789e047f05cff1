#!/usr/bin/env python3
"""print the rows of a rocprofv3 kernel_stats.csv: tools/kstats.py <dir or file> [rows]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print(p)
for r in list(csv.DictReader(open(p)))[:n]:
    print("%-100s %5s %10.3f ms avg %8.3f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
