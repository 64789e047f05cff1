#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs written by tools/profile_frame6.sh: per-launch averages for k_frame6s."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_frame6t"
for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_frame6" in row["Name"] and kern not in row["Name"]:
            print("other kernel: %s calls=%s avg_ns=%s" % (row["Name"][:40], row["Calls"], row["AverageNs"]))
for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if kern in row["Name"]:
            print("kernel_stats: calls=%s avg_ns=%s min_ns=%s max_ns=%s" % (row["Calls"], row["AverageNs"], row["MinNs"], row["MaxNs"]))
for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if kern in row["Kernel_Name"]:
            print("dispatch: grid=%s wg=%s lds=%s vgpr=%s sgpr=%s scratch=%s" % (
                row.get("Grid_Size"), row.get("Workgroup_Size"), row.get("LDS_Block_Size"), row.get("VGPR_Count"),
                row.get("SGPR_Count"), row.get("Scratch_Size")))
            break
sums, cnt = defaultdict(float), defaultdict(int)
for path in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(path)):
        if kern in row["Kernel_Name"]:
            sums[row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]] += 1
for name in sorted(sums):
    print("%-32s per-launch avg = %.6g   (n=%d)" % (name, sums[name] / cnt[name], cnt[name]))
