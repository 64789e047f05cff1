#!/usr/bin/env python3
"""Timeline of the last gmg_mg_score_reads call in a rocprofv3 kernel trace (csv): tools/mg_timeline.py <kernel_trace.csv>.
Prints every kernel of the call with start / end relative to the call's first kernel: where the streams overlap and where
the device waits for the host."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = from the last k_frame6t on
idx = max(i for i, r in enumerate(rows) if "k_frame6t" in r["Kernel_Name"] or "k_frame6t" in r.get("Kernel_Name", ""))
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    name = r["Kernel_Name"]
    name = name[:name.index("(")] if "(" in name else name
    if "rocprim" in name or "hipcub" in name:
        name = "scan/select " + name.split("::")[-1][:30]
    print("%9.3f %9.3f ms  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Queue_Id", "?"), name[:90]))
