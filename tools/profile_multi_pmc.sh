#!/bin/bash
# L2 hit / miss of k_frame6t<MULTI> (gmg_mg_score_groups, 64 DISTINCT gene tables) beside the single-ICM call: tools/profile_multi_pmc.sh <tag>
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_multi_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BENCH_SAME_MODEL=${2:-relabel} BENCH_PER_GROUP_CALLS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tests/bench/bench_classes.py 1000000 64 100 3 > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
tail -1 "$OUT/trace.log" | cut -c1-400
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
grep "k_frame6" "$f" | cut -d, -f1-4 | cut -c1-200
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/pmc1" -- python3 tests/bench/bench_classes.py 1000000 64 100 1 > "$OUT/pmc1.log" 2>&1 || { tail -5 "$OUT/pmc1.log"; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
s, c = defaultdict(float), defaultdict(int)
for path in glob.glob(os.path.join(sys.argv[1], "pmc1", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_frame6t" in row["Kernel_Name"]:
            multi = "MULTI" if "true>(Frame6Args)" in row["Kernel_Name"] else "single"
            s[(multi, row["Counter_Name"])] += float(row["Counter_Value"]); c[(multi, row["Counter_Name"])] += 1
for k in sorted(s):
    print("%-8s %-22s per-launch avg = %.6g (n=%d)" % (k[0], k[1], s[k] / c[k], c[k]))
PY
