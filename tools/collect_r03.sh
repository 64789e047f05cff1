#!/bin/bash
# copy the summaries of tools/profile_r03.sh from gpurun_out/r03/ into profiles/r03_* and renew profiles/traffic.json
set -u
cd "$(dirname "$0")/.."
R=gpurun_out/r03
for f in $R/*.json $R/*.jsonl $R/*.csv $R/*.txt; do [ -s "$f" ] && cp "$f" profiles/r03_$(basename "$f"); done
python3 tools/update_traffic.py gpurun_out/prof_r03f6 > /dev/null && echo "traffic.json renewed"
ls profiles/r03_* | wc -l
