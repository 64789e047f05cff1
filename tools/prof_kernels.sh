#!/bin/bash
# kernel trace of one command on the GPU box: tools/prof_kernels.sh <tag> <command...>  -> gpurun_out/prof_<tag>/kernel_stats (top 14 rows printed)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- "$@" > "$OUT/run.log" 2>&1
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
echo "== $TAG"; [ -n "$f" ] && head -15 "$f" | cut -d, -f1-5 | cut -c1-170 || tail -5 "$OUT/run.log"
