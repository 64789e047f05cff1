#!/bin/bash
# Kernel trace of the error branch (glimmer-mg -i) on the GPU box: tools/profile_err.sh [reads]
# (counters: collect them per kernel in short separate runs -- a --pmc pass over the whole bench serialises the long
#  walking kernels and takes tens of minutes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-1000000}
OUT=gpurun_out/prof_err
mkdir -p $OUT
BENCH_ERR=${BENCH_ERR:-indel} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tests/bench/bench_mg.py $N 2 ragged > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-400
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs head -16 | cut -d, -f1-4 | cut -c1-160
