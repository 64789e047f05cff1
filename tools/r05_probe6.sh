#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out/r05p6; mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_OWN_TABLE=1 bash tools/prof_kernels.sh r05rg python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/rg_trace.log 2>&1
f=$(find gpurun_out/prof_r05rg -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/ragged_timeline.txt
BENCH_OWN_TABLE=1 BENCH_ERR=sub bash tools/prof_kernels.sh r05sb python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/sb_trace.log 2>&1
f=$(find gpurun_out/prof_r05sb -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/sub_timeline.txt
BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/prof_kernels.sh r05in python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/in_trace.log 2>&1
f=$(find gpurun_out/prof_r05in -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/indel_timeline.txt
echo done
