#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out/r05p7; mkdir -p $R; rm -f $R/*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_mg_err.py -m gpu -q -x > $R/pytest1.txt 2>&1; tail -3 $R/pytest1.txt
V=glimmer-mg_amd/lib/variants
for e in indel sub; do for l in $V/libgmg_r05c.so glimmer-mg_amd/lib/libgmg.so $V/libgmg_r05c.so glimmer-mg_amd/lib/libgmg.so; do GMG_LIB_PATH=$PWD/$l BENCH_OWN_TABLE=1 BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged >> $R/err_ab.jsonl 2>> $R/err.txt; done; done
cat $R/err_ab.jsonl | cut -c1-140
BENCH_OWN_TABLE=1 BENCH_ERR=sub bash tools/prof_kernels.sh r05sb python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/sb_trace.log 2>&1
f=$(find gpurun_out/prof_r05sb -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/sub_timeline.txt; sed -n 8,24p $R/sub_timeline.txt
echo done
