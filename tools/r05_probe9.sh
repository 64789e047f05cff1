#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out/r05p9; mkdir -p $R; rm -f $R/*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_mg_err.py tests/test_gpu_mg.py tests/test_gpu_classes.py -m gpu -q -x > $R/pytest1.txt 2>&1; tail -3 $R/pytest1.txt
timeout -k 10 400 python3 tests/bench/stress_mg.py 20000 5 > $R/stress_mg.txt 2>&1; tail -2 $R/stress_mg.txt
V=glimmer-mg_amd/lib/variants
for e in indel sub; do for l in $V/libgmg_r05c.so glimmer-mg_amd/lib/libgmg.so $V/libgmg_r05c.so glimmer-mg_amd/lib/libgmg.so; do GMG_LIB_PATH=$PWD/$l BENCH_OWN_TABLE=1 BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged >> $R/err_ab.jsonl 2>> $R/err.txt; done; done
cat $R/err_ab.jsonl | cut -c1-140
timeout -k 10 300 python3 tools/mg_ab.py ragged $V/libgmg_r05c.so glimmer-mg_amd/lib/libgmg.so > $R/mg_ab.txt 2>&1; tail -2 $R/mg_ab.txt
BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/prof_kernels.sh r05in python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/in_trace.log 2>&1
f=$(find gpurun_out/prof_r05in -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/indel_timeline.txt; tail -12 $R/indel_timeline.txt
echo done
