#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out/r05p5; mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_mg.py -m gpu -q -x > $R/pytest1.txt 2>&1; tail -3 $R/pytest1.txt
V=glimmer-mg_amd/lib/variants
timeout -k 10 300 python3 tools/mg_ab.py $V/libgmg_r05b.so glimmer-mg_amd/lib/libgmg.so > $R/mg_ab.txt 2>&1
timeout -k 10 300 python3 tools/mg_ab.py ragged $V/libgmg_r05b.so glimmer-mg_amd/lib/libgmg.so >> $R/mg_ab.txt 2>&1
cat $R/mg_ab.txt
for e in indel sub; do for l in $V/libgmg_r05b.so glimmer-mg_amd/lib/libgmg.so; do GMG_LIB_PATH=$PWD/$l BENCH_OWN_TABLE=1 BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged >> $R/err_ab.jsonl 2>> $R/err.txt; done; done
cat $R/err_ab.jsonl | cut -c1-140
BENCH_OWN_TABLE=1 bash tools/prof_kernels.sh r05mg5 python3 tests/bench/bench_mg.py 1000000 3 > $R/mg_trace.log 2>&1
f=$(find gpurun_out/prof_r05mg5 -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mg_timeline.txt; cat $R/mg_timeline.txt
echo "[$(date +%T)] done"
