#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out/r05p4; mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_mg.py tests/test_gpu_orfbits.py tests/test_gpu_parity.py tests/test_gpu_classes.py -m gpu -q -x > $R/pytest1.txt 2>&1; tail -5 $R/pytest1.txt
echo "[$(date +%T)] tests 1 done"
V=glimmer-mg_amd/lib/variants
timeout -k 10 300 python3 tools/mg_ab.py $V/libgmg_r04.so $V/libgmg_r05a.so glimmer-mg_amd/lib/libgmg.so > $R/mg_ab.txt 2>&1
timeout -k 10 300 python3 tools/mg_ab.py ragged $V/libgmg_r04.so $V/libgmg_r05a.so glimmer-mg_amd/lib/libgmg.so >> $R/mg_ab.txt 2>&1
cat $R/mg_ab.txt
for l in $V/libgmg_r04.so glimmer-mg_amd/lib/libgmg.so; do
  echo "$l" >> $R/prn_ab.jsonl
  GMG_LIB_PATH=$PWD/$l BENCH_PER_GROUP_CALLS=0 BENCH_SAME_MODEL=1 timeout -k 10 200 python3 tests/bench/bench_classes.py 1000000 64 100 7 >> $R/prn_ab.jsonl 2>> $R/err.txt
done
cat $R/prn_ab.jsonl | cut -c1-200
for e in indel sub; do for l in $V/libgmg_r05a.so glimmer-mg_amd/lib/libgmg.so; do GMG_LIB_PATH=$PWD/$l BENCH_OWN_TABLE=1 BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged >> $R/err_ab.jsonl 2>> $R/err.txt; done; done
cat $R/err_ab.jsonl | cut -c1-300
BENCH_OWN_TABLE=1 bash tools/prof_kernels.sh r05mg4 python3 tests/bench/bench_mg.py 1000000 3 > $R/mg_trace.log 2>&1
f=$(find gpurun_out/prof_r05mg4 -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mg_timeline.txt; cat $R/mg_timeline.txt
echo "[$(date +%T)] done"
