#!/bin/bash
# Round-2 evidence run on the GPU box (one gpurun call): writes everything under gpurun_out/r02/, the summaries are then copied
# into profiles/r02_* by hand.  Every step under its own timeout; output appended as it comes.
set -u
R=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
say() { echo "[$(date +%T)] $*" | tee -a $R/progress.log; }
say bench;       timeout -k 10 400 python3 bench.py > $R/bench.json 2> $R/bench.err
say frame6-prof; timeout -k 10 1500 bash tools/profile_frame6.sh r02f6 > $R/profile_frame6.log 2>&1
say mg;          for m in "" ragged; do timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 3 $m >> $R/mg.jsonl 2>> $R/mg.err; BENCH_OWN_TABLE=1 timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 3 $m >> $R/mg_own.jsonl 2>> $R/mg.err; done
say mg-timing;   BENCH_OWN_TABLE=1 GMG_MG_TIMING=1 timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 2 2> $R/mg_stage_timing.txt > /dev/null
say mg-err;      for e in indel sub; do BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 3 ragged >> $R/mg_err.jsonl 2>> $R/mg.err; done
say mg-trace;    bash tools/prof_kernels.sh r02mg python3 tests/bench/bench_mg.py 1000000 3 > $R/mg_trace.txt 2>&1
say strings;     timeout -k 10 300 python3 tests/bench/bench_strings.py 1000000 64 > $R/strings.json 2>> $R/strings.err; GMG_STRINGS_FUSED=0 timeout -k 10 300 python3 tests/bench/bench_strings.py 1000000 64 > $R/strings_two_pass.json 2>> $R/strings.err
say strings-trace; bash tools/prof_kernels.sh r02str python3 tests/bench/bench_strings.py 1000000 8 > $R/strings_trace.txt 2>&1
say orfs;        timeout -k 10 300 python3 tests/bench/bench_orfs.py > $R/orfs.json 2>> $R/misc.err
say ingest;      timeout -k 10 300 python3 tests/bench/bench_ingest.py > $R/ingest.json 2>> $R/misc.err
say cli;         timeout -k 10 600 python3 tests/bench/bench_cli.py 200000 > $R/cli.json 2>> $R/misc.err; BENCH_CLI_DEV_OPTS="--shards 2" BENCH_CLI_SKIP_G3=1 timeout -k 10 600 python3 tests/bench/bench_cli.py 200000 > $R/cli_shards2.json 2>> $R/misc.err
say train;       timeout -k 10 300 python3 tests/bench/bench_train.py 1600 > $R/train_1600.json 2>> $R/misc.err; timeout -k 10 400 python3 tests/bench/bench_train.py 64000 > $R/train_64000.json 2>> $R/misc.err
say stamps;      true
say done
